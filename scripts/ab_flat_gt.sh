#!/bin/bash
# flat kernel: tiles of a view per workgroup (MRX_BVH_GROUP_TILES) on the multi-tile shapes, kernel us
run() {
  local name="$1"; shift
  for gt in $GTS; do
    MRX_BVH_GROUP_TILES=$gt python3 bench.py --no-extra --no-cpu-baseline --no-strong --variant 2 "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('   %-26s group tiles %2d  kernel %8.2f us  %s' % ('$name', $gt, o['roofline']['kernel_us'], o['placement']['candidates_us']))"
  done
}
GTS="16 8 4 2" run "c5 4096x256^2 rt tex" --worlds 4096 --width 256 --height 256 --textured --mode Raytracer --steps 50 --warmup 10
GTS="16 8 4" run "c5/8 512x256^2 rt tex" --worlds 512 --width 256 --height 256 --textured --mode Raytracer --steps 300 --warmup 50
GTS="4 2 1" run "c3 4096x128^2+wall" --worlds 4096 --width 128 --height 128 --wall --steps 300 --warmup 50
GTS="4 2 1" run "1024x128^2 rt" --worlds 1024 --width 128 --height 128 --mode Raytracer --steps 500 --warmup 50
