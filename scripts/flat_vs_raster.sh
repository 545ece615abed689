#!/bin/bash
# On the GPU box: the BASELINE shapes through the default dispatch (raster group kernel) and through the BVH
# path's flat kernel (variant 2), back to back on one box.  usage: scripts/flat_vs_raster.sh <out file>
set -u
OUT="$1"; : > "$OUT"
run() {
  local name="$1"; shift
  for v in 0 2; do
    python3 bench.py --no-extra --no-cpu-baseline --no-strong --variant $v "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('%-28s variant %d  kernel %8.2f us  wall %8.2f us  frac_kernel %.3f  placement %s' % ('$name', $v, o['roofline']['kernel_us'], o['ms_per_step']*1000, o['roofline'].get('frac_hbm', o['roofline']['frac']), o['placement']['candidates_us']))" >> "$OUT"
  done
}
run "hl 4096x64^2" --steps 2000 --warmup 200
run "c2 1024x64^2" --steps 2000 --warmup 200 --worlds 1024
run "c3 4096x128^2+wall" --steps 300 --warmup 50 --worlds 4096 --width 128 --height 128 --wall
run "c3tex 4096x128^2+wall tex" --steps 300 --warmup 50 --worlds 4096 --width 128 --height 128 --wall --textured
run "c5 4096x256^2 rt tex" --steps 50 --warmup 10 --worlds 4096 --width 256 --height 256 --textured --mode Raytracer
run "c5/8 512x256^2 rt tex" --steps 300 --warmup 50 --worlds 512 --width 256 --height 256 --textured --mode Raytracer
run "1024x128^2 rt" --steps 300 --warmup 50 --worlds 1024 --width 128 --height 128 --mode Raytracer
