#!/bin/bash
# configs[4]'s shape by view count: default dispatch (raster kernel) against the BVH path (flat kernel), kernel us
for w in 256 512 1024 2048 4096; do
  for v in 0 2; do
    python3 bench.py --no-extra --no-cpu-baseline --no-strong --variant $v --worlds $w --width 256 --height 256 --textured --mode Raytracer --steps 100 --warmup 20 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('   %5d views x 256^2 rt tex  variant %d  kernel %8.2f us  %s' % ($w, $v, o['roofline']['kernel_us'], o['placement']['candidates_us']))"
  done
done
for w in 1024 4096; do
  for v in 0 2; do
    python3 bench.py --no-extra --no-cpu-baseline --no-strong --variant $v --worlds $w --width 256 --height 256 --mode Raytracer --steps 100 --warmup 20 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('   %5d views x 256^2 rt untex variant %d  kernel %8.2f us  %s' % ($w, $v, o['roofline']['kernel_us'], o['placement']['candidates_us']))"
  done
done
for w in 1024 4096; do
  for v in 0 2; do
    python3 bench.py --no-extra --no-cpu-baseline --no-strong --variant $v --worlds $w --width 256 --height 256 --textured --steps 100 --warmup 20 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('   %5d views x 256^2 raster-mode tex variant %d  kernel %8.2f us  %s' % ($w, $v, o['roofline']['kernel_us'], o['placement']['candidates_us']))"
  done
done
