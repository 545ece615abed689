#!/bin/bash
# A/B helper for the flat BVH kernel: kernel us of three shapes through variant 2
for args in "--worlds 4096 --width 256 --height 256 --textured --mode Raytracer --steps 50 --warmup 10" "--worlds 4096 --width 128 --height 128 --wall --steps 300 --warmup 50" "--worlds 1024 --width 128 --height 128 --mode Raytracer --steps 500 --warmup 50" "--worlds 1024 --steps 1000 --warmup 100"; do
  MRX_PLACEMENT_TRIES=${TRIES:-2} python3 bench.py --no-extra --no-cpu-baseline --no-strong --variant 2 $args 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('   %-70s kernel %8.2f us  %s' % ('$args'[:70], o['roofline']['kernel_us'], o['placement']['candidates_us']))"
done
