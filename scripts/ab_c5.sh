#!/bin/bash
# configs[4] through the BVH path, kernel us (MRX_PLACEMENT_TRIES=1: the first allocation, whatever its mode)
for gt in 0 8 4; do
  if [ $gt = 0 ]; then unset MRX_BVH_GROUP_TILES; else export MRX_BVH_GROUP_TILES=$gt; fi
  python3 bench.py --no-extra --no-cpu-baseline --no-strong --variant 2 --worlds 4096 --width 256 --height 256 --textured --mode Raytracer --steps 50 --warmup 10 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('   group tiles %-7s kernel %8.2f us  %s' % ('$gt' if '$gt' != '0' else 'default', o['roofline']['kernel_us'], o['placement']['candidates_us']))"
done
