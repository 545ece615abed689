#!/bin/bash
# small-batch shapes of the raster group kernel under MRX_GROUP_VIEWS / MRX_XCD_SKEW overrides (kernel us)
for w in 1024 2048; do
  for gv in 0 1 2 4; do
    for sk in -1; do
      MRX_GROUP_VIEWS=$gv python3 bench.py --no-extra --no-cpu-baseline --no-strong --worlds $w --steps 3000 --warmup 300 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('worlds %5d  MRX_GROUP_VIEWS=%d  kernel %6.2f us  wall %6.2f us' % ($w, $gv, o['roofline']['kernel_us'], o['ms_per_step']*1000))"
    done
  done
done
