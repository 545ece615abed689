#!/bin/bash
# BASELINE configs[2] (4096 worlds x 128x128 cube+plane+wall): the LDS tile-size sweep on the round-4 kernels, kernel us
line() {
  python3 -c "
import json,sys
o=json.loads(sys.stdin.readline()); r=o['roofline']
print('   %-58s kernel %8.2f us  %.3f of 8 TB/s   %s' % ('$1', r['kernel_us'], r.get('frac_hbm', r['frac']), o['placement']['candidates_us']))"
}
A="--no-extra --no-cpu-baseline --no-strong --worlds 4096 --width 128 --height 128 --wall --steps 200 --warmup 30"
python3 bench.py $A 2>/dev/null | line "default dispatch: tiled raster kernel (z in registers)"
python3 bench.py $A --variant 2 2>/dev/null | line "LDS depth buffer, flat kernel, 64x64 tiles"
MRX_BVH_FLAT=0 python3 bench.py $A --variant 2 2>/dev/null | line "LDS depth buffer, general kernel, 64x64 tiles"
MRX_BVH_TILE=1 python3 bench.py $A --variant 2 2>/dev/null | line "LDS depth buffer, general kernel, 64x32 tiles"
MRX_BVH_TILE=2 python3 bench.py $A --variant 2 2>/dev/null | line "LDS depth buffer, general kernel, 32x32 tiles"
