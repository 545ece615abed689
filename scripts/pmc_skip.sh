#!/bin/bash
# SQ_INSTS_VALU / SALU per ablation level (MRX_DEBUG_SKIP) -- instruction budget per phase.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp
for skip in 15 7 3 1 0; do
  export MRX_DEBUG_SKIP=$skip
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/skip$skip -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/skip$skip.log 2>&1
  python3 - $OUT/skip$skip $skip <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'raster' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
print('skip', sys.argv[2], ' '.join(f"{k}={sum(v)/len(v)/4096:.0f}/tile" for k, v in sorted(agg.items())))
PY
done
