#!/bin/bash
# rocprofv3 PMC passes for the raster kernel (run on the GPU box via gpurun).
# usage: scripts/pmc.sh <outdir-under-gpurun_out> "<counters pass 1>" "<counters pass 2>" ...
set -eu
export TMPDIR=/tmp
OUT="${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/pass$i" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-strong ${BENCH_ARGS:-} > "$OUT/pass$i.log" 2>&1 || tail -5 "$OUT/pass$i.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'raster' in row['Kernel_Name'] or 'bvh' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
