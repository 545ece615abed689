#!/bin/bash
# Per-launch durations and gaps of the driver-style K = 20 timed region (rocprofv3 --kernel-trace of
# bench.py --steps 20 --warmup 5): where does K = 20 lose against the settled 22.6 us?  (GPU box)
set -eu
export TMPDIR=/tmp
OUT="${GRAFT_REPO_ROOT:?}/gpurun_out/${1:-k20trace}"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-strong ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if 'raster' in r['Kernel_Name'] or 'bvh' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][40:80]))
rows.sort()
last = rows[-26:]
prev_end = None
for s, e, n in last:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("dur %7.2f us  gap before %9.2f us  %s" % ((e - s) / 1e3, gap, n))
    prev_end = e
t = last[-20:]
print("timed 20: span %.2f us/step, mean duration %.2f us" % ((t[-1][1] - t[0][0]) / 20e3, sum(e - s for s, e, _ in t) / 20e3))
PY
