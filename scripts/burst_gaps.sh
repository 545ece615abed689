#!/bin/bash
set -eu
export TMPDIR=/tmp
OUT="${GRAFT_REPO_ROOT:?}/gpurun_out/${1:-bursts}"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
python3 "$GRAFT_REPO_ROOT/scripts/burst_gaps.py" > "$OUT/plain.txt" 2>&1 || true
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/scripts/burst_gaps.py" > "$OUT/traced.txt" 2> "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if 'raster' in r['Kernel_Name'] or 'bvh' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
rows.sort()
# bursts: separated by gaps > 1 ms
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 1_000_000:
        bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
for i, bu in enumerate(bursts[-6:]):
    gaps = [(k + 1, (bu[k + 1][0] - bu[k][1]) / 1e3) for k in range(len(bu) - 1)]
    big = ["#%d: %.1f us" % (k, g) for k, g in gaps if g > 1.5]
    durs = [(e - s) / 1e3 for s, e in bu]
    print("burst %d: %d launches, first %.2f us, mean of the rest %.2f us, span/launch %.2f us; idle before launch %s" % (
        i, len(bu), durs[0], sum(durs[1:]) / max(1, len(durs) - 1), (bu[-1][1] - bu[0][0]) / 1e3 / len(bu), ", ".join(big) or "none"))
PY
cat "$OUT/plain.txt" "$OUT/traced.txt" | grep burst
