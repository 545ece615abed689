#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers rest on.
#   scripts/profile_round.sh r01      (run on the GPU box through gpurun)
# Writes gpurun_out/<tag>/: kernel_stats.csv (--kernel-trace --stats of the default
# bench command), pmc_write.csv / pmc_fetch.csv (separate --pmc passes), summary.json.
set -eu
export TMPDIR=/tmp
TAG="${1:-r02}"
OUT="${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
STEPS="${STEPS:-2000}"
# BENCH_ARGS selects another configuration, e.g. "--worlds 4096 --width 128 --height 128 --wall"
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 200 --no-cpu-baseline --no-extra --no-strong ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmcw" -- $BENCH > /dev/null 2> "$OUT/pmcw.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmcf" -- $BENCH > /dev/null 2> "$OUT/pmcf.log"
cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"
python3 - "$OUT" "$STEPS" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def is_ours(name):
    return 'raster' in name or 'bvh' in name
def pmc(d, name):
    vals = []
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if is_ours(row['Kernel_Name']) and row['Counter_Name'] == name:
                vals.append(float(row['Counter_Value']))
    return vals
w, f = pmc('pmcw', 'WRITE_SIZE'), pmc('pmcf', 'FETCH_SIZE')
stats = [r for r in csv.DictReader(open(f"{out}/kernel_stats.csv")) if is_ours(r['Name'])]
# the timed region is the last STEPS launches (bench.py renders for a quarter
# second first, to let the clocks settle, then warms up, then times)
steps = int(sys.argv[2])
rows = []
for path in glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if is_ours(row['Kernel_Name']):
            rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp'])))
rows.sort()
timed = rows[-steps:]
# WRITE_SIZE / FETCH_SIZE are in KB; gfx950 FETCH_SIZE reports half of a wide
# coalesced read stream (MI355X_MICROARCH.md, HBM section) -> doubled
summary = {
    "kernel": stats[0]['Name'] if stats else None,
    "calls": int(stats[0]['Calls']) if stats else 0,
    "avg_ns": float(stats[0]['AverageNs']) if stats else None,
    "timed_region_calls": len(timed),
    "timed_region_avg_ns": sum(e - s for s, e in timed) / len(timed) if timed else None,
    "write_bytes_per_launch": sum(w) / len(w) * 1024 if w else None,
    "fetch_bytes_per_launch_corrected": sum(f) / len(f) * 1024 * 2 if f else None,
}
if w and f:
    summary["hbm_bytes_per_launch"] = summary["write_bytes_per_launch"] + summary["fetch_bytes_per_launch_corrected"]
json.dump(summary, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(summary))
PY
tail -1 "$OUT/bench_trace.json"
