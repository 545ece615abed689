#!/bin/bash
# Round profile set (run on the GPU box through gpurun):  scripts/profile_all.sh r02
# For every configuration bench.py is quoted on: rocprofv3 --kernel-trace --stats and the
# separate --pmc WRITE_SIZE / FETCH_SIZE passes (scripts/profile_round.sh), summaries under
# gpurun_out/<tag>_<config>/summary.json, plus the un-profiled bench line of each.
set -eu
R="${1:-r02}"
ONLY="${2:-}"          # optional: space-separated list of configuration names to run
run() {  # name, steps, bench args
  local name="$1" steps="$2"; shift 2
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return 0; fi
  echo "== $name: $*"
  STEPS="$steps" BENCH_ARGS="$*" bash "$GRAFT_REPO_ROOT/scripts/profile_round.sh" "${R}_${name}" > "$GRAFT_REPO_ROOT/gpurun_out/${R}_${name}.log" 2>&1 || tail -5 "$GRAFT_REPO_ROOT/gpurun_out/${R}_${name}.log"
  python3 "$GRAFT_REPO_ROOT/bench.py" --steps "$steps" --warmup 200 --no-extra --no-cpu-baseline --no-strong "$@" > "$GRAFT_REPO_ROOT/gpurun_out/${R}_${name}/bench.json" 2>/dev/null || true
  tail -2 "$GRAFT_REPO_ROOT/gpurun_out/${R}_${name}.log" | cut -c1-600
}
run hl 2000
run c2 2000 --worlds 1024
run c4shard 2000 --worlds 2048 --first-world 14336
run c3 500 --worlds 4096 --width 128 --height 128 --wall
run c5 100 --worlds 4096 --width 256 --height 256 --textured --mode Raytracer
run c5bvh 30 --worlds 4096 --width 256 --height 256 --textured --mode Raytracer --variant 2
run c5raster 100 --worlds 4096 --width 256 --height 256 --textured --mode Raytracer --variant 3
run bvh482 1000 --worlds 1024 --cubes 40
run bvh1202 500 --worlds 1024 --cubes 100
