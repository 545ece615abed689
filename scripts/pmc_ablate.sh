#!/bin/bash
# Dynamic instruction counts of the BVH kernel's phases: SQ counters with phases switched off (MRX_DEBUG_SKIP).
# Needs a library built with MRX_EXTRA_HIPCC_FLAGS=-DMRX_BVH_DIAG=1 (python -m madrona_renderer_amd.build --force).
set -u
export BENCH_ARGS="--cubes 40 --worlds 1024"
for skip in 0 32 64 96 2 10 14 1; do
  echo "== MRX_DEBUG_SKIP=$skip"
  MRX_DEBUG_SKIP=$skip bash scripts/pmc.sh pmc_abl_$skip "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY" | grep -v GRBM
done
