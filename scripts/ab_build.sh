#!/bin/bash
# Build libmrx_hip.so variants for an A/B on the GPU box:
#   scripts/ab_build.sh <name> [extra hipcc flags]   ->  ab/libmrx_hip.so.<name>
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
C="$ROOT/madrona_renderer_amd/csrc"
mkdir -p "$ROOT/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -mllvm -disable-promote-alloca-to-vector -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=12 -Wall -Wno-unused-function "$@" \
  "$C/raster.hip" "$C/bvh.hip" "$C/bvh.cpp" "$C/mrx_api.cpp" "$C/assets.cpp" "$C/ktx2.cpp" \
  -lz -ldl -Wl,-rpath,/opt/rocm/lib -o "$ROOT/ab/libmrx_hip.so.$NAME"
