#!/bin/bash
# Per-kernel register / LDS / scratch figures of a .hip file, from the device
# assembly hipcc emits (no GPU needed):  scripts/kernel_stats.sh raster.hip
set -eu
SRC="$(cd "$(dirname "$0")/.." && pwd)/madrona_renderer_amd/csrc/${1:-raster.hip}"
OUT="${TMPDIR:-/tmp}/kstats_$$.s"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math \
  -mllvm -disable-promote-alloca-to-vector -fno-slp-vectorize \
  --cuda-device-only -S -o "$OUT" "$SRC"
python3 - "$OUT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    pass
# the amdhsa.kernels metadata block: one entry per kernel
meta = txt[txt.index("amdhsa.kernels"):]
for ent in meta.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", ent) or [None, "?"])[1]
    name = g("name")
    short = re.sub(r"^_ZN3mrx\d+_GLOBAL__N_1", "", name)
    print(f"{short[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} "
          f"vspill {g('vgpr_spill_count'):>3s} sspill {g('sgpr_spill_count'):>3s} "
          f"lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")
PY
rm -f "$OUT"
