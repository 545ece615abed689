#!/usr/bin/env python3
"""Copies what scripts/profile_all.sh <tag> left under gpurun_out/ into profiles/ (tracked) and rewrites
profiles/pmc_latest.json from it:  python scripts/collect_profiles.py r04
  gpurun_out/<tag>_<cfg>/{summary.json,kernel_stats.csv,bench.json} -> profiles/<tag>_<cfg>_{summary.json,kernel_stats.csv,bench.json}
SQ-counter entries ("sq") of pmc_latest.json come from scripts/pmc.sh runs: gpurun_out/<tag>_<cfg>_sq.txt, if present
(SQ_INSTS_VALU / SQ_WAVES ... per launch), and carry the hash of the kernel sources (bench.bvh_kernel_hash) so that
bench.py can tell counters of another build from its own."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONFIGS = {   # name -> (pmc_latest key, bench args)
    "hl": ("4096x64x64", ""),
    "c2": ("1024x64x64", "--worlds 1024"),
    "c4shard": ("2048x64x64", "--worlds 2048 --first-world 14336"),
    "c3": ("4096x128x128+wall", "--worlds 4096 --width 128 --height 128 --wall"),
    "c5": ("4096x256x256+tex+rt", "--worlds 4096 --width 256 --height 256 --textured --mode Raytracer"),
    "c5bvh": ("4096x256x256+tex+rt+variant2", "--worlds 4096 --width 256 --height 256 --textured --mode Raytracer --variant 2"),
    "c5raster": ("4096x256x256+tex+rt+variant3", "--worlds 4096 --width 256 --height 256 --textured --mode Raytracer --variant 3"),
    "bvh482": ("1024x64x64+cubes40", "--worlds 1024 --cubes 40"),
    "bvh1202": ("1024x64x64+cubes100", "--worlds 1024 --cubes 100"),
}


def sq_table(path):
    d = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+n=\s*\d+ mean=(\S+)", line)
        if m:
            d[m.group(1)] = float(m.group(2))
    return d


def main():
    tag = sys.argv[1]
    out = {}
    for name, (key, args) in CONFIGS.items():
        src = os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, name))
        if not os.path.exists(os.path.join(src, "summary.json")):
            continue
        for f in ("summary.json", "kernel_stats.csv", "bench.json"):
            if os.path.exists(os.path.join(src, f)):
                shutil.copy(os.path.join(src, f), os.path.join(ROOT, "profiles", "%s_%s_%s" % (tag, name, f)))
        s = json.load(open(os.path.join(src, "summary.json")))
        ent = {"bytes": s.get("hbm_bytes_per_launch"),
               "source": "profiles/%s_%s_summary.json: rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE (separate passes, KB units; "
                         "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section), mean per launch of `bench.py --no-extra "
                         "--no-cpu-baseline --no-strong %s`" % (tag, name, args)}
        sq = os.path.join(ROOT, "gpurun_out", "%s_%s_sq.txt" % (tag, name))
        if os.path.exists(sq):
            t = sq_table(sq)
            if t.get("SQ_WAVES") and t.get("SQ_INSTS_VALU"):
                # the unit bench.py counts in: per (view, tile, wave) -- 8 waves per tile
                units = 1024 * 8 if "cubes" in key else t["SQ_WAVES"]
                ent["sq"] = {"valu_per_wave": t["SQ_INSTS_VALU"] / units,
                             "salu_per_wave": t.get("SQ_INSTS_SALU", 0.0) / units,
                             "lds_per_wave": t.get("SQ_INSTS_LDS", 0.0) / units,
                             "kernel_hash": bench.bvh_kernel_hash(),
                             "source": "profiles/%s_%s_pmc_sq.txt (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ..., scripts/pmc.sh; "
                                       "per (view, wave) = per launch / %d)" % (tag, name, units)}
                shutil.copy(sq, os.path.join(ROOT, "profiles", "%s_%s_pmc_sq.txt" % (tag, name)))
        out[key] = ent
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
    for k, v in out.items():
        print(k, "%.4g B" % v["bytes"], "sq" if "sq" in v else "")


if __name__ == "__main__":
    main()
