#!/usr/bin/env python3
"""Register / spill / scratch usage of every kernel of a device source, as the compiler
reports it (-Rpass-analysis=kernel-resource-usage) under the product's flags.  No GPU needed.

  python scripts/kernel_resources.py madrona_renderer_amd/csrc/bvh.hip [extra hipcc flags]
  python scripts/kernel_resources.py --write      # profiles/kernel_resources_latest.txt, both sources
"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-mllvm", "-disable-promote-alloca-to-vector", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=12"]


def resources(src, extra=()):
    p = subprocess.run(["hipcc"] + FLAGS + list(extra) + ["-c", src, "-o", "/dev/null",
                                                         "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr)
        raise SystemExit(p.returncode)
    out, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?): (\S+) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            name = subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()
            name = name.replace("void mrx::(anonymous namespace)::", "").replace("(mrx::RasterParams)", "")
            cur = {"name": name}
            out.append(cur)
        elif cur is not None:
            cur[key] = val
    return out


def line(k):
    return ("%-52s VGPR %3s  SGPR %3s  scratch %4s B/lane  vgpr-spill %3s  sgpr-spill %3s  occupancy %s"
            % (k["name"], k.get("VGPRs"), k.get("TotalSGPRs"), k.get("ScratchSize [bytes/lane]"),
               k.get("VGPRs Spill"), k.get("SGPRs Spill"), k.get("Occupancy [waves/SIMD]")))


SOURCES = ["madrona_renderer_amd/csrc/raster.hip", "madrona_renderer_amd/csrc/bvh.hip"]

if __name__ == "__main__":
    if sys.argv[1:2] == ["--write"]:
        # the committed table tests/test_kernel_resources.py compares HEAD with
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with open(os.path.join(root, "profiles", "kernel_resources_latest.txt"), "w") as f:
            f.write("## hipcc -Rpass-analysis=kernel-resource-usage under the product's flags (scripts/kernel_resources.py --write);\n"
                    "## tests/test_kernel_resources.py fails when this table is not what HEAD compiles to\n")
            for src in SOURCES:
                f.write("# %s\n" % src)
                for k in resources(os.path.join(root, src)):
                    f.write(line(k) + "\n")
        raise SystemExit(0)
    for k in resources(sys.argv[1], sys.argv[2:]):
        print(line(k))
