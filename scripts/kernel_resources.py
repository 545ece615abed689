#!/usr/bin/env python3
"""Register / spill / scratch usage of every kernel of a device source, as the compiler
reports it (-Rpass-analysis=kernel-resource-usage) under the product's flags.  No GPU needed.

  python scripts/kernel_resources.py madrona_renderer_amd/csrc/bvh.hip [extra hipcc flags]
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


if __name__ == "__main__":
    for k in resources(sys.argv[1], sys.argv[2:]):
        print("%-52s VGPR %3s  SGPR %3s  scratch %4s B/lane  vgpr-spill %3s  sgpr-spill %3s  occupancy %s"
              % (k["name"], k.get("VGPRs"), k.get("TotalSGPRs"), k.get("ScratchSize [bytes/lane]"),
                 k.get("VGPRs Spill"), k.get("SGPRs Spill"), k.get("Occupancy [waves/SIMD]")))
