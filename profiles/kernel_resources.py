#!/usr/bin/env python
"""Per-kernel register / LDS / scratch usage of a HIP source as hipcc reports it for gfx950.
usage: python profiles/kernel_resources.py ophelia_amd/csrc/oph_kernels.hip [regex]"""
import re
import subprocess
import sys

src = sys.argv[1]
filt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()}
        rows.append(cur)
    else:
        cur[k.split(" ")[0]] = v
for r in rows:
    if filt.search(r["name"]):
        print("%-72s vgpr=%s agpr=%s sgpr=%s scratch=%s occ=%s lds=%s" % (r["name"][:72], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                                                        r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
