#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy summary of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [file.hip] [include dir] [substring filter ...]"""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "thr3ed_atom_amd/csrc/relu_field_kernels.hip"
inc = sys.argv[2] if len(sys.argv) > 2 else "include"
filters = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", inc, src, "-o", "/tmp/kr.so",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
keys = ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]")
for line in out.splitlines():
    if " error" in line:
        print(line)
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur)
        cur = re.sub(r"\(.*", "", cur).replace("void ", "")
        rows[cur] = {}
        continue
    for k in keys:
        m = re.search(re.escape(k) + r": (\d+)", line)
        if m and cur and k not in rows[cur]:
            rows[cur][k] = int(m.group(1))
for k, v in rows.items():
    if filters and not any(f in k for f in filters):
        continue
    g = lambda n: v.get(n, 0)
    print(f"{k:52s} vgpr {g('VGPRs'):4d} agpr {g('AGPRs'):3d} scratch {g('ScratchSize [bytes/lane]'):4d} occ {g('Occupancy [waves/SIMD]')} sgpr-spill {g('SGPRs Spill'):3d} vgpr-spill {g('VGPRs Spill'):3d} lds {g('LDS Size [bytes/block]')}")
