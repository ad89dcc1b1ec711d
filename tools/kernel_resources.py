#!/usr/bin/env python
"""Per-kernel register / LDS / scratch usage from the code-object metadata of a device-only assembly listing:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include --cuda-device-only -S -o /tmp/capi.s segmminterest_amd/csrc/capi.hip
    python tools/kernel_resources.py /tmp/capi.s [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
meta = txt[txt.index("amdhsa.kernels:"):]
ents = re.split(r"\n  - \.", meta)[1:]
flt = sys.argv[2:]
rows = []
for e in ents:
    f = dict(re.findall(r"\.?(\w+):\s+(\S+)", e))
    name = f.get("name", "?")
    try:
        name = subprocess.check_output(["c++filt", name], text=True).strip()
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("segmm::", "")
    if flt and not any(s in name for s in flt):
        continue
    rows.append((name, f.get("vgpr_count"), f.get("agpr_count"), f.get("sgpr_count"), f.get("vgpr_spill_count"), f.get("private_segment_fixed_size"),
                 f.get("group_segment_fixed_size"), f.get("max_flat_workgroup_size")))
print("%-60s %5s %5s %5s %6s %8s %8s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "spill", "scratch", "lds", "wg"))
for r in sorted(rows):
    print("%-60s %5s %5s %5s %6s %8s %8s %6s" % r)
