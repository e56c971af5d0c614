#!/bin/bash
# Per-kernel register / LDS / occupancy report of the HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")/../genz-tokenize_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-align-mismatch \
    -x hip gz_kernels.hip -c -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    k, _, v = m.group(1).partition(": ")
    k = k.strip()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k] = v
print("%-28s %5s %5s %7s %6s %5s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ"))
for r in rows:
    n = re.sub(r"^_Z\d+", "", r["name"]); n = re.split(r"(PK|P|I[a-z]E|\d|ILi)", n)[0] if n.startswith("gz_") else n
    print("%-28s %5s %5s %7s %6s %5s" % (r["name"][:60] if not n else n[:28], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
