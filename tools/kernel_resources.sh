#!/bin/bash
# VGPRs / SGPRs / spills / LDS / occupancy of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), e.g.
#   tools/kernel_resources.sh gm_render.hip [dir-with-the-sources]
f=${1:?file}; d=${2:-$(dirname $0)/../gaussianmesh_amd/csrc}
extra=""
case $f in gm_render.hip) extra="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form";; gm_deform.hip) extra="-fno-slp-vectorize";; esac
cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    n = r["name"]
    try:
        n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        pass
    n = re.sub(r"\(.*", "", n)
    print("%4s VGPRs %3s SGPRs  spill v/s %s/%s  LDS %6s  occ %s  %s" % (r.get("VGPRs"), r.get("SGPRs"), r.get("VGPRs Spill", r.get("VGPR Spill")), r.get("SGPRs Spill", r.get("SGPR Spill")), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]"), n[:120]))
'
