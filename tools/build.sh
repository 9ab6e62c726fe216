#!/bin/bash
# Forced rebuild of libeuclider_amd.so with the per-kernel resource report in build_report.txt (diagnostic helper).
set -e
cd "$(dirname "$0")/../euclider_amd/csrc"
touch renderer.hip
make EXTRA="-Rpass-analysis=kernel-resource-usage $*" > /tmp/eu_build.log 2>&1 || { grep -E "error" /tmp/eu_build.log | head; exit 1; }
python3 - <<'PY'
import re
txt=open("/tmp/eu_build.log").read()
cur=None
for line in txt.split("\n"):
    m=re.search(r"Function Name: (\S+)", line)
    if m: cur=m.group(1); vals={}
    for key in ("VGPRs:","AGPRs:","ScratchSize [bytes/lane]:","Occupancy [waves/SIMD]:","SGPRs Spill:","VGPRs Spill:"):
        if key in line and cur:
            vals[key]=line.split(key)[1].split()[0]
            if key=="VGPRs Spill:" and ("eu_wf" in cur) and "ILi4" not in cur:
                print("%-46s vgpr %s agpr %s scratch %s occ %s sgpr-spill %s vgpr-spill %s" % (cur[:46], vals.get("VGPRs:"), vals.get("AGPRs:"), vals.get("ScratchSize [bytes/lane]:"), vals.get("Occupancy [waves/SIMD]:"), vals.get("SGPRs Spill:"), vals.get("VGPRs Spill:")))
PY
ls -la --time-style=full-iso ../libeuclider_amd.so | cut -c30-120
