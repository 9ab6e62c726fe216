#!/bin/bash
# Diagnostic (-DEU_DEBUG_SKIP build): duration of the shade kernel with parts replaced by constants (EU_DEBUG_SKIP_SHADE bits:
# 1 background lookup, 2 surface colour (constant, opaque), 4 reflection ratio (0)).  Results are NOT frames: timing only.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mask in ${MASKS:-0 1 2 4 7}; do
  rm -rf $R/gpurun_out/st_$mask
  EU_DEBUG_SKIP_SHADE=$mask EU_WF_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st_$mask -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --max-depth ${DEPTH:-1} > $R/gpurun_out/st_$mask.log 2>&1
  python3 - $R/gpurun_out/st_$mask $mask <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "eu_wf_shade" in r["Kernel_Name"] or "eu_wf_intersect" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0][-28:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("skip-shade", sys.argv[2], {k: "%.1f us x%d" % (sum(v) / len(v), len(v) // 4) for k, v in d.items()})
PY
done
