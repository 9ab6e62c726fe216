#!/bin/bash
# Diagnostic (-DEU_DEBUG_SKIP build): duration of the generation-0 intersect kernel with entity subsets left out.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mask in ${MASKS:-0x00 0x7f}; do
  rm -rf $R/gpurun_out/it_$mask
  EU_DEBUG_SKIP_ENTITIES=$mask EU_WF_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/it_$mask -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --max-depth ${DEPTH:-1} > $R/gpurun_out/it_$mask.log 2>&1
  python3 - $R/gpurun_out/it_$mask $mask <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "eu_" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0][-28:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("skip", sys.argv[2], {k: "%.1f" % (sum(v) / len(v)) for k, v in d.items()})
PY
done
