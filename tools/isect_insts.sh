#!/bin/bash
# Diagnostic (needs a -DEU_PROFILE_ISECT -DEU_PROFILE_ISECT2 build, which has no s_memtime stamps active in the hot loop...):
# SQ_INSTS_VALU of the intersect kernel with entity subsets left out -> wave-level instructions per entity.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mask in ${MASKS:-0x00 0x7f 0x4f 0x3f 0x7b 0x77 0x7e 0x7d}; do
  rm -rf $R/gpurun_out/ii_$mask
  EU_DEBUG_SKIP_ENTITIES=$mask EU_WF_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/ii_$mask -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-depth 1 > $R/gpurun_out/ii_$mask.log 2>&1
  python3 - $R/gpurun_out/ii_$mask $mask <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "intersect" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("skip", sys.argv[2], {k: "%.4g" % v for k, v in agg.items()}, "launches", dict(n))
PY
done
