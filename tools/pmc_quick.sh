#!/bin/bash
# Diagnostic: one rocprofv3 --pmc pass (instruction counts) of a short bench run.  Usage: BENCH_ARGS="..." tools/pmc_quick.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pq_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --repeats 1 --no-alone --no-other-configs --streams 1 --frames-in-flight 1 ${BENCH_ARGS:-} > $R/gpurun_out/pq_$tag.log 2>&1 || { echo "rc=$?"; tail -5 $R/gpurun_out/pq_$tag.log; exit 1; }
python3 - "$R/gpurun_out/pq_$tag" <<'PY'
import csv, glob, sys, collections
v = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "eu_" in n: v[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(v.items()):
    w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"])
    print("%-28s launches %3d waves %9.0f" % (n, len(cs["SQ_WAVES"]), w), " ".join("%s/wave %.0f" % (c.replace("SQ_", ""), sum(x) / len(x) / w) for c, x in sorted(cs.items()) if c != "SQ_WAVES"))
PY
