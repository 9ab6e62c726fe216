#!/bin/bash
# rocprofv3 per-kernel statistics of bench.py (run on the GPU box from the repo root).  Usage: BENCH_ARGS="--streams 1" tools/kernel_stats.sh <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o k -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --repeats 1 --no-alone --no-other-configs $BENCH_ARGS > gpurun_out/prof_$tag.log 2>&1 || exit 1
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/prof_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
for row in csv.DictReader(open(f[0])):
    print("%-64s calls %6s total_ms %9.3f avg_us %9.2f  %5s%%" % (row["Name"][:64], row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY
