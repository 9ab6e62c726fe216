#!/bin/bash
# Diagnostic: every kernel dispatch of the LAST frame of a short bench run, in start order, with duration and gap to the
# previous dispatch's end (rocprofv3 --kernel-trace).  Usage: BENCH_ARGS="--streams 1" tools/frame_dispatches.sh <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/disp_$tag -o k -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --repeats 1 --no-alone $BENCH_ARGS > gpurun_out/disp_$tag.log 2>&1 || exit 1
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/disp_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
# last frame: from the last generation-0 intersect kernel on (with several band streams: the first of the last group)
starts = [i for i, r in enumerate(rows) if "intersect0" in r["Kernel_Name"]]
first = starts[-1] if starts else 0
while first in starts and starts.index(first) > 0 and first - starts[starts.index(first) - 1] < 4:
    first = starts[starts.index(first) - 1]
prev_end = None
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%8.1f us  +%7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:60]))
    prev_end = max(prev_end or 0, e)
PY
