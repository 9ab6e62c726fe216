#!/bin/bash
# Diagnostic: every kernel dispatch of the LAST frame of a short bench run, in start order, with duration and gap to the
# previous dispatch's end (rocprofv3 --kernel-trace).  Usage: tools/frame_dispatches.sh <tag> [ENV=VAL ...]   (BENCH_ARGS for bench.py)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/disp_$tag -o k -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline $BENCH_ARGS > gpurun_out/disp_$tag.log 2>&1 || exit 1
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/disp_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
# last frame: from the last eu_wf_gen / eu_ts kernel on
starts = [i for i, r in enumerate(rows) if "gen_kernel" in r["Kernel_Name"] or "eu_ts_kernel" in r["Kernel_Name"]]
first = starts[-1] if starts else 0
# with two band streams the last frame has two gen kernels
if len(starts) > 1 and "gen_kernel" in rows[starts[-1]]["Kernel_Name"] and starts[-1] - starts[-2] < 4:
    first = starts[-2]
prev_end = None
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%8.1f us  +%7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:60]))
    prev_end = max(prev_end or 0, e)
PY
