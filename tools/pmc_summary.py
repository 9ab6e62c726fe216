"""Summarise rocprofv3 --pmc CSVs (tools/pmc_passes.sh output) per kernel (diagnostic helper)."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
out = collections.defaultdict(dict)
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "eu_" in name:
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, cn), v in agg.items():
        out[name][cn] = sum(v) / len(v)     # average per launch
for name, c in out.items():
    line = {k: float("%.4g" % v) for k, v in sorted(c.items())}
    d = {}
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["VALU_lane_util_%"] = round(100 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64), 1)
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if k in c:
                d[k + "/WAVE_CYCLES_%"] = round(100 * c[k] / c["SQ_WAVE_CYCLES"], 1)
    print(name, json.dumps(line))
    print("   ", d)
