"""Per-kernel timeline of the last frame in a rocprofv3 --kernel-trace CSV (diagnostic helper)."""
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if "eu_wf_gen" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
tot = {}
for r in last:
    if "eu_" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[name] = tot.get(name, 0) + dur
    print("%-30s start %8.1f us  dur %8.1f us  grid %s" % (name, (int(r["Start_Timestamp"]) - t0) / 1e3, dur, r["Grid_Size_X"]))
print({k: round(v, 1) for k, v in tot.items()}, "sum", round(sum(tot.values()), 1))
