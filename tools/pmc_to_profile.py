"""Turn the rocprofv3 --pmc passes of tools/pmc_passes.sh (gpurun_out/pmc) into profiles/<tag>_pmc.json and refresh
profiles/traffic.json (HBM bytes and wave-level VALU instructions of one frame pipeline).
Usage: python tools/pmc_to_profile.py r02_room "stage description" ["3d_room.json 1920x1080 depth 8" [pmc_dir [round]]] """
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stage = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "3d_room.json 1920x1080 depth 8"
pmc_dir = sys.argv[4] if len(sys.argv) > 4 else "pmc"
round_no = int(sys.argv[5]) if len(sys.argv) > 5 else 3
vals = collections.defaultdict(lambda: collections.defaultdict(list))
# one file per pass: gpurun MERGES its output into gpurun_out/, so an earlier refresh leaves its files (other run ids) beside the new ones
latest = {}
for f in glob.glob(os.path.join(ROOT, "gpurun_out", pmc_dir, "*/*/*counter_collection.csv")):
    key = os.path.relpath(f, os.path.join(ROOT, "gpurun_out", pmc_dir)).split(os.sep)[0]
    if key not in latest or os.path.getmtime(f) > os.path.getmtime(latest[key]):
        latest[key] = f
for f in sorted(latest.values()):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "eu_" in name:
            vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
frames = None
per_kernel, derived, launches = {}, {}, {}
for name, cs in vals.items():
    per_kernel[name] = {c: {"avg_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in sorted(cs.items())}
for name in vals:
    n = per_kernel[name]["SQ_WAVES"]["launches"] if "SQ_WAVES" in per_kernel[name] else 0
    if "intersect0" in name:      # one generation-0 intersect launch per frame (single band stream, frames up to 4 Mpixel)
        frames = n
for name in vals:
    n = max(v["launches"] for v in per_kernel[name].values())
    launches[name] = n // frames if frames else n
    c = {k: v["avg_per_launch"] for k, v in per_kernel[name].items()}
    d = {}
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"]:
        d["VALU_lane_utilisation_pct"] = round(100 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64), 1)
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if k in c:
                d[k + "_over_SQ_WAVE_CYCLES_pct"] = round(100 * c[k] / c["SQ_WAVE_CYCLES"], 1)
    if "SQ_INSTS_VALU" in c:
        d["SQ_INSTS_VALU_per_launch"] = c["SQ_INSTS_VALU"]
    derived[name] = d
fetch = sum(launches[n] * per_kernel[n]["FETCH_SIZE"]["avg_per_launch"] * 1024 for n in vals if "FETCH_SIZE" in per_kernel[n])
write = sum(launches[n] * per_kernel[n]["WRITE_SIZE"]["avg_per_launch"] * 1024 for n in vals if "WRITE_SIZE" in per_kernel[n])
# Calibration on kernels of known traffic (round 1: 4 B and 8 B per lane accesses read exactly -- a kernel reading 72.9 MB with
# 8-B/4-B loads -> FETCH_SIZE 70.5 MB; one storing 121.5 MB -> WRITE_SIZE 122.1 MB); only the pack kernel's 16 B per
# lane loads are reported at half (8.29 MB -> 4.16 MB), as MI355X_MICROARCH.md says for wide loads: its FETCH_SIZE is doubled.
wide = [n for n in vals if n.startswith("eu_pack_rgb") and "FETCH_SIZE" in per_kernel[n]]
fetch_corrected = fetch + sum(launches[n] * per_kernel[n]["FETCH_SIZE"]["avg_per_launch"] * 1024 for n in wide)
valu = sum(launches[n] * derived[n].get("SQ_INSTS_VALU_per_launch", 0.0) for n in vals)
lanes = sum(launches[n] * derived[n].get("SQ_INSTS_VALU_per_launch", 0.0) * derived[n].get("VALU_lane_utilisation_pct", 0.0) / 100 for n in vals)
out = {"round": round_no, "stage": stage + "; counters collected with one band stream (bench.py --streams 1) so that per-dispatch counters do not mix between concurrent kernels",
       "workload": workload, "launches_per_frame_single_stream": launches, "derived": derived, "per_kernel": per_kernel,
       "frame_hbm_bytes": {"FETCH_SIZE_sum": fetch, "WRITE_SIZE_sum": write, "hbm_bytes_per_frame_raw": fetch + write,
                           "hbm_bytes_per_frame_fetch_doubled": 2 * fetch + write, "hbm_bytes_per_frame_calibrated": fetch_corrected + write,
                           "note": "sum over the frame's launches of rocprofv3 FETCH_SIZE/WRITE_SIZE (KiB*1024), separate --pmc passes "
                                   "(tools/pmc_passes.sh). MI355X_MICROARCH.md: FETCH_SIZE reads half the bytes of wide (16 B/lane) streaming "
                                   "reads; the accesses here are 4-8 B/lane (uncalibrated), so both the raw sum and the doubled-read figure are given"}}
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w"), indent=1)
tp = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tp)) if os.path.exists(tp) else {}
t[workload] = {"hbm_bytes_per_launch": fetch_corrected + write, "fetch_bytes": fetch_corrected, "write_bytes": write,
               "definition": "one 'launch' = the frame pipeline of one eu_render_device call (per generation one intersect + one shade launch, then one resolve launch per generation, + pack); "
                             "FETCH_SIZE+WRITE_SIZE summed over its kernels (profiles/%s_pmc.json), FETCH_SIZE of the pack kernel (16 B/lane loads) doubled; "
                             "4-8 B/lane accesses read exactly (calibrated in round 1 on kernels of known traffic)" % tag,
               "valu_wave_insts_per_launch": valu, "valu_lane_utilisation": lanes / valu if valu else None,
               "valu_definition": "SQ_INSTS_VALU (wave-level VALU instructions) summed over the frame pipeline's kernels, and their mean active-lane share (profiles/%s_pmc.json)" % tag}
json.dump(t, open(tp, "w"), indent=1)
print(json.dumps({"fetch": fetch, "write": write, "valu": valu, "lane_util": lanes / valu if valu else None, "launches": launches}, indent=1))
for n, d in derived.items():
    print(n, d)
