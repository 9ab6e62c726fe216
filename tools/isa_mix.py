"""Instruction mix of the trace kernels per 64-ray batch, round over round: dynamic counts from the PMC passes (profiles/<tag>_pmc.json:
SQ_INSTS_* per launch, one frame at a time) divided by the batches of an average launch, next to the static register / spill figures of
the specialised kernels' code object.  Usage: python tools/isa_mix.py  ->  profiles/r03_isa_mix.json"""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RAYS = {"r02_room": 9602240, "r03_room": 9602251}
out = {"note": "dynamic wave-level instructions per 64-ray batch (a frame's rays / 64 / the kernel's launches per frame = batches per launch; "
               "SQ_INSTS_* per launch from separate --pmc passes); 3d_room 1920x1080 depth 8", "kernels": {}}
for tag in ("r02_room", "r03_room"):
    d = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc.json")))
    for k, v in d["per_kernel"].items():
        if "intersect" not in k and "shade" not in k:
            continue
        launches = sum(n for kk, n in d["launches_per_frame_single_stream"].items() if ("intersect" in kk) == ("intersect" in k) and ("shade" in kk) == ("shade" in k))
        batches = RAYS[tag] / 64.0 / launches
        out["kernels"][tag + ":" + k] = {c.replace("SQ_INSTS_", "").lower() + "_per_batch": round(x["avg_per_launch"] / batches, 1)
                                         for c, x in v.items() if c.startswith("SQ_INSTS_")}
# static figures of the specialised kernels (3d_room, f64)
from euclider_amd import Parser  # noqa: E402
env = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_room.json"))
info = env.jit_precompile(os.path.join(ROOT, "euclider_amd", "jit_cache"))
env.close()
co = glob.glob(os.path.join(ROOT, "euclider_amd", "jit_cache", info["key"] + ".hsaco"))[0]
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
static = {}
for block in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    static[name] = {f: int(re.search(r"\.%s:\s+(\d+)" % f, block).group(1)) for f in
                    ("vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
out["specialised_kernels_static"] = static
out["interpreter_kernels_static_round2"] = {"eu_wf_intersect_kernel<3,0>": "129-158 VGPRs, 46-51 spilled SGPRs, no scratch", "eu_wf_shade_kernel<3,true>": "168 VGPRs, 218-329 spilled SGPRs, up to 50 spilled VGPRs"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_isa_mix.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
