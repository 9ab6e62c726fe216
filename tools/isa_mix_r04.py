"""Round 4: what prices `roofline.valu` in bench.py, from THIS round's kernels only.
Static: the f64 share of every specialised kernel's VALU instructions, counted in the disassembly of the code object the default renderer
loads (euclider_amd/jit_cache, llvm-objdump).  Dynamic (when profiles/r04_<tag>_pmc.json exist): wave-level VALU instructions per launch
and launches per frame of each kernel.  -> profiles/r04_isa_mix.json: per workload {kernel: {valu_static, valu_f64_static, valu_f64_share,
SQ_INSTS_VALU_per_launch, launches_per_frame}, cycles_per_valu_instruction (f64: 4 SIMD cycles, other VALU: 2, weighted with the dynamic counts)}.
Usage: python tools/isa_mix_r04.py"""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euclider_amd import Parser  # noqa: E402

WORKLOADS = {"room": ("3d_room.json", "3d_room.json 1920x1080 depth 8"), "hallways": ("3d_hallways.json", "3d_hallways.json 1920x1080 depth 12"),
             "4dframe": ("4d_frame.json", "4d_frame.json 1920x1080 depth 8")}
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
out = {"note": "static f64 share of the VALU instructions of each specialised kernel (disassembly of the cached code object) and, where this round's PMC passes "
               "exist, the dynamic VALU counts that weight them; a wave64 f64 VALU instruction occupies its SIMD for 4 cycles, any other VALU instruction for 2",
       "workloads": {}}
for tag, (scene, key) in WORKLOADS.items():
    env = Parser().parse_file(os.path.join(ROOT, "scenes", scene))
    info = env.jit_precompile(os.path.join(ROOT, "euclider_amd", "jit_cache"))
    env.close()
    co = glob.glob(os.path.join(ROOT, "euclider_amd", "jit_cache", info["key"] + ".hsaco"))[0]
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {"valu_static": 0, "valu_f64_static": 0}
            continue
        t = line.split()
        if cur is None or not t:
            continue
        op = t[0]
        if op.startswith("v_") and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_nop")):
            kernels[cur]["valu_static"] += 1
            if "f64" in op or op.startswith(("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
                kernels[cur]["valu_f64_static"] += 1
    kernels = {k: v for k, v in kernels.items() if v["valu_static"] > 50 and "eu_" in k}
    for v in kernels.values():
        v["valu_f64_share"] = round(v["valu_f64_static"] / v["valu_static"], 4)
    rec = {"workload": key, "code_object_key": info["key"], "kernels": kernels}
    pmc_path = os.path.join(ROOT, "profiles", "r04_%s_pmc.json" % tag)
    if os.path.exists(pmc_path):
        d = json.load(open(pmc_path))
        num = den = 0.0
        for k, dv in d["derived"].items():
            short = k.split("<")[0].split("(")[0]
            cand = [n for n in kernels if n.startswith(short) or short.startswith(n.split("(")[0][:20])]
            if "SQ_INSTS_VALU_per_launch" not in dv:
                continue
            n_l = d["launches_per_frame_single_stream"].get(k, 1)
            share = kernels[cand[0]]["valu_f64_share"] if cand else None
            if cand:
                kernels[cand[0]]["SQ_INSTS_VALU_per_launch"] = dv["SQ_INSTS_VALU_per_launch"]
                kernels[cand[0]]["launches_per_frame"] = n_l
            if share is not None:
                wgt = dv["SQ_INSTS_VALU_per_launch"] * n_l
                num += wgt * (4.0 * share + 2.0 * (1.0 - share))
                den += wgt
        if den:
            rec["cycles_per_valu_instruction"] = round(num / den, 4)
            rec["valu_wave_insts_per_frame"] = den
    out["workloads"][key] = rec
    print(key, {k: v["valu_f64_share"] for k, v in kernels.items()}, rec.get("cycles_per_valu_instruction"))
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_isa_mix.json"), "w"), indent=1)
