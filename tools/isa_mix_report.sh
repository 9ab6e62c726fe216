#!/bin/bash
# Static instruction mix of the frame pipeline's kernels (D = 3 variants the 3d_room bench runs) -> profiles/r02_isa_mix.json.
# Compiles a probe translation unit that instantiates just those kernels (seconds, no GPU needed).
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d)
cat > $tmp/probe.hip <<'EOP'
#include <hip/hip_runtime.h>
#include "trace_device.h"
#include "trace_wavefront.h"
template __global__ void eu_wf_gen_kernel<3>(const uint64_t *, uint32_t, EuDevCamera, EuDevFrame, EuWfBuffers, EuDevCounters *, uint32_t *, double *, double *);
template __global__ void eu_wf_intersect_kernel<3, 0>(const uint64_t *, uint32_t, uint32_t, uint32_t, uint32_t, EuWfBuffers, EuDevCounters *, double *);
template __global__ void eu_wf_shade_kernel<3, true>(const uint64_t *, uint32_t, uint32_t, uint32_t, double, EuWfBuffers, EuDevCounters *, uint32_t *, double *);
template __global__ void eu_wf_resolve_kernel<3>(uint32_t, EuWfBuffers, EuDevCounters *, uint32_t *, double *);
EOP
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 --cuda-device-only -S -I$root/euclider_amd/csrc $tmp/probe.hip -o $tmp/probe.s 2>/dev/null
python3 - $tmp/probe.s $root/profiles/r02_isa_mix.json <<'PY'
import collections, json, re, sys
s = open(sys.argv[1]).read()
out = {}
for f in re.split(r'\n(?=_Z\w+:)', s):
    m = re.match(r'(_Z\w+):', f)
    if not m or "eu_wf" not in m.group(1):
        continue
    name = re.search(r'eu_wf_\w+?_kernel', m.group(1)).group(0)
    lines = [l.strip() for l in f.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(l.split()[0] for l in lines if l)
    valu = {k: v for k, v in c.items() if k.startswith("v_")}
    f64 = sum(v for k, v in valu.items() if "_f64" in k)
    trans = sum(v for k, v in valu.items() if re.match(r"v_(rcp|rsq|sqrt)_f64", k))
    out[name] = {"instructions": sum(c.values()), "valu": sum(valu.values()), "valu_f64": f64, "valu_f64_share": f64 / max(1, sum(valu.values())),
                 "valu_f64_transcendental": trans, "salu": sum(v for k, v in c.items() if k.startswith("s_")),
                 "lds": sum(v for k, v in c.items() if k.startswith("ds_")), "vmem": sum(v for k, v in c.items() if k.startswith(("global_", "flat_", "buffer_", "scratch_")))}
json.dump({"note": "static mix of the shipped D = 3 kernels (hipcc -S); MI355X_MICROARCH.md: a wave64 f64 VALU instruction occupies its SIMD for 4 cycles, a 32-bit one for 2; bench.py weights the VALU issue peak with these shares and the kernels' dynamic VALU counts (profiles/r02_room_pmc.json)", "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
rm -rf $tmp
