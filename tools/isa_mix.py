"""Static instruction mix per kernel from a hipcc --save-temps .s file (diagnostic helper)."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in re.split(r'\n(?=_Z\w+:)', s):
    m = re.match(r'(_Z\w+):', f)
    if not m or flt not in m.group(1):
        continue
    lines = [l.strip() for l in f.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(l.split()[0] for l in lines if l)
    g = collections.Counter()
    for k, v in c.items():
        if k.startswith("scratch_"): g["scratch"] += v
        elif k.startswith("ds_"): g["lds"] += v
        elif k.startswith(("global_", "flat_", "buffer_")): g["vmem"] += v
        elif k.startswith("s_"): g["salu"] += v
        elif "_f64" in k: g["f64"] += v
        elif k.startswith("v_"): g["valu_other"] += v
    print(m.group(1)[:48], sum(c.values()), dict(g), "rd/wr-lane", c.get("v_readlane_b32", 0), c.get("v_writelane_b32", 0),
          "calls", c.get("s_swappc_b64", 0))
