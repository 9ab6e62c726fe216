"""The one output of the real reference binary that ships with it (preview/preview_3d_fresnel_sphere.png) against the oracle,
at picture level (tools/preview_fit.py wrote tests/golden/preview_3d_fresnel_fit.json in the container that has /root/reference).

What the fit pins: the RawImage2d bytes carry no gamma (the window shows them through an sRGB decode), the restated
uv_sphere_3 orientation is the one of the four flips that fits, the glass sphere lines up at the fitted pose.  What it does not
pin: numbers (the screenshot's field of view is not the shipped camera's; see `reading` in the JSON)."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "preview_3d_fresnel_fit.json")
REF = "/root/reference/preview/preview_3d_fresnel_sphere.png"


def test_committed_fit_says_what_design_md_says():
    d = json.load(open(GOLDEN))
    tc = d["transfer_candidates_coarse_cost"]
    assert d["transfer_between_rawimage_and_screenshot"] == "srgb_decode"
    assert tc["srgb_decode"] < tc["identity"] < tc["srgb_encode"]
    uv = d["background_scan"]["uv_variants_at_fov_90"]
    assert min(uv, key=lambda k: uv[k]["cost"]) == "as_implemented"
    assert uv["as_implemented"]["cost"] + 8 < min(uv[k]["cost"] for k in uv if k != "as_implemented")


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree is only present in the build container")
def test_transfer_function_recheck_against_the_screenshot():
    """Re-derives the colour-transfer finding from the screenshot itself at the committed pose (one oracle render)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import preview_fit as pf
    from PIL import Image
    d = json.load(open(GOLDEN))
    p = tuple(d["fitted_pose"]["location"]) + (d["fitted_pose"]["yaw_rad"], d["fitted_pose"]["pitch_rad"])
    osc = pf.sl.load_scene_file(pf.SCENE)
    ref = np.asarray(Image.open(REF).convert("RGB"))
    ref_small = pf.reduced(ref, 256, 192, 1.5)
    raw = pf.render(osc, p, 256, 192)
    cost = {name: float(np.abs(pf.reduced(tf(raw), 256, 192, 1.5) - ref_small).mean()) for name, tf in pf.TRANSFERS.items()}
    assert cost["srgb_decode"] < cost["identity"] < cost["srgb_encode"], cost
