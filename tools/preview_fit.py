"""Use the one output of the REAL reference binary that ships with it -- preview/preview_3d_fresnel_sphere.png (1024x768 =
simulation.rs:51 at resolution 1, scenes/3d_fresnel.json, resources/pixelcg_uv.jpg) -- to check the oracle at picture level.

The screenshot was taken after the author had moved the camera, so the pose is fitted first: location (3), yaw and pitch of
the PitchYawCamera3 (d3/entity/camera.rs:76-145), by minimising the mean absolute difference of blurred, reduced images
(Nelder-Mead from a coarse grid).  Then the full-size oracle frame at the fitted pose is compared with the screenshot:
  - flat background tiles far from the sphere: settles gamma / no gamma, to_pixel, the texture's orientation (uv_sphere_3);
  - the whole frame: Fresnel / Snell / general_rotation at a visible level (the grid seen through the glass).
Writes tests/golden/preview_3d_fresnel_fit.json (pose + error metrics) and, with --png, a side-by-side picture.
Needs /root/reference (this container only); the committed JSON is what the CPU test reads.
Usage: python tools/preview_fit.py [--png out.png] [--quick]"""
import json
import math
import os
import sys

import numpy as np
from PIL import Image, ImageFilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import scene_loader as sl  # noqa: E402

REF = "/root/reference/preview/preview_3d_fresnel_sphere.png"
SCENE = os.path.join(ROOT, "scenes", "3d_fresnel.json")
W, H, DEPTH = 1024, 768, 10


def pose_camera(osc, p):
    x, y, z, yaw, pitch = p
    cam = osc.camera()
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    fwd = (cp * cy, cp * sy, sp)
    up = (-sp * cy, -sp * sy, cp)
    left = (up[1] * fwd[2] - up[2] * fwd[1], up[2] * fwd[0] - up[0] * fwd[2], up[0] * fwd[1] - up[1] * fwd[0])
    for i in range(3):
        cam.location[i] = (x, y, z)[i]
        cam.forward[i] = fwd[i]
        cam.up[i] = up[i]
        cam.left[i] = left[i]
    return cam


def render(osc, p, w, h):
    rgb, _, _ = osc.render(w, h, max_depth=DEPTH, camera=pose_camera(osc, p))
    return rgb[::-1]          # row 0 of the RawImage2d is the bottom row on screen


def srgb_decode(img):
    """What the screenshot shows of the RawImage2d bytes: the window's GL path treats them as sRGB-encoded (see the note in main)."""
    x = img.astype(np.float64) / 255.0
    lin = np.where(x <= 0.04045, x / 12.92, np.power((x + 0.055) / 1.055, 2.4))
    return np.clip(lin * 255.0 + 0.5, 0, 255).astype(np.uint8)


def srgb_encode(img):
    x = img.astype(np.float64) / 255.0
    enc = np.where(x <= 0.0031308, x * 12.92, 1.055 * np.power(x, 1 / 2.4) - 0.055)
    return np.clip(enc * 255.0 + 0.5, 0, 255).astype(np.uint8)


TRANSFERS = {"identity": lambda a: a, "srgb_decode": srgb_decode, "srgb_encode": srgb_encode}


def reduced(img, w, h, blur):
    im = Image.fromarray(img).convert("RGB").resize((w, h), Image.BILINEAR).filter(ImageFilter.GaussianBlur(blur))
    return np.asarray(im, dtype=np.float64)


def background_scan(ref_full):
    """Background only (no sphere), in numpy: the direction -> texture mapping under the four flips of (u, v), and the field of
    view as a free parameter; compared with the screenshot outside the sphere's silhouette, best yaw / pitch for each."""
    from scipy.optimize import minimize
    Wb, Hb = 256, 192
    ref = np.asarray(Image.fromarray(ref_full).resize((Wb, Hb), Image.BILINEAR).filter(ImageFilter.GaussianBlur(1.0)), dtype=np.float64)
    tex = np.asarray(Image.open(os.path.join(ROOT, "resources", "pixelcg_uv.jpg")).convert("RGB"), dtype=np.float64) / 255.0
    texlin = np.where(tex <= 0.04045, tex / 12.92, ((tex + 0.055) / 1.055) ** 2.4) * 255.0      # the transfer found above
    TH, TW = tex.shape[:2]
    ys, xs = np.mgrid[0:Hb, 0:Wb]
    mask = (xs - 430 / 4) ** 2 + (ys - 392 / 4) ** 2 > (350 / 4) ** 2
    relx = (xs - Wb / 2) + 0.5
    rely = ((Hb - 1 - ys) - Hb / 2) + 0.5

    def render_bg(yaw, pitch, fov, variant):
        dist = math.sqrt(Wb * Wb + Hb * Hb) / (2 * math.tan(math.radians(fov) / 2))
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        fwd = np.array([cp * cy, cp * sy, sp]); up = np.array([-sp * cy, -sp * sy, cp]); right = np.cross(fwd, up)
        d = fwd[None, None, :] * dist + up[None, None, :] * rely[..., None] + right[None, None, :] * relx[..., None]
        d /= np.linalg.norm(d, axis=2, keepdims=True)
        u = 0.5 + np.arctan2(d[..., 1], d[..., 0]) / (2 * math.pi)
        v = 0.5 - np.arcsin(d[..., 2]) / math.pi
        if variant & 1:
            u = 1 - u
        if variant & 2:
            v = 1 - v
        return texlin[np.clip((v * TH).astype(int), 0, TH - 1), np.clip((u * TW).astype(int) % TW, 0, TW - 1)]

    def fit(variant, fov, free_fov):
        def cost(p):
            return float(np.abs(render_bg(p[0], p[1], p[2] if free_fov else fov, variant) - ref)[mask].mean())
        grid = [(y, pp, fov) for y in np.linspace(-math.pi, math.pi, 73)[:-1] for pp in np.linspace(-1.2, 1.2, 25)]
        g = min(grid, key=cost)
        r = minimize(cost, np.array(g), method="Nelder-Mead", options={"maxiter": 200, "xatol": 1e-4, "fatol": 1e-3})
        return {"cost": float(r.fun), "yaw": float(r.x[0]), "pitch": float(r.x[1]), "fov_deg": float(r.x[2]) if free_fov else fov}
    out = {"uv_variants_at_fov_90": {name: fit(v, 90.0, False) for v, name in enumerate(["as_implemented", "u_flipped", "v_flipped", "both_flipped"])},
           "fov_free": {"start_%d" % f: fit(0, float(f), True) for f in (45, 60, 90)}}
    return out


def main():
    quick = "--quick" in sys.argv
    ref_full = np.asarray(Image.open(REF).convert("RGB"))
    osc = sl.load_scene_file(SCENE)
    from scipy.optimize import minimize
    # which transfer function lies between the RawImage2d bytes and the screenshot?  Fit a coarse pose under each and keep the best.
    transfer_costs = {}
    ref_small0 = reduced(ref_full, 128, 96, 2.0)
    cands0 = [(x, y, z, yaw, pitch) for x in (2.0, 3.5, 5.0) for y in (-1.0, 0.0, 1.0) for z in (-0.5, 0.0, 0.5) for yaw in (-0.1, 0.0, 0.1) for pitch in (-0.1, 0.0, 0.1)]
    raw0 = [render(osc, p, 128, 96) for p in cands0]
    for name, tf in TRANSFERS.items():
        transfer_costs[name] = min(float(np.abs(reduced(tf(r), 128, 96, 2.0) - ref_small0).mean()) for r in raw0)
    transfer = min(transfer_costs, key=transfer_costs.get)
    tf = TRANSFERS[transfer]
    print("transfer candidates (coarse-grid best cost):", transfer_costs, "->", transfer, flush=True)
    # Stage 1: orientation from the background alone.  The background is a texture on the sphere at infinity (uv_sphere_3 of the
    # DIRECTION, universe/mod.rs:183), so it depends on yaw and pitch only; with max_depth 0 every pixel is background.  Compared
    # on the parts of the screenshot that are certainly background (right quarter, top and bottom strips).
    rw, rh = 256, 192
    ref_small = reduced(ref_full, rw, rh, 1.5)
    bg_mask = np.zeros((rh, rw), dtype=bool)
    bg_mask[:, int(rw * 0.80):] = True
    bg_mask[:int(rh * 0.06), :] = True
    bg_mask[int(rh * 0.95):, :] = True

    def bg_cost(yp):
        cam = pose_camera(osc, (0.0, 0.0, 0.0, yp[0], yp[1]))
        rgb, _, _ = osc.render(rw, rh, max_depth=0, camera=cam)
        return float(np.abs(reduced(tf(rgb[::-1]), rw, rh, 1.5) - ref_small)[bg_mask].mean())
    grid = [(y, p) for y in np.linspace(-math.pi, math.pi, 49)[:-1] for p in np.linspace(-1.2, 1.2, 17)]
    yp0 = min(grid, key=bg_cost)
    res = minimize(bg_cost, np.array(yp0), method="Nelder-Mead", options={"maxiter": 200, "xatol": 1e-5, "fatol": 1e-4})
    yaw, pitch = float(res.x[0]), float(res.x[1])
    print("stage 1 (orientation from the background): yaw %.4f pitch %.4f cost %.3f" % (yaw, pitch, res.fun), flush=True)
    # Stage 2: location.  Start from the sphere's silhouette in the screenshot (centre about (430, 392), radius about 335 px of the
    # 1024x768 window whose pin-hole constant is 640): angular radius -> distance, centre -> direction.
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    fwd = np.array([cp * cy, cp * sy, sp]); up = np.array([-sp * cy, -sp * sy, cp]); right = np.cross(fwd, up)
    px, py, pr = 430.0, 392.0, 335.0
    u = fwd * 640.0 + right * (px - 512.0) + up * (384.0 - py)
    u /= np.linalg.norm(u)
    dist = 3.0 / math.sin(math.atan(pr / 640.0))
    loc0 = np.array([10.0, 0.0, 0.0]) - u * dist
    best = (float(loc0[0]), float(loc0[1]), float(loc0[2]), yaw, pitch)
    for (rw, rh, blur, iters) in ((256, 192, 1.5, 300),) if not quick else ((128, 96, 2.0, 100),):
        ref_small = reduced(ref_full, rw, rh, blur)

        def cost(p):
            return float(np.abs(reduced(tf(render(osc, p, rw, rh)), rw, rh, blur) - ref_small).mean())
        res = minimize(cost, np.array(best), method="Nelder-Mead", options={"maxiter": iters, "xatol": 1e-4, "fatol": 1e-4})
        best = tuple(float(v) for v in res.x)
        print("stage 2 at %dx%d: cost %.3f pose %s" % (rw, rh, res.fun, ["%.4f" % v for v in best]), flush=True)
    raw = render(osc, best, W, H)
    ours = tf(raw)
    diff = np.abs(ours.astype(np.int16) - ref_full.astype(np.int16))
    # flat background tiles: pixels whose 9x9 neighbourhood is uniform in BOTH images (away from grid lines, glyphs and the sphere)
    def flat_mask(img):
        f = np.asarray(Image.fromarray(img).filter(ImageFilter.MaxFilter(9)), dtype=np.int16) - np.asarray(Image.fromarray(img).filter(ImageFilter.MinFilter(9)), dtype=np.int16)
        return f.max(axis=2) <= 6
    mask = flat_mask(ours) & flat_mask(ref_full)
    flat = diff[mask]
    out = {"reference_image": "preview/preview_3d_fresnel_sphere.png (1024x768 RGBA, the real binary's window)",
           "scene": "scenes/3d_fresnel.json", "max_depth": DEPTH,
           "fitted_pose": {"location": best[:3], "yaw_rad": best[3], "pitch_rad": best[4]},
           "transfer_between_rawimage_and_screenshot": transfer, "transfer_candidates_coarse_cost": transfer_costs,
           "whole_frame": {"mean_abs_diff": float(diff.mean()), "median_abs_diff": float(np.median(diff)),
                           "share_within_8": float((diff <= 8).mean()), "share_within_24": float((diff <= 24).mean())},
           "flat_background_tiles": {"pixels": int(mask.sum()), "mean_abs_diff": float(flat.mean()), "p95_abs_diff": float(np.percentile(flat, 95)),
                                     "mean_signed_diff_rgb": [float((ours.astype(np.int16) - ref_full.astype(np.int16))[mask][:, c].mean()) for c in range(3)]},
           "note": "the pose is fitted (5 parameters), so sub-pixel misalignment remains: edges and glyphs differ, flat regions do not. "
                   "A gamma-encoded to_pixel (palette's sRGB path) would shift flat mid-tones by 30-60 levels; the observed shift is the figure above."}
    out["background_scan"] = background_scan(ref_full)
    out["reading"] = ("Robust at picture level: (1) the screenshot shows the RawImage2d bytes through an sRGB DECODE (the window's GL path), not as they are "
                      "and not gamma-encoded: Rgb::to_pixel of the trace path applies no gamma, as restated; (2) of the four flips of (u, v) the restated "
                      "uv_sphere_3 orientation fits best; (3) the glass sphere's silhouette, the refracted grid and the mirrored glyphs line up at the fitted "
                      "pose.  Not reproduced: the background's angular scale -- the screenshot shows the texture tiles about twice as large as fov 90 "
                      "(d3/entity/camera.rs:49,176-180) gives; a free field of view does not settle on one value either, so the picture predates the "
                      "shipped camera code or was taken with other settings.  It therefore pins look and conventions, not numbers.")
    for name, f in TRANSFERS.items():      # the same flat pixels under the other transfer functions, for scale
        out["flat_background_tiles"]["mean_abs_diff_" + name] = float(np.abs(f(raw)[mask].astype(np.int16) - ref_full[mask].astype(np.int16)).mean())
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "preview_3d_fresnel_fit.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    if "--png" in sys.argv:
        side = np.concatenate([ref_full, ours, np.clip(diff * 4, 0, 255).astype(np.uint8)], axis=1)
        Image.fromarray(side).save(sys.argv[sys.argv.index("--png") + 1])


if __name__ == "__main__":
    main()
