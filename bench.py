#!/usr/bin/env python3
"""bench.py -- Mray/s of the per-pixel trace loop on MI355X (BASELINE.json metric).

One "step" = one Environment::render pass: every pixel of the frame is traced by the HIP
wavefront pipeline (intersect + shade per generation, resolve; scene, textures and the output frame resident in HBM; by default with
kernels specialised for the scene, compiled when the renderer is created) and the
RGBA8 result is packed to the reference's RGB8 RawImage2d layout, also in HBM.  Steps are issued round-robin to `--frames-in-flight`
renderers (default 12), each with a stream and buffers of its own, so that consecutive frames overlap on the device the way the frame
loop of the C ABI (eu_sequence_*) overlaps them; the timed region still holds exactly K whole frames between two device
synchronisations, and `config.one_frame_alone` carries the time of a single frame with nothing else in flight.  The HIP runtime
maps streams onto GPU_MAX_HW_QUEUES hardware queues (its default: 4); this script asks for 8 unless the environment already says
otherwise, and reports the value in `config.gpu_max_hw_queues` (round 4's final kernels on config 2, queues x frames in flight, the
timed region alone in the process: 8 x 8 8.4-8.5 Gray/s, 8 x 12 9.1, 12 x 12 9.3-9.4, 12 x 16 9.3, 16 x 12 9.2, 24 x 24 7.6:
profiles/r04_ab/frames_in_flight_final*.txt.  Twelve queues lose most of that again once the process has created and destroyed other
renderers before -- the whole record: 8.6 Gray/s, and one-frame-at-a-time numbers up to 40 % worse -- so the default is 12 frames on 8
queues: 9.0 Gray/s in the whole record; INTEGRATION.md tells a host how to set it).

  N = 1 : scenes/3d_room.json, 1920x1080, max depth 8 (BASELINE.json configs[1]).
  N > 1 : the frame grows with N (weak scaling: 1920x1080 pixels per GPU, aspect kept, same
          camera/fov), is cut into 8-row strips dealt round-robin over the ranks (row tiles),
          each rank traces its strips and packs them to RGB8, ONE RCCL gather collects them on
          rank 0, which restores row order (= the reference's RawImage2d).

Prints one JSON line (rank 0).  `value` = rays of all ranks / max-over-ranks wall time.  At N = 1 the line also carries
`cpu_baseline` (the oracle timed on this host), `parity` (the frame the timed run left in HBM compared byte for byte with the
oracle's frame) and `other_configs` (BASELINE configs 3 and 4, 4d_cylinders and the 8K frame on one GPU, a few steps each);
at N > 1 also `config5` (the 8K frame over the same ranks, strong scaling).  `python bench.py --gpus N` without a launcher
starts the N ranks itself (torch.distributed.run as a child process); with a launcher, a WORLD_SIZE that differs from
--gpus is an error.
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before anything initialises HIP (see the docstring)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks (default: WORLD_SIZE when a launcher started us, else 1)")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="3d_room.json")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fixed-frame", action="store_true", help="keep width x height for every N (strong scaling, e.g. BASELINE config 5: 7680x4320 on 8 GPUs)")
    ap.add_argument("--animate", action="store_true",
                    help="N=1 only, not the headline: fly-through (Camera::update through trace_path each frame) with the frame "
                         "sequence driver, images read back to pinned host memory -- the PCIe-inclusive rate")
    ap.add_argument("--slots", type=int, default=4, help="--animate: frames in flight")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="0 = whole frame")
    ap.add_argument("--low-precision", action="store_true",
                    help="F = f32: the reference's `low_precision` cargo feature (libeuclider_amd_f32.so against libeo_oracle_f32.so). "
                         "A separate mode, never the headline: narrower than the reference's default arithmetic")
    ap.add_argument("--no-other-configs", action="store_true", help="N=1: skip the short runs of BASELINE configs 3, 4 and the 8K frame")
    ap.add_argument("--specialize", choices=["sync", "off"], default="sync",
                    help="sync: trace kernels specialised for the scene, compiled (hiprtc) when the renderer is created, outside the timed "
                         "region; off: the ahead-of-time kernels that interpret the flat scene")
    ap.add_argument("--streams", type=int, default=0, help="band pipelines in flight per frame (0 = the library's default)")
    ap.add_argument("--jit-flags", default=None, help="extra hiprtc flags for the specialised kernels (tuning experiments)")
    ap.add_argument("--renderer-flags", type=int, default=0, help="eu_renderer_opts.flags")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames traced concurrently, each by a renderer of its own on a stream of its own, one band stream each (0 = 12, on 8 hardware queues: the "
                         "docstring has the sweep; round 3: 5 / 8 / 12 in flight = 8.5 / 8.9 / 8.8 Gray/s with 8 hardware queues, 8.0 / 8.2 / 8.4 with the runtime's 4; "
                         "1 = one frame at a time on two band streams)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps steps between two synchronisations) is run this many times; value and ms_per_step are the MEDIAN region, "
                         "config.timed_regions carries min / median / max")
    ap.add_argument("--no-alone", action="store_true", help="skip the one-frame-at-a-time measurement (profiling passes: only the timed region's launches)")
    ap.add_argument("--abi-child", type=int, default=0,
                    help="internal: ONE process drives this many GPUs through the C ABI (eu_render_multi) on the 8K frame and prints a JSON object")
    return ap.parse_args()


def frame_dims(w, h, n):
    """Weak scaling: n times the pixels, aspect kept, multiples of 8."""
    if n == 1:
        return w, h
    s = math.sqrt(n)
    return 8 * int(round(w * s / 8.0)), 8 * int(round(h * s / 8.0))


def host_threads():
    """Threads the CPU baseline may use: the cgroup CPU quota if there is one, else the affinity mask, capped at 64
    (a GPU box is shared: one GPU's share of the host is 16-32 cores)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(scene_path, w, h, depth, sample_rows, variant=""):
    """The oracle (CPU restatement, kind "port") timed on this host's cores -- a reported baseline.  Also returns the
    oracle's frame (or row band) so that the GPU frame of the timed run can be compared with it."""
    from oracle.scene_loader import load_scene_file
    threads = host_threads()
    osc = load_scene_file(scene_path, variant=variant)
    runs = 3
    if sample_rows and sample_rows < h:
        r0 = (h - sample_rows) // 2
        rows = (r0, r0 + sample_rows)
        sample = "%s %dx%d depth %d, rows %d..%d (centre band), best of %d runs after 1 warm-up" % (os.path.basename(scene_path), w, h, depth, rows[0], rows[1], runs)
    else:
        rows = None
        sample = "%s %dx%d depth %d, whole frame, best of %d runs after 1 warm-up" % (os.path.basename(scene_path), w, h, depth, runs)
    t0 = time.perf_counter()
    orgb, _, st = osc.render(w, h, max_depth=depth, threads=threads, rows=rows)     # warm-up (also bounds the cost)
    first = time.perf_counter() - t0
    dt = first
    if first < 10.0:
        for _ in range(runs):
            t0 = time.perf_counter()
            _, _, st = osc.render(w, h, max_depth=depth, threads=threads, rows=rows)
            dt = min(dt, time.perf_counter() - t0)
    else:
        sample = sample.replace("best of %d runs after 1 warm-up" % runs, "1 run")
    return {"value": st["rays"] / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "host_threads_total": os.cpu_count(), "sample": sample, "seconds": round(dt, 3), "rays": st["rays"]}, orgb, rows, st


def load_pmc(workload_key):
    """Per-launch PMC figures of this workload from the committed rocprofv3 passes (profiles/traffic.json), if any."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(workload_key, {})
    except Exception:
        return {}


_GOLDEN_FRAMES = None


def golden_frames():
    global _GOLDEN_FRAMES
    if _GOLDEN_FRAMES is None:
        try:
            with open(os.path.join(ROOT, "tests", "golden", "frame_sha.json")) as f:
                _GOLDEN_FRAMES = json.load(f)
        except Exception:
            _GOLDEN_FRAMES = {}
    return _GOLDEN_FRAMES


def load_traffic(workload_key):
    return load_pmc(workload_key).get("hbm_bytes_per_launch")


# MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz; a wave64 f64 VALU instruction occupies its SIMD for 4 cycles (16 lanes per
# clock; f64 FMA is full rate: 256*4*16*2*2.4e9 = 78.6 TFLOP/s), a 32-bit one for 2 cycles.
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9
F64_PEAK_TFLOPS = 78.6


def valu_figure(workload_key, kernel_ms):
    """Secondary roofline (the path is f64-VALU bound, not HBM bound): wave-level VALU instructions of one frame
    pipeline over the kernel time, against the VALU issue peak.  Everything comes from THIS round's profiles of the kernels that
    ran (profiles/r04_isa_mix.json, made by tools/isa_mix_r04.py from profiles/r04_<workload>_pmc.json and the disassembly of the
    specialised code objects): the instruction count per frame is the PMC passes' SQ_INSTS_VALU x launches per frame, and the peak
    is priced with the same kernels' instruction mix -- 4 SIMD cycles for an f64 VALU instruction, 2 for any other, each kernel's
    static f64 share weighted with its dynamic VALU count.  A workload without such a profile gets no figure."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r04_isa_mix.json")))["workloads"].get(workload_key)
    except Exception:
        rec = None
    if not rec or "cycles_per_valu_instruction" not in rec:
        return None
    n = rec["valu_wave_insts_per_frame"]
    cycles = rec["cycles_per_valu_instruction"]
    peak = SIMD_CYCLES_PER_S / cycles / 1e9
    ach = n / (kernel_ms * 1e-3) / 1e9
    return {"achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak, "cycles_per_instruction": cycles,
            "mix": "profiles/r04_isa_mix.json (this round's specialised kernels: static f64 share per kernel x dynamic VALU count per kernel)",
            "lane_utilisation": load_pmc(workload_key).get("valu_lane_utilisation"), "wave_insts_per_launch": n}


def flops_figure(workload_key, rays, kernel_ms):
    """Compute roofline: f64 operations of the reference ALGORITHM per ray, counted by the oracle's instrumented
    build (profiles/r03_oracle_flops.json; add/sub/mul, div, sqrt and transcendental calls count 1 each), times this frame's
    rays, over the live kernel time, against the f64 vector peak.  The kernels execute several instructions per division,
    square root and transcendental call, so their own f64 instruction rate is higher than this figure (see `valu`)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r03_oracle_flops.json")))["workloads"].get(workload_key)
    except Exception:
        rec = None
    if not rec:
        return None
    per_ray = rec["per_ray_all"]
    ach = per_ray * rays / (kernel_ms * 1e-3) / 1e12
    return {"per_ray": per_ray, "per_ray_breakdown": rec["per_ray"], "achieved_TFLOPs": ach, "peak": F64_PEAK_TFLOPS, "frac": ach / F64_PEAK_TFLOPS,
            "counted_by": "oracle/libeo_oracle_flops.so on the same workload (tools/oracle_variants_report.py)"}


def animate(args, env, scene_path):
    """Fly-through: every step moves the camera (W held, slow yaw: Camera::update -> eu_trace_path on the GPU), submits the
    frame to the sequence driver and takes the oldest finished image from pinned host memory."""
    from euclider_amd import FrameSequence, SimulationContext
    W, H = args.width, args.height
    ctx = SimulationContext(resolution=1, pressed_keys=["W"], delta_mouse=(1, 0))
    rays = 0
    with FrameSequence(env, (W, H), slots=args.slots) as seq:
        def step(k):
            env.update(0.008, ctx, speed=2.0)
            if seq.in_flight == args.slots:
                img = seq.next(copy=False)
                return img.stats["rays"]
            seq.submit((W, H), time=k * 0.008)
            return 0
        for k in range(args.warmup):
            step(k)
        while seq.in_flight:
            seq.next(copy=False)
        t0 = time.perf_counter()
        taken = 0
        k = 0
        while taken < args.steps:
            if seq.in_flight == args.slots or k >= args.steps:
                rays += seq.next(copy=False).stats["rays"]
                taken += 1
            if k < args.steps:
                env.update(0.008, ctx, speed=2.0)
                seq.submit((W, H), time=(args.warmup + k) * 0.008)
                k += 1
        elapsed = time.perf_counter() - t0
    out = {"metric": "Mray/s, fly-through incl. Camera::update and read-back to pinned host memory (not the headline)",
           "value": rays / elapsed / 1e6, "unit": "Mray/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "fps": args.steps / elapsed, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%s %dx%d depth %d fly-through, %d frames in flight" % (args.scene, W, H, args.max_depth, args.slots),
                      "camera_end": list(env.camera.location)[:env.dim], "readback_bytes_per_frame": W * H * 3}}
    print(json.dumps(out), flush=True)


def abi_child(args):
    """BASELINE config 5 through the C ABI: one process, eu_multi over N devices (what a Rust `impl Environment` would call).  Runs in a
    process of its own, started by rank 0 of the per-process run after its own measurements."""
    from euclider_amd import Parser
    n = args.abi_child
    env = Parser().parse_file(os.path.join(ROOT, "scenes", args.scene)).configure(specialize=args.specialize)
    env.camera.max_depth = args.max_depth
    W, H = 7680, 4320
    img = env.render_multi((W, H), list(range(n)))      # warm-up: renderers, kernels, buffers
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        img = env.render_multi((W, H), list(range(n)))
    dt = (time.perf_counter() - t0) / steps
    # the frame-loop form: two frames in flight (eu_render_multi_begin / _end), frame k's gather behind frame k + 1's trace
    devs = list(range(n))
    env.render_multi_begin((W, H), devs)
    t0 = time.perf_counter()
    for _ in range(steps):
        env.render_multi_begin((W, H), devs)
        img2 = env.render_multi_end(devs)
    dt2 = (time.perf_counter() - t0) / steps
    last = env.render_multi_end(devs)
    same = bool((img2.data == img.data).all() and (last.data == img.data).all())
    print(json.dumps({"workload": "%s %dx%d depth %d, ONE process, eu_render_multi over %d devices (host image included)" % (args.scene, W, H, args.max_depth, n),
                      "value": img.stats["rays"] / dt / 1e6, "unit": "Mray/s", "ms_per_step": dt * 1e3, "steps": steps, "rays_per_frame": int(img.stats["rays"]),
                      "note": "synchronous call: traces, gathers over xGMI (peer copies), restores row order and copies the 99.5 MB image to the host",
                      "two_frames_in_flight": {"value": img.stats["rays"] / dt2 / 1e6, "ms_per_step": dt2 * 1e3, "frames_equal": same,
                                               "note": "eu_render_multi_begin / _end: pack, peer copies, row restore and read-back of frame k overlap the trace of frame k + 1"}}), flush=True)
    env.close()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run (before anything here
    has touched the GPU) and leave with its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


OTHER_CONFIGS = (("3d_hallways.json", 1920, 1080, 12, 40, False), ("4d_frame.json", 1920, 1080, 8, 40, False),
                 ("4d_cylinders.json", 1920, 1080, 8, 32, False), ("3d_room.json", 7680, 4320, 8, 6, False),
                 ("3d_room.json", 1920, 1080, 10, 40, False), ("3d_room.json", 1920, 1080, 8, 40, True))


def other_configs_alone(torch, dev, Parser, args):
    """ONE frame at a time for each of the other configurations, on a renderer with the library's defaults (the reference's call shape) -- all of
    them measured before any of the pipelined runs' renderers and streams exist: which hardware queues a renderer's band streams get depends on
    what the process has created before (a record of this round had these numbers 20-50 % apart between two runs that differed in how many
    renderers an EARLIER configuration had kept alive)."""
    res = []
    for scene, W, H, depth, steps, lp in OTHER_CONFIGS:
        ea = Parser(low_precision=lp).parse_file(os.path.join(ROOT, "scenes", scene))
        ea.configure(specialize=args.specialize, streams=args.streams, jit_flags=args.jit_flags, flags=args.renderer_flags)
        ea.camera.max_depth = depth
        frame0 = ea.frame(W, H, time=0.0, rows=(0, H))
        rgba0 = torch.zeros((H, W), dtype=torch.int32, device=dev)
        rgb0 = torch.empty((H * W * 3 + 16,), dtype=torch.uint8, device=dev)
        st0 = torch.cuda.current_stream(dev)
        raw = st0.cuda_stream
        ts = []
        for k in range(2 + (4 if W * H > (4 << 20) else 9)):
            t1 = time.perf_counter()
            ea.render_device(frame0, rgba0.data_ptr(), None, raw, device=dev.index)
            ea.pack_rgb_device(rgba0.data_ptr(), rgb0.data_ptr(), H * W, raw, device=dev.index)
            st0.synchronize()
            if k >= 2:
                ts.append((time.perf_counter() - t1) * 1e3)
        ts.sort()
        res.append((ts, ea.kernel_ms_history(4, device=dev.index)))
        ea.close()
        del rgba0, rgb0
    return res


def other_configs(torch, dev, Parser, args, in_flight, alone_results):
    """BASELINE.json configs 3 and 4, the extra 4-D scene, the 8K frame on one GPU, the reference's own depth and the f32 build: a few
    steps each, so that the driver's record carries them too.  Same timing rule as the headline (device-resident, synchronised on both
    sides, the same number of frames in flight; two for the 8K frame, whose bands are 4 Mpixel each)."""
    out = []
    for (scene, W, H, depth, steps, lp), (ts, alone_kms) in zip(OTHER_CONFIGS, alone_results):
        R = min(in_flight, 2) if W * H > (4 << 20) else in_flight
        envs = []
        while len(envs) < R:
            e = Parser(low_precision=lp).parse_file(os.path.join(ROOT, "scenes", scene))
            e.configure(specialize=args.specialize, streams=args.streams or (1 if R > 1 else 0), jit_flags=args.jit_flags, flags=args.renderer_flags)
            e.camera.max_depth = depth
            envs.append(e)
            # (round 3 kept five frames in flight for a scene whose hit stack lives in scratch -- 4d_cylinders' straight-line kernels with their
            # 1 300 spilled SGPRs: 4.9 against 4.5 Gray/s; with one loop body per run of congruent entities eight are better: 5.6 against 5.4)
        env = envs[0]
        frame = env.frame(W, H, time=0.0, rows=(0, H))
        streams = [torch.cuda.Stream(dev) for _ in range(R)]
        rgba = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(R)]
        rgb = [torch.empty((H * W * 3 + 16,), dtype=torch.uint8, device=dev) for _ in range(R)]

        def step(k):
            j = k % R
            envs[j].render_device(frame, rgba[j].data_ptr(), None, streams[j].cuda_stream, device=dev.index)
            envs[j].pack_rgb_device(rgba[j].data_ptr(), rgb[j].data_ptr(), H * W, streams[j].cuda_stream, device=dev.index)
        for k in range(2 * R):
            step(k)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        st = env.stats(device=dev.index)
        agree = all(torch.equal(rgb[0][:H * W * 3], x[:H * W * 3]) for x in rgb[1:])
        # the frame this run left in HBM against the ORACLE's frame of the same workload, by checksum (tests/golden/frame_sha.json, made on the CPU
        # by tools/make_frame_sha.py): the comparison costs no oracle time here
        import hashlib
        key = "%s %dx%d depth %d%s" % (scene, W, H, depth, " f32" if lp else "")
        gold = golden_frames().get(key)
        sha = hashlib.sha256(rgb[0][:H * W * 3].cpu().numpy().tobytes()).hexdigest()
        vs_oracle = None if gold is None else {"frame_equal": sha == gold["sha256"], "rays_equal": int(st["rays"]) == gold["rays"], "bytes_compared": gold["bytes"]}
        kernel_ms = dt * 1e3      # a frame's share of the device (frames overlap; see the headline's roofline)
        alg = 4.0 * W * H + 16.0 * st["bg_samples"] + env.info.flat_bytes
        ach = alg / (kernel_ms * 1e-3) / 1e9
        out.append({"workload": "%s %dx%d depth %d%s" % (scene, W, H, depth, ", low_precision (F = f32, its own rays and pixels)" if lp else ""),
                    "dtype": "f32" if lp else "f64", "value": st["rays"] / dt / 1e6, "unit": "Mray/s",
                    "ms_per_step": dt * 1e3, "steps": steps, "frames_in_flight": R, "slots_agree": agree, "vs_oracle": vs_oracle, "rays_per_frame": int(st["rays"]),
                    "one_frame_alone": {"ms": ts[len(ts) // 2], "Mray/s": st["rays"] / ts[len(ts) // 2] / 1e3, "frames": len(ts), "kernel_ms": sum(alone_kms) / max(1, len(alone_kms))},
                    "would_panic_events": int(st["nan_pixels"] + st["errors"]),
                    "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                 "kernel_ms": kernel_ms, "algorithmic_bytes": alg, "traffic": None if lp else load_traffic("%s %dx%d depth %d" % (scene, W, H, depth)),
                                 "valu": None if lp else valu_figure("%s %dx%d depth %d" % (scene, W, H, depth), kernel_ms),
                                 "flops": None if lp else flops_figure("%s %dx%d depth %d" % (scene, W, H, depth), st["rays"], kernel_ms)},
                    "specialized": env.jit_info(device=dev.index)["active"]})
        for e in envs:
            e.close()
        del rgba, rgb
    return out


def main():
    args = parse_args()
    if args.abi_child:
        abi_child(args)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus <= 0:      # no --gpus: adapt to the launcher
        args.gpus = world
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to report a line for "
                         "a different N" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    from euclider_amd import Parser

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the trace path)")
    # EU_BENCH_SMOKE_GLOO=1: rehearsal of the N > 1 control flow on a ONE-GPU box (all ranks on cuda:0, gather staged
    # through the host with gloo).  Never used for reported numbers; the driver's multi-GPU runs use RCCL.
    smoke_gloo = os.environ.get("EU_BENCH_SMOKE_GLOO") == "1"
    if smoke_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if smoke_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    scene_path = os.path.join(ROOT, "scenes", args.scene)
    in_flight = args.frames_in_flight if args.frames_in_flight > 0 else 12
    band_streams = args.streams or (1 if in_flight > 1 else 0)      # frames in flight fill each other's kernel tails; a lone frame is cut into two bands for that

    def make_env(streams=None):
        e = Parser(low_precision=args.low_precision).parse_file(scene_path)
        e.configure(specialize=args.specialize, streams=band_streams if streams is None else streams, jit_flags=args.jit_flags, flags=args.renderer_flags)
        e.camera.max_depth = args.max_depth
        return e
    env = make_env()
    envs = [env]
    jit = env.jit_info(device=local_rank)      # (creates the renderer: a specialised one compiles or fetches its kernels here, before any timing)
    if args.specialize == "sync" and not jit["active"]:
        raise SystemExit("bench.py: the specialised kernels did not build; run with --specialize off to time the interpreter kernels")
    if args.animate:
        if world != 1:
            raise SystemExit("--animate is a one-GPU mode")
        animate(args, env, scene_path)
        env.close()
        return
    def timed_run(W, H, steps, warmup, n_slots, repeats=1):
        """K steps of the partitioned frame (this rank's strips traced, packed, gathered on rank 0, rows restored), timed
        between barriers + device synchronisation; returns max-over-ranks wall time and summed counters.  Step k uses slot k % n_slots:
        a renderer, a stream and buffers of its own, so that up to n_slots frames are in flight (the production frame loop,
        eu_sequence_*, does the same)."""
        strips = (rank, world) if world > 1 else None
        while len(envs) < n_slots:
            envs.append(make_env())
        frame = env.frame(W, H, time=0.0, rows=(0, H), strips=strips)
        local_rows = env.local_rows(frame)
        if world > 1:      # equal counts for the gather: pad to the largest rank
            t = torch.tensor([local_rows], device=torch.device("cpu") if smoke_gloo else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            max_rows = int(t.item())
        else:
            max_rows = local_rows
        perm = None
        if world > 1 and rank == 0:
            from euclider_amd.partition import gather_permutation
            perm = torch.tensor(gather_permutation(H, world, max_rows), dtype=torch.int64, device=dev)
        slots = []
        for j in range(n_slots):
            sl = {"env": envs[j], "stream": torch.cuda.Stream(dev) if n_slots > 1 else torch.cuda.current_stream(dev),
                  "rgba": torch.zeros((max_rows, W), dtype=torch.int32, device=dev),
                  "rgb_out": torch.empty((H * W * 3 + 16,), dtype=torch.uint8, device=dev) if rank == 0 else None}
            if world > 1:
                # every rank packs its own strips to RGB8 before the gather (3 bytes per pixel travel, not 4; the root only reorders rows)
                sl["rgb_local"] = torch.empty((max_rows, W * 3), dtype=torch.uint8, device=dev)
                if rank == 0:
                    sl["allbuf"] = torch.empty((world * max_rows, W * 3), dtype=torch.uint8, device=dev)   # the gather lands in place: no concatenation
                    sl["gathered"] = [sl["allbuf"][k * max_rows:(k + 1) * max_rows] for k in range(world)]
                    sl["full"] = sl["rgb_out"][:H * W * 3].view(H, W * 3)
            slots.append(sl)

        def step(k):
            sl = slots[k % n_slots]
            e, raw = sl["env"], sl["stream"].cuda_stream
            with torch.cuda.stream(sl["stream"]):
                e.render_device(frame, sl["rgba"].data_ptr(), None, raw, device=local_rank)
                if world > 1:
                    e.pack_rgb_device(sl["rgba"].data_ptr(), sl["rgb_local"].data_ptr(), max_rows * W, raw, device=local_rank)
                    if smoke_gloo:                                          # one-GPU rehearsal only (see above)
                        torch.cuda.synchronize(dev)
                        host = sl["rgb_local"].cpu()
                        hg = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                        dist.gather(host, hg, dst=0)
                        if rank == 0:
                            for q in range(world):
                                sl["gathered"][q].copy_(hg[q])
                    else:
                        dist.gather(sl["rgb_local"], sl.get("gathered"), dst=0)             # the single RCCL gather
                    if rank == 0:
                        torch.index_select(sl["allbuf"], 0, perm, out=sl["full"])       # rows back in frame order = the RawImage2d
                else:
                    e.pack_rgb_device(sl["rgba"].data_ptr(), sl["rgb_out"].data_ptr(), H * W, raw, device=local_rank)

        def sync():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        # every slot's renderer sizes its work buffers on its first frame; should the device run out of memory (eight slots of a large
        # weak-scaled frame), the run goes on with fewer frames in flight -- all ranks agree on the number -- instead of dying
        from euclider_amd.environment import EuError
        while True:
            ok = 1
            try:
                for k in range(n_slots):
                    step(k)
                torch.cuda.synchronize(dev)
            except (EuError, RuntimeError) as e:
                if n_slots == 1:
                    raise
                ok = 0
                print("bench.py: rank %d: %d frames in flight do not fit (%s); trying %d" % (rank, n_slots, str(e)[:120], max(1, n_slots // 2)), file=sys.stderr, flush=True)
            if world > 1:
                t_ok = torch.tensor([ok], device=torch.device("cpu") if smoke_gloo else dev)
                dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
                ok = int(t_ok.item())
            if ok:
                break
            n_slots = max(1, n_slots // 2)
            for sl in slots[n_slots:]:
                sl["env"].close()
            del slots[n_slots:]
            del envs[n_slots:]
            torch.cuda.empty_cache()
        for k in range(warmup):
            step(k)
        regions = []
        rdev = torch.device("cpu") if smoke_gloo else dev
        for _ in range(max(1, repeats)):           # the timed region, several times over: exactly `steps` steps between two synchronisations each
            sync()
            t0 = time.perf_counter()
            for k in range(steps):
                step(k)
            sync()
            tr = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=rdev)
            if world > 1:
                dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            regions.append(tr.item())
        elapsed = sorted(regions)[len(regions) // 2]
        st = env.stats(device=local_rank)
        kms = env.kernel_ms_history(3, device=local_rank)
        same = True
        if rank == 0 and n_slots > 1:      # every slot rendered the same frame: their images must be identical
            same = all(torch.equal(slots[0]["rgb_out"][:H * W * 3], sl["rgb_out"][:H * W * 3]) for sl in slots[1:])
        tot = torch.tensor([float(st["rays"]), float(st["bg_samples"]), float(st["nan_pixels"] + st["errors"])], dtype=torch.float64, device=rdev)
        if world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        return {"W": W, "H": H, "elapsed": elapsed, "regions": regions, "rays": tot[0].item(), "panic": tot[2].item(), "st": st, "kms": kms,
                "local_rows": local_rows, "rgb_out": slots[0]["rgb_out"], "steps": steps, "slots_agree": same, "step": step, "sync": sync, "slots": slots, "n_slots": n_slots}

    parity_failed = False
    W, H = (args.width, args.height) if args.fixed_frame else frame_dims(args.width, args.height, world)
    alone_rgb = None
    # (measured FIRST, before the renderers of the pipelined run exist: the HIP runtime deals streams to its hardware queues as they come, and with the
    # eight slots' streams alive a renderer's three band streams shared queues -- 1.60 instead of 1.28 ms in the same process)
    # ONE frame at a time -- the reference's call shape (Environment::render is synchronous: simulation.rs:86, universe/mod.rs:300-357): a
    # renderer with the library's own defaults (its band pipelines on its own streams), launch + pack + wait, frame after frame
    alone = None
    if world == 1 and not args.no_alone:
        ea = make_env(streams=args.streams)
        frame1 = ea.frame(W, H, time=0.0, rows=(0, H))
        rgba1 = torch.zeros((H, W), dtype=torch.int32, device=dev)
        rgb1 = torch.empty((H * W * 3 + 16,), dtype=torch.uint8, device=dev)
        st1 = torch.cuda.current_stream(dev)
        raw = st1.cuda_stream
        ts = []
        for k in range(3 + 15):
            t1 = time.perf_counter()
            ea.render_device(frame1, rgba1.data_ptr(), None, raw, device=local_rank)
            ea.pack_rgb_device(rgba1.data_ptr(), rgb1.data_ptr(), H * W, raw, device=local_rank)
            st1.synchronize()      # hipStreamSynchronize on the stream the frame went to (what eu_render does; hipDeviceSynchronize walks every stream of the device: +15..30 us, tools/alone_sync_forms.py)
            if k >= 3:
                ts.append((time.perf_counter() - t1) * 1e3)
        ts.sort()
        akms = ea.kernel_ms_history(8, device=local_rank)
        alone = {"ms": ts[len(ts) // 2], "ms_min": ts[0], "ms_max": ts[-1], "frames": len(ts), "Mray/s": ea.stats(device=local_rank)["rays"] / ts[len(ts) // 2] / 1e3,
                 "kernel_ms": sum(akms) / max(1, len(akms)),
                 "band_streams": args.streams or "library default",
                 "note": "launch, pack, wait (hipStreamSynchronize on the caller's stream) -- repeat: one renderer with the library's defaults, nothing else in flight (host launch time and the final wait included; kernel_ms = HIP events around the frame's pipeline)"}
        alone_rgb = rgb1[:H * W * 3].clone()
        ea.close()
        del rgba1, rgb1
    others_alone = None
    if world == 1 and not args.no_other_configs and not args.fixed_frame and args.scene == "3d_room.json" and not args.low_precision:
        others_alone = other_configs_alone(torch, dev, Parser, args)

    run = timed_run(W, H, args.steps, args.warmup, in_flight, args.repeats)
    elapsed, rays_per_step, st, kms, local_rows, rgb_out = run["elapsed"], run["rays"], run["st"], run["kms"], run["local_rows"], run["rgb_out"]
    tot = [run["rays"], 0.0, run["panic"]]
    stream = torch.cuda.current_stream(dev).cuda_stream

    if alone is not None and alone_rgb is not None and rank == 0:
        alone["same_frame"] = bool(torch.equal(alone_rgb, run["rgb_out"][:H * W * 3]))
    del alone_rgb
    cfg5 = cfg5_abi = None
    if world > 1 and not args.fixed_frame:      # BASELINE config 5 next to the weak-scaling value: the 8K frame over the same ranks (strong scaling)
        r5 = timed_run(7680, 4320, 4, 1, min(in_flight, 2))
        cfg5 = {"workload": "%s 7680x4320 depth %d, %d ranks" % (args.scene, args.max_depth, world), "scaling": "strong",
                "value": r5["rays"] * r5["steps"] / r5["elapsed"] / 1e6, "unit": "Mray/s", "ms_per_step": r5["elapsed"] / r5["steps"] * 1e3,
                "steps": r5["steps"], "rays_per_frame": int(r5["rays"])}
        del r5
        if rank == 0 and not smoke_gloo:      # the same frame through the C ABI's one-process path, in a process of its own (the other ranks wait at the next barrier)
            import subprocess
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--abi-child", str(world), "--scene", args.scene, "--max-depth", str(args.max_depth),
                                     "--specialize", args.specialize], capture_output=True, text=True, timeout=240,
                                    env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")})
                cfg5_abi = json.loads(cp.stdout.strip().split("\n")[-1]) if cp.returncode == 0 else {"error": (cp.stderr or cp.stdout)[-400:]}
            except Exception as e:      # never let the extra measurement take the line down
                cfg5_abi = {"error": repr(e)[:400]}
        if world > 1:
            dist.barrier()
    if rank == 0 and os.environ.get("EU_BENCH_DUMP"):
        import numpy as np
        np.save(os.environ["EU_BENCH_DUMP"], rgb_out[:H * W * 3].cpu().numpy().reshape(H, W, 3))
    if rank == 0:
        value = rays_per_step * args.steps / elapsed / 1e6
        # device time per frame: with frames in flight the pipelines overlap, so a frame's share of the device is the timed region / steps (HIP
        # events bracket it: torch.cuda.synchronize on both sides); `frame_alone_kernel_ms` is the HIP-event span of one frame's pipeline on its own
        kernel_ms = elapsed / args.steps * 1e3 if in_flight > 1 else sum(kms) / max(1, len(kms))
        # algorithmic bytes of ONE launch of the trace kernel on this rank (DESIGN.md "Roofline"):
        # 4 B RGBA8 store per pixel + 16 B (4 RGBA8 texels) per background sample + the flat scene once
        alg_bytes = 4.0 * local_rows * W + 16.0 * st["bg_samples"] + env.info.flat_bytes
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        workload = "%s %dx%d depth %d" % (args.scene, W, H, args.max_depth)
        out = {
            "metric": "Mray/s (primary+secondary) at 1920\u00d71080 depth-8; frac of HBM roofline",     # BASELINE.json's metric, verbatim
            "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.fixed_frame else "weak",
            "vs_baseline": None, "dtype": "f32" if args.low_precision else "f64", "data": "synthetic",
            "config": {"workload": workload + (" low_precision (F = f32: a separate mode, not the headline)" if args.low_precision else ""), "scene": args.scene, "width": W, "height": H, "max_depth": args.max_depth,
                       "rays_per_frame": int(rays_per_step), "mpixel_per_s": W * H * args.steps / elapsed / 1e6,
                       "would_panic_events": int(tot[2]),
                       "frames_in_flight": run["n_slots"], "gpu_max_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "band_streams_per_frame": band_streams or "library default",
                       "one_frame_alone": alone,
                       "timed_regions": {"repeats": len(run["regions"]), "ms_per_step_min": min(run["regions"]) / args.steps * 1e3,
                                         "ms_per_step_median": elapsed / args.steps * 1e3, "ms_per_step_max": max(run["regions"]) / args.steps * 1e3},
                       "slots_agree": run["slots_agree"],
                       "kernels": ("specialised for the scene at renderer creation (hiprtc%s, %.0f ms)" % (", code object from the cache" if jit["from_cache"] else "", jit["compile_ms"])) if jit["active"] else "ahead-of-time, interpreting the flat scene",
                       "partition": ("%d ranks, 8-row strips round-robin, 1 RCCL gather" % world) if world > 1 else "1 GPU, whole frame",
                       "background": "procedural 1024x512 UV grid (reference's universe_dim.jpg is not shipped)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": load_traffic(workload), "kernel": "%s frame pipeline (generation-0 intersect, %d fused shade+intersect launches, %d resolve launches)" % ("eu_jit_intersect0 + eu_jit_fshade0/fshade + eu_wf_resolve" if jit["active"] else "eu_wf_*", args.max_depth, args.max_depth), "kernel_ms": kernel_ms,
                         "algorithmic_bytes": alg_bytes, "valu": valu_figure(workload, kernel_ms), "flops": flops_figure(workload, rays_per_step, kernel_ms),
                         "note": "bound by chains of dependent f64 arithmetic and control flow at 3 waves per SIMD, not by HBM (DESIGN.md section 4); HBM fraction reported because BASELINE asks for it; traffic / valu: profiles/r04_*_pmc.json, profiles/r04_isa_mix.json"},
        }
        # both methods at the top level, labelled: `value` is throughput with frames in flight; the reference's own call
        # shape -- one synchronous Environment::render after the other -- is `value_one_frame_at_a_time`
        out["value_method"] = "%d frames in flight on %d renderers (throughput; inputs resident, barrier + synchronize around the timed region)" % (run["n_slots"], run["n_slots"])
        if alone is not None:
            out["value_one_frame_at_a_time"] = alone["Mray/s"]
            out["value_one_frame_at_a_time_method"] = "one renderer, library defaults, launch + pack + hipStreamSynchronize per frame, default hardware queues apart from GPU_MAX_HW_QUEUES=%s set for the pipelined run" % os.environ.get("GPU_MAX_HW_QUEUES", "4")
        if cfg5 is not None:
            out["config5"] = cfg5
        if cfg5_abi is not None:
            out["config5_abi"] = cfg5_abi
        if world == 1 and not args.no_cpu_baseline:
            import numpy as np
            cb, orgb, rows, ost = cpu_baseline(scene_path, W, H, args.max_depth, args.cpu_sample_rows, "f32" if args.low_precision else "")
            out["cpu_baseline"] = cb
            out["config"]["gpu_over_cpu"] = value / cb["value"]
            # the frame the timed run left in HBM against the oracle's frame of the same inputs (whole frame unless --cpu-sample-rows)
            gpu_rgb = rgb_out[:H * W * 3].cpu().numpy().reshape(H, W, 3)
            if rows is not None:
                gpu_rgb = gpu_rgb[rows[0]:rows[1]]
            out["parity"] = {"bytes_compared": int(orgb.size), "mismatch": int((gpu_rgb != orgb).sum()),
                             "rays_equal": (rows is None and int(st["rays"]) == int(ost["rays"])) if rows is None else None,
                             "checked_against": "oracle/ (CPU restatement), same scene, camera, frame"}
        if world == 1 and not args.no_other_configs and not args.fixed_frame and args.scene == "3d_room.json" and not args.low_precision:
            del rgb_out
            run = None
            for e in envs:      # (their streams would share hardware queues with the next renderers' band streams)
                e.close()
            del envs[:]
            out["other_configs"] = other_configs(torch, dev, Parser, args, in_flight, others_alone)
        print(json.dumps(out), flush=True)
        if out.get("parity", {}).get("mismatch") or not out["config"]["slots_agree"]:
            parity_failed = True
    for e in envs:
        e.close()
    if world > 1:
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench.py: the timed frame differs from the oracle's (see \"parity\" in the line above)")


if __name__ == "__main__":
    main()
