"""N > 1 path on CPU: 2 ranks (gloo) render their strips, ONE gather, rank 0 restores row order.

The GPU kernels cannot run here, so the oracle stands in as the per-rank renderer; what is under test is
the partition arithmetic (euclider_amd.partition == eu_frame_local_rows), the gather and the reorder that
bench.py uses for --gpus N.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, depth, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from euclider_amd import partition
    from oracle.scene_loader import load_scene_file
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    osc = load_scene_file(os.path.join(ROOT, "scenes", "3d_room.json"))
    rows = partition.local_rows(H, rank, world)
    max_rows = max(partition.local_rows(H, r, world) for r in range(world))
    local = np.zeros((max_rows, W, 3), dtype=np.uint8)
    for lr0 in range(0, rows, partition.STRIP):
        g0 = partition.global_row(lr0, rank, world)
        g1 = min(g0 + partition.STRIP, H)
        if g0 >= H:
            continue
        rgb, _, _ = osc.render(W, H, max_depth=depth, threads=2, rows=(g0, g1))
        local[lr0:lr0 + (g1 - g0)] = rgb
    t = torch.from_numpy(local)
    gathered = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gathered, dst=0)            # the single gather
    if rank == 0:
        perm = torch.tensor(partition.gather_permutation(H, world, max_rows))
        full = torch.index_select(torch.cat(gathered, 0), 0, perm).numpy()
        np.save(out_path, full)
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [48, 52])        # 52: last strip is partial (padding rows)
def test_two_rank_strip_gather_matches_full_frame(tmp_path, H):
    import torch.multiprocessing as mp
    from oracle.scene_loader import load_scene_file
    W, depth, world = 64, 4, 2
    out = str(tmp_path / "full.npy")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, depth, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    full = np.load(out)
    ref, _, _ = load_scene_file(os.path.join(ROOT, "scenes", "3d_room.json")).render(W, H, max_depth=depth, threads=2)
    assert np.array_equal(full, ref)


def test_partition_matches_c_abi():
    import ctypes as C
    from euclider_amd import _capi, partition
    L = _capi.lib()
    for H in (1, 7, 8, 9, 64, 1080, 1528, 4320):
        for world in (1, 2, 3, 4, 8):
            seen = set()
            for rank in range(world):
                fr = _capi.Frame(16, H, 0, H, 0, 0, world if world > 1 else 0, rank, 0)
                rows = L.eu_frame_local_rows(C.byref(fr))
                assert rows == partition.local_rows(H, rank, world)
                for lr in range(rows):
                    g = partition.global_row(lr, rank, world)
                    if g < H:
                        assert g not in seen
                        seen.add(g)
            assert seen == set(range(H))
            mr = max(partition.local_rows(H, r, world) for r in range(world))
            perm = partition.gather_permutation(H, world, mr)
            assert len(set(perm)) == H
