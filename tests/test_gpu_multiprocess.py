"""The per-process multi-GPU flow of bench.py --gpus N with the PRODUCT's strip path: 2 processes share the one device of this box,
each traces its 8-row strips on the GPU through the C ABI (eu_frame.strip_*), ONE gather (gloo here; RCCL on a multi-GPU node)
brings the packed strips to rank 0, which restores row order.  The result must equal the oracle's frame."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.interpreter_only]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, depth, specialize, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from euclider_amd import Parser, partition
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_room.json")).configure(specialize=specialize)
    env.camera.max_depth = depth
    img = env.render((W, H), strips=(rank, world))          # the HIP strip path
    rays = img.stats["rays"]
    env.close()
    max_rows = max(partition.local_rows(H, r, world) for r in range(world))
    local = np.zeros((max_rows, W, 3), dtype=np.uint8)
    local[:img.data.shape[0]] = img.data
    t = torch.from_numpy(local)
    gathered = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gathered, dst=0)            # the single gather
    tot = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(tot)
    if rank == 0:
        perm = torch.tensor(partition.gather_permutation(H, world, max_rows))
        full = torch.index_select(torch.cat(gathered, 0), 0, perm).numpy()
        np.save(out_path, full)
        np.save(out_path + ".rays.npy", tot.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("specialize", ["off", "sync"])
def test_two_processes_one_device_strip_gather(tmp_path, specialize):
    import torch.multiprocessing as mp
    from oracle.scene_loader import load_scene_file
    W, H, depth, world = 320, 180, 6, 2
    out = str(tmp_path / "full.npy")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, depth, specialize, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    full = np.load(out)
    ref, _, st = load_scene_file(os.path.join(ROOT, "scenes", "3d_room.json")).render(W, H, max_depth=depth)
    assert np.array_equal(full, ref)
    assert int(np.load(out + ".rays.npy")[0]) == st["rays"]
