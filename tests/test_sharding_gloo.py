"""N>1 path on the CPU: two gloo ranks each produce their band of the frame (from the oracle:
no GPU here) and all-gather it exactly as bench.py does with RCCL -- checks the band
arithmetic, the in-place all_gather_into_tensor layout and rank-0 assembly."""
import os
import socket
import sys

import numpy as np
import pytest

from tests import helpers as H


def _worker(rank, world, port, W, Hh, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, H.REPO)
    import tiny_renderer_amd as T
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mesh, texs = T.synthetic_scene(n_lat=10, n_lon=20, tex_size=128)
    s = O.Scene(W, Hh, mesh, texs, "phong")
    s.clear()
    s.set_light_direction(H.light(0.2))
    s.set_camera(*H.camera(0.4))
    s.render()
    full_ref = s.get_frame_buffer()
    rows = [(r * Hh) // world for r in range(world + 1)]
    fb = torch.zeros(Hh * W * 3, dtype=torch.uint8)
    n = (rows[rank + 1] - rows[rank]) * W * 3
    chunk = fb[rank * n:(rank + 1) * n]
    chunk.copy_(torch.from_numpy(full_ref[rows[rank]:rows[rank + 1]].reshape(-1)))  # this rank's band only
    dist.all_gather_into_tensor(fb, chunk)
    ok = np.array_equal(fb.numpy().reshape(Hh, W, 3), full_ref)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_two_rank_band_allgather():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 96, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
