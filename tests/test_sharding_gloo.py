"""N>1 path on the CPU (gloo, world size 2 and 3): every rank asks the PRODUCT's partition function
(tr_band_rows, through the C ABI) for its band, renders ONLY that band -- with the oracle, whose
clamp rectangle (scene.rs:236-239) is cut to the band exactly as k_setup cuts it on the GPU -- into
its slice of a full-frame tensor, and the bands are exchanged with the in-place
all_gather_into_tensor bench.py issues over RCCL.  The gathered frame must equal the single-rank
frame byte for byte: overlapping, gapped, unequal or mis-ordered bands all fail here, and so does
a double-buffering scheme that hands a slot back too early (two frames in flight, moving camera).

The GPU side of the same contract (bands rendered by the HIP path, a caller's stream, two frame
tensors) is tests/test_gpu_parity.py::test_band_shards_reassemble and
::test_caller_stream_consumes_frames_without_sync."""
import os
import socket
import sys

import numpy as np
import pytest

from tests import helpers as H


def _worker(rank, world, port, W, Hh, pipe, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, H.REPO)
    import tiny_renderer_amd as T
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    try:
        assert dist.get_world_size() == world
        mesh, texs = T.synthetic_scene(n_lat=10, n_lon=20, tex_size=128)
        row0, row1 = T.band_rows(Hh, world, rank)            # the product's partition, via the C ABI
        n = (row1 - row0) * W * 3
        assert n * world == Hh * W * 3, "in-place all-gather needs equal bands"
        full = O.Scene(W, Hh, mesh, texs, pipe)              # what one GPU would render
        mine = O.Scene(W, Hh, mesh, texs, pipe)              # this rank: its band only
        mine.set_output_band(row0, row1)
        fbs = [torch.zeros(Hh * W * 3, dtype=torch.uint8) for _ in range(2)]   # double-buffered, as in bench.py
        works = [None, None]
        for f in range(4):
            b = f % 2
            if works[b] is not None:
                works[b].wait()                              # slot free again (bench: render waits for `gathered[b]`)
                ok = ok and np.array_equal(fbs[b].numpy().reshape(Hh, W, 3), expect[b])
            for s in (full, mine):
                s.clear()
                s.set_light_direction(H.light(0.2 + 0.1 * f))
                s.set_camera(*H.camera(0.4 * f))
                assert s.render() == 0
            band = mine.get_frame_buffer()[row0:row1]
            # rows outside the band stay cleared: the shard really rendered its band only
            outside = np.delete(mine.get_frame_buffer(), np.s_[row0:row1], axis=0)
            ok = ok and not outside.any()
            chunk = fbs[b][rank * n:(rank + 1) * n]
            chunk.copy_(torch.from_numpy(np.ascontiguousarray(band).reshape(-1)))
            if f == 0:
                expect = [None, None]
            expect[b] = full.get_frame_buffer()
            works[b] = dist.all_gather_into_tensor(fbs[b], chunk, async_op=True)   # exchange of f under the render of f+1
        for b in range(2):
            works[b].wait()
            ok = ok and np.array_equal(fbs[b].numpy().reshape(Hh, W, 3), expect[b])
        dist.barrier()
    finally:
        dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world,pipe", [(2, "phong"), (2, "shadow"), (3, "phong")])
def test_band_allgather_ranks(built, world, pipe):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 96, 66 if world == 3 else 64, pipe, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_bench_refuses_mismatched_world(built):
    """`bench.py --gpus N` inside a launcher whose WORLD_SIZE is not N must not run (it used to print a
    1-GPU number); without a launcher and with N > 1 it starts the ranks itself (checked on the GPU box)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)


def _sparse_worker(rank, world, port, W, Hh, q):
    """The sparse exchange's PROTOCOL on the CPU (tr_exchange_all_gather_tiles / k_push_tiles, csrc): a rank sends a tile of
    its band only if it is not the cleared colour, or if its record of the PEER's copy says the peer still holds
    something else there -- and then it sends zeros; the record follows what was sent.  Tiles are the product's
    128 x 16 pixels, bands the product's tr_band_rows.  Frames where the model moves out of tiles it covered before are
    the interesting ones: a tile that became empty must be zeroed on the peer exactly once."""
    import torch.distributed as dist
    sys.path.insert(0, H.REPO)
    import tiny_renderer_amd as T
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok, sent_tiles, dense_tiles = True, 0, 0
    try:
        mesh, texs = T.synthetic_scene(n_lat=10, n_lon=20, tex_size=128)
        mesh = dict(mesh, pos=(mesh["pos"] * np.float32(0.45)).astype(np.float32))   # a model that leaves most tiles empty
        row0, row1 = T.band_rows(Hh, world, rank)
        full = O.Scene(W, Hh, mesh, texs, "phong")
        mine = O.Scene(W, Hh, mesh, texs, "phong")
        mine.set_output_band(row0, row1)
        frame = np.zeros((Hh, W, 3), np.uint8)                       # this rank's copy of the whole frame
        TW, TH = 128, 16
        # tiles are rows of 16 pixels counted from the BOTTOM of the frame (y up), clipped to the band
        tiles = [(x0, max(row0, Hh - (ty + 1) * TH), min(row1, Hh - ty * TH))
                 for ty in range((Hh + TH - 1) // TH) for x0 in range(0, W, TW)]
        tiles = [(x0, r0, r1) for (x0, r0, r1) in tiles if r0 < r1]
        remote_clean = {p: [False] * len(tiles) for p in range(world) if p != rank}   # "the peer's copy holds zeros"
        for f, ca in enumerate([0.0, 0.9, 1.8, 0.9, 0.0, 2.7]):
            for s_ in (full, mine):
                s_.clear()
                s_.set_light_direction(H.light(0.3))
                s_.set_camera(*H.camera(ca))
                assert s_.render() == 0
            band = mine.get_frame_buffer()
            frame[row0:row1] = band[row0:row1]
            out = {p: [] for p in remote_clean}
            for t, (x0, r0, r1) in enumerate(tiles):
                px = band[r0:r1, x0:x0 + TW]
                zeros = not px.any()
                for p in remote_clean:
                    dense_tiles += 1
                    if zeros and remote_clean[p][t]:
                        continue                                       # zeros here, zeros there: stays home
                    out[p].append((t, None if zeros else px.copy()))
                    remote_clean[p][t] = zeros
                    sent_tiles += 1
            got = [None] * world
            dist.all_gather_object(got, (rank, tiles, out))
            for (src, src_tiles, src_out) in got:
                if src == rank:
                    continue
                for (t, px) in src_out[rank]:
                    x0, r0, r1 = src_tiles[t]
                    frame[r0:r1, x0:x0 + TW] = 0 if px is None else px
            ok = ok and np.array_equal(frame, full.get_frame_buffer())
        dist.barrier()
    finally:
        dist.destroy_process_group()
    q.put((rank, bool(ok), sent_tiles, dense_tiles))


def test_sparse_tile_exchange_protocol(built):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, 512, 256, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[:2] for r in res) == [(r, True) for r in range(world)]
    # most tiles never travel (the first frame sends every tile once: the peers' copies are unknown)
    assert all(r[2] < 0.6 * r[3] for r in res), res
