"""Worker of tests/test_gpu_parity.py::test_sharded_scene_two_ranks_one_gpu: one RANK of a two-rank ShardedScene whose
ranks share this box's one GPU (started twice by torch.distributed.run; gloo carries the control messages, the library's
peer transport the bands -- RCCL refuses two ranks on one device).  What runs here is the product's multi-rank path with a
real peer: band scenes, double-buffered frames, groups of frames, the collective repair of a bin overflow that ONE rank
had, an error in ONE band (every rank must raise), and on EVERY rank the assembled frame against the oracle's.
    python -m torch.distributed.run --nproc-per-node 2 ... tests/sharded_ranks_worker.py peer|peer-sparse"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import tiny_renderer_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tiny_renderer_amd.sharded import ShardedScene  # noqa: E402

def main():
    exchange = sys.argv[1] if len(sys.argv) > 1 else "peer"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world >= 2, "this worker is about a rank that HAS peers"
    mesh, texs = T.synthetic_scene()
    W, Hh = 512, 384


    def frames(n, c0):
        p = np.zeros((n, 12), np.float32)
        for i in range(n):
            p[i, 0:3] = H.light(0.2 + 0.1 * i)
            p[i, 3:6], p[i, 6:9], p[i, 9:12] = H.camera(c0 + 0.3 * i)
        return p


    def oracle(m, pipe, q):
        cpu = O.Scene(W, Hh, m, texs, pipe)
        cpu.clear(), cpu.set_light_direction(q[0:3]), cpu.set_camera(q[3:6], q[6:9], q[9:12])
        cpu.render()
        out = cpu.get_frame_buffer()
        cpu.close()
        return out


    def check(s, m, pipe, q, what):
        got = s.get_frame_buffer()    # collective; the WHOLE frame, all bands, on every rank
        want = oracle(m, pipe, q)
        assert np.array_equal(got, want), "rank %d, %s %s: %d pixels differ (%s)" % (rank, exchange, pipe, int((got != want).any(-1).sum()), what)


    # cap: records in a pass's pool.  (64, 0): only rank 0's pools are too small -- the other rank's frame is fine, and it
    # must render again all the same (ShardedScene.sync: any_rank), or the ranks fall out of step
    for pipe, caps in (("phong", (0, 0)), ("shadow", (0, 0)), ("phong", (64, 0)), ("specular", (0, 64))):
        s = ShardedScene(W, Hh, mesh, texs, pipe, exchange=exchange, frames_per_launch=4, bin_capacity=caps[rank % 2])
        p = frames(5, 0.0)
        for i in range(5):                                   # the reference's per-frame protocol, frames double-buffered
            s.clear(), s.set_light_direction(p[i, 0:3]), s.set_camera(p[i, 3:6], p[i, 6:9], p[i, 9:12]), s.render()
        check(s, mesh, pipe, p[-1], "five per-frame renders")
        p = frames(9, 0.5)
        s.render_frames(p)                                   # groups of 4, 4, 1
        check(s, mesh, pipe, p[-1], "after 9 frames in groups")
        s.clear(), s.set_light_direction(p[3, 0:3]), s.set_camera(p[3, 3:6], p[3, 6:9], p[3, 9:12]), s.render()
        check(s, mesh, pipe, p[3], "per-frame render after a group call")
        p2 = frames(6, 2.0)
        s.render_frames(p2)
        check(s, mesh, pipe, p2[-1], "second group call")
        if exchange == "peer-sparse":
            sent, dense = s.exchange_bytes_sent(), (Hh // world) * W * 3 * (world - 1) * (5 + 9 + 1 + 6)
            assert 0 < sent < dense, (sent, dense)           # tiles that are the cleared colour on both sides stayed home
        s.close()

    # An error in ONE band only: a small polygon at the top of the picture whose texture coordinates leave the image
    # (util.rs:40: the reference's get_pixel panics).  Only the rank that owns those rows sees it -- and every rank must raise.
    bad = {k: np.array(v) for k, v in mesh.items()}
    n_pos, n_tex, n_nrm = len(bad["pos"]), len(bad["tex"]), len(bad["nrm"])
    bad["pos"] = np.concatenate([bad["pos"], np.array([[-0.05, 0.93, 0.0], [0.05, 0.93, 0.0], [0.0, 0.99, 0.0]], np.float32)])
    bad["tex"] = np.concatenate([bad["tex"], np.array([[1.5, 0.5, 0.0]] * 3, np.float32)])
    bad["nrm"] = np.concatenate([bad["nrm"], np.array([[0.0, 0.0, 1.0]] * 3, np.float32)])
    tri = [n_pos, n_tex, n_nrm, n_pos + 1, n_tex + 1, n_nrm + 1, n_pos + 2, n_tex + 2, n_nrm + 2]
    bad["idx"] = np.concatenate([bad["idx"], np.array([tri], np.uint32)])
    s = ShardedScene(W, Hh, bad, texs, "phong", exchange=exchange)
    q = frames(1, 0.0)[0]
    s.clear(), s.set_light_direction(q[0:3]), s.set_camera(q[3:6], q[6:9], q[9:12]), s.render()
    try:
        s.sync()
        raise AssertionError("rank %d: no error raised" % rank)
    except T.TinyRendererError as e:
        assert e.code == -5, (rank, e.code, str(e))          # TR_E_OOB_LOOKUP, on the rank that saw it AND on the other
        assert ("another rank" in str(e)) == (rank != 0), (rank, str(e))   # (rank 0 owns the top rows)
    s.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d OK" % rank)


try:
    main()
except BaseException:
    import traceback
    # (to STDOUT, with the rank: the launcher's own traceback buries a rank's stderr)
    print("RANK %s FAILED\n%s" % (os.environ.get("RANK", "?"), traceback.format_exc()), flush=True)
    raise
