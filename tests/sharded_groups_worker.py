"""Worker of tests/test_gpu_parity.py::test_sharded_scene_render_frames: a one-rank RCCL process group, a
ShardedScene, groups of frames with a moving camera, per-frame renders in between, and bins that overflow inside a
group (the collective retry of ShardedScene.sync).  Prints OK or raises."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import tiny_renderer_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tiny_renderer_amd.sharded import ShardedScene  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
mesh, texs = T.synthetic_scene()
W, Hh = 256, 256


def frames(n, c0):
    p = np.zeros((n, 12), np.float32)
    for i in range(n):
        p[i, 0:3] = H.light(0.2 + 0.1 * i)
        p[i, 3:6], p[i, 6:9], p[i, 9:12] = H.camera(c0 + 0.3 * i)
    return p


def oracle(pipe, q):
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    cpu.clear(), cpu.set_light_direction(q[0:3]), cpu.set_camera(q[3:6], q[6:9], q[9:12])
    assert cpu.render() == 0
    return cpu.get_frame_buffer()


for pipe, cap in (("phong", 0), ("shadow", 0), ("phong", 64)):   # cap 64: 5 022 polygons overflow the bins at 256^2
    s = ShardedScene(W, Hh, mesh, texs, pipe, frames_per_launch=4, bin_capacity=cap)
    p = frames(9, 0.0)
    s.render_frames(p)                                   # groups of 4, 4, 1
    assert np.array_equal(s.get_frame_buffer(), oracle(pipe, p[-1])), (pipe, cap, "after 9 frames")
    s.clear(), s.set_light_direction(p[3, 0:3]), s.set_camera(p[3, 3:6], p[3, 6:9], p[3, 9:12]), s.render()
    assert np.array_equal(s.get_frame_buffer(), oracle(pipe, p[3])), (pipe, cap, "per-frame render after a group call")
    p2 = frames(6, 2.0)
    s.render_frames(p2)
    assert np.array_equal(s.get_frame_buffer(), oracle(pipe, p2[-1])), (pipe, cap, "second call")
    s.close()
dist.barrier()
dist.destroy_process_group()
print("OK")
