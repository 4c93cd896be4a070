"""Minimal frame loop for profilers: N frames of one workload, nothing else on the GPU.

    python scripts/frame_loop.py SIZE PIPELINE FRAMES [MODEL [GRID]]

Submits like bench.py: one tr_scene_render_frames call (frame groups); SUBMIT=frame in the environment
issues the reference's four calls per frame instead.  FRAMES should be a multiple of the group size
(printed) so that every launch covers the same number of frames.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pipe = sys.argv[2] if len(sys.argv) > 2 else "phong"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 32
model = sys.argv[4] if len(sys.argv) > 4 else "diablo"
grid = int(sys.argv[5]) if len(sys.argv) > 5 else 1
adir = find_assets(model)
mesh, texs = T.load_assets(adir) if adir else T.synthetic_scene()
if grid > 1:
    mesh = T.instanced_grid(mesh, grid)
s = T.Scene(size, size, mesh, texs, pipe)
if os.environ.get("SUBMIT") == "frame":
    for _ in range(frames):
        s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()
    print("frames", frames, "frames_per_launch", 1)
else:
    p = np.zeros((frames, 12), np.float32)
    p[:, 0:3] = light(0.0)
    p[:, 3:6], p[:, 6:9], p[:, 9:12] = camera(0.0)
    # (the profilers' launches must all hold the same, known number of frames: groups of the scene's own size, call by
    # call -- a single call of FRAMES frames would go out in fewer, larger groups)
    g = s.frames_per_launch
    for i in range(0, frames, g):
        s.render_frames(p[i:i + g])
    print("frames", frames, "frames_per_launch", g)
print("status", s.sync())
