"""Per-frame kernel times of a workload through tr_scene_render_frames (TR_LIBRARY honoured):
    python scripts/probe_floor.py SIZE PIPE GRID [away]      away: camera turned so that nothing is on screen"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size, pipe, grid = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
away = len(sys.argv) > 4
mesh, texs = T.load_assets(find_assets("diablo"))
if grid > 1:
    mesh = T.instanced_grid(mesh, grid)
s = T.Scene(size, size, mesh, texs, pipe)
n = 4 * s.frames_per_launch
p = np.zeros((n, 12), np.float32)
p[:, 0:3] = light(0.0)
cam = camera(0.0)
p[:, 3:6], p[:, 6:9], p[:, 9:12] = cam
if away:
    p[:, 6:9] = (0.0, 0.0, 20.0)   # look_at behind the camera: the model projects off screen
    p[:, 3:6] = (0.0, 0.0, 10.0)
s.render_frames(p); s.sync()
s.profile_enable(True)
t0 = time.perf_counter(); s.render_frames(p); st = s.sync(); dt = (time.perf_counter() - t0) / n * 1e6
pr = s.profile_read(); s.profile_enable(False)
print("%d %s x%d%s: frame %.1f us status %d per frame: %s" % (size, pipe, grid * grid, " away" if away else "", dt, st,
      {k: round(v["total_ms"] / v["frames"] * 1e3, 2) for k, v in pr.items()}), flush=True)
