"""configs[4]'s geometry (diablo x64 grid, 8192x8192) under several pipelines: k_tile per frame.
    python scripts/probe_pipe5.py [pipelines...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
mesh, texs = T.load_assets(find_assets("diablo"))
mesh = T.instanced_grid(mesh, 8)
for pipe in (sys.argv[1:] or ["phong", "normal_map", "specular"]):
    s = T.Scene(8192, 8192, mesh, texs, pipe)
    p = np.zeros((16, 12), np.float32)
    p[:, 0:3] = light(0.0)
    p[:, 3:6], p[:, 6:9], p[:, 9:12] = camera(0.0)
    s.render_frames(p); s.sync()
    s.profile_enable(True)
    t0 = time.perf_counter(); s.render_frames(p); s.sync(); dt = (time.perf_counter() - t0) / 16 * 1e6
    pr = s.profile_read(); s.profile_enable(False)
    print("%-10s frame %.1f us  per frame: %s" % (pipe, dt, {k: round(v["total_ms"] / v["frames"] * 1e3, 1) for k, v in pr.items()}), flush=True)
    s.close()
