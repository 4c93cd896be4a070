"""A plain loop of frame-group calls for profilers: python3 scripts/group_loop.py SIZE PIPE FRAMES [MODEL [GRID [FPL]]]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size, pipe, frames = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
model = sys.argv[4] if len(sys.argv) > 4 else "diablo"
grid = int(sys.argv[5]) if len(sys.argv) > 5 else 1
fpl = int(sys.argv[6]) if len(sys.argv) > 6 else 0
mesh, texs = T.load_assets(find_assets(model))
if grid > 1:
    mesh = T.instanced_grid(mesh, grid)
p = np.zeros((frames, 12), np.float32)
for i in range(frames):
    p[i, 0:3] = light(0.0); f, a, u = camera(0.0); p[i, 3:6], p[i, 6:9], p[i, 9:12] = f, a, u
s = T.Scene(size, size, mesh, texs, pipe, frames_per_launch=fpl)
s.render_frames(p[:4 * s.frames_per_launch]); s.sync()
s.render_frames(p); s.sync()
s.close()
