"""Frame groups vs the per-frame path: python scripts/probe_groups.py SIZE GRID MODEL PIPE [frames_per_launch...]
Prints wall-clock us/frame of a 400-frame run and the tile kernel's time per frame from dispatch events."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size, grid, model, pipe = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
fpls = [int(x) for x in sys.argv[5:]] or [0]
waves = int(os.environ.get("WAVES", "0")); mode = int(os.environ.get("MODE", "0"))
mesh, texs = T.load_assets(find_assets(model))
if grid > 1:
    mesh = T.instanced_grid(mesh, grid)
N = int(os.environ.get("FRAMES", "400"))
p = np.zeros((N, 12), np.float32)
for i in range(N):
    p[i, 0:3] = light(0.0); f, a, u = camera(0.0); p[i, 3:6], p[i, 6:9], p[i, 9:12] = f, a, u

def per_frame(s, n):
    for _ in range(n):
        s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()

s = T.Scene(size, size, mesh, texs, pipe, tile_waves=waves, tile_mode=mode)
per_frame(s, 20); s.sync()
t0 = time.perf_counter(); per_frame(s, N); s.sync(); t1 = time.perf_counter()
s.profile_enable(True); per_frame(s, 40); pr = s.profile_read(); s.profile_enable(False)
print("%5d x%d %-8s %-9s per-frame path : %6.1f us/frame wall | k_tile %6.1f us" % (
    size, grid * grid, model, pipe, (t1 - t0) / N * 1e6, pr["k_tile"]["total_ms"] / pr["k_tile"]["frames"] * 1e3), flush=True)
s.close()
for fpl in fpls:
    s = T.Scene(size, size, mesh, texs, pipe, frames_per_launch=fpl, tile_waves=waves, tile_mode=mode)
    s.render_frames(p[:32]); s.sync()
    t0 = time.perf_counter(); s.render_frames(p); s.sync(); t1 = time.perf_counter()
    s.profile_enable(True); s.render_frames(p[:64]); pr = s.profile_read(); s.profile_enable(False)
    kt = pr["k_tile"]
    extra = ""
    if "k_tile_depth" in pr:
        extra = " + depth %6.1f" % (pr["k_tile_depth"]["total_ms"] / pr["k_tile_depth"]["frames"] * 1e3)
    print("%5d x%d %-8s %-9s groups of %2d     : %6.1f us/frame wall | k_tile %6.1f us/frame%s (%d launches) | setup %.1f order %.1f per launch" % (
        size, grid * grid, model, pipe, s.frames_per_launch, (t1 - t0) / N * 1e6, kt["total_ms"] / kt["frames"] * 1e3, extra, kt["launches"],
        pr["k_setup"]["total_ms"] / pr["k_setup"]["launches"] * 1e3, pr["k_order"]["total_ms"] / pr["k_order"]["launches"] * 1e3), flush=True)
    s.close()
