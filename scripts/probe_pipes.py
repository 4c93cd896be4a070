"""k_tile time of several pipelines on one geometry: python scripts/probe_pipes.py SIZE GRID pipe..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size, grid = int(sys.argv[1]), int(sys.argv[2])
mesh, texs = T.load_assets(find_assets("diablo"))
if grid > 1:
    mesh = T.instanced_grid(mesh, grid)
for pipe in sys.argv[3:]:
    s = T.Scene(size, size, mesh, texs, pipe)
    def step():
        s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()
    for _ in range(5): step()
    s.sync()
    s.profile_enable(True)
    for _ in range(20): step()
    p = s.profile_read()
    print("%5d x%d %-10s %s" % (size, grid * grid, pipe, {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in p.items()}), flush=True)
    s.close()
