"""Frame and kernel times of the headline workload and the five BASELINE configs, one line each.
    python scripts/probe_configs.py [tags...]      tags: headline cfg0 cfg1 cfg2 cfg3 cfg4"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light

CFG = {"headline": ("diablo", "phong", 4096, 1, 400), "cfg0": ("african_head", "default", 800, 1, 400),
       "cfg1": ("diablo", "phong", 2048, 1, 400), "cfg2": ("diablo", "darboux", 4096, 1, 200),
       "cfg3": ("diablo", "shadow", 4096, 1, 200), "cfg4": ("diablo", "specular", 8192, 8, 40)}
kw = {}
if os.environ.get("TILE_WAVES"):
    kw["tile_waves"] = int(os.environ["TILE_WAVES"])
for tag in (sys.argv[1:] or list(CFG)):
    model, pipe, size, grid, steps = CFG[tag]
    mesh, texs = T.load_assets(find_assets(model))
    if grid > 1:
        mesh = T.instanced_grid(mesh, grid)
    s = T.Scene(size, size, mesh, texs, pipe, **kw)
    def step():
        s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()
    for _ in range(160): step()   # (past the point where a long run's groups grow: a one-time allocation of a few ms)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    s.sync()
    dt = (time.perf_counter() - t0) / steps * 1e6
    lat = []
    for _ in range(10):
        t0 = time.perf_counter(); step(); s.sync(); lat.append((time.perf_counter() - t0) * 1e6)
    s.profile_enable(True)
    for _ in range(min(steps, 100)): step()
    p = s.profile_read()
    s.profile_enable(False)
    print("%-9s %-12s %-8s %5d : frame %7.1f us  latency %7.1f us  kernels %s" % (
        tag, model, pipe, size, dt, sorted(lat)[len(lat) // 2],
        {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in p.items()}), flush=True)
    s.close()
