"""Perf probe: per-kernel device times (HIP events) for a few variations of the workload."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light

def run(label, W, H, mesh, texs, pipe, steps=50, **kw):
    s = T.Scene(W, H, mesh, texs, pipe, **kw)
    def step():
        s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()
    for _ in range(5): step()
    s.sync()
    s.profile_enable(True)
    for _ in range(steps): step()
    p = s.profile_read()
    print(label, {k: round(v["total_ms"]/v["launches"]*1e3, 1) for k, v in p.items()}, flush=True)
    s.close()

adir = find_assets("diablo")
mesh, texs = T.load_assets(adir) if adir else T.synthetic_scene()
far = dict(mesh); far["pos"] = mesh["pos"] + np.float32(100.0)
which = sys.argv[1:] or ["all"]
if "all" in which or "empty" in which:
    run("empty 4096 phong", 4096, 4096, far, texs, "phong")
if "all" in which or "base" in which:
    run("diablo 4096 phong", 4096, 4096, mesh, texs, "phong")
    run("diablo 4096 default", 4096, 4096, mesh, texs, "default")
    run("diablo 4096 darboux", 4096, 4096, mesh, texs, "darboux")
    run("diablo 4096 shadow", 4096, 4096, mesh, texs, "shadow")
if "all" in which or "sizes" in which:
    run("diablo 2048 phong", 2048, 2048, mesh, texs, "phong")
    run("diablo 1024 phong", 1024, 1024, mesh, texs, "phong")
    run("diablo 8192 phong", 8192, 8192, mesh, texs, "phong")
