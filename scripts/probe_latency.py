"""Single-frame latency (clear -> render -> sync, nothing in flight) and the kernels' own durations in that loop.
    python scripts/probe_latency.py [SIZE PIPE MODEL]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pipe = sys.argv[2] if len(sys.argv) > 2 else "phong"
model = sys.argv[3] if len(sys.argv) > 3 else "diablo"
mesh, texs = T.load_assets(find_assets(model))
kw = {}
if os.environ.get("TILE_WAVES"): kw["tile_waves"] = int(os.environ["TILE_WAVES"])
if os.environ.get("TILE_MODE"): kw["tile_mode"] = int(os.environ["TILE_MODE"])
s = T.Scene(size, size, mesh, texs, pipe, **kw)
def step():
    s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()
for _ in range(10):
    step(); s.sync()
lat = []
for _ in range(40):
    t0 = time.perf_counter(); step(); s.sync(); lat.append((time.perf_counter() - t0) * 1e6)
t0 = time.perf_counter()
for _ in range(40): s.sync()
sync_only = (time.perf_counter() - t0) / 40 * 1e6
t0 = time.perf_counter()
for _ in range(40): s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0))
calls3 = (time.perf_counter() - t0) / 40 * 1e6
s.profile_enable(True)
for _ in range(40):
    step(); s.sync()
p = s.profile_read(); s.profile_enable(False)
lat.sort()
print("%d %s %s: latency median %.1f min %.1f us | empty sync %.1f | clear+set_light+set_camera %.1f | kernels (own duration, us): %s" % (
    size, pipe, model, lat[len(lat) // 2], lat[0], sync_only, calls3, {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in p.items()}))
