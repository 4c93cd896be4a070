"""The per-frame protocol with the automatic groups switched off (one launch of each kernel per frame, pipelined over
the scene's two streams): us per frame.   python scripts/probe_unfused.py [SIZE PIPE]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pipe = sys.argv[2] if len(sys.argv) > 2 else "phong"
mesh, texs = T.load_assets(find_assets("diablo"))
s = T.Scene(size, size, mesh, texs, pipe, auto_group=False)
def step(a):
    s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(a)); s.render()
for i in range(100): step(0.0)
s.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(500): step(0.0)
    s.sync()
    print("%d %s unfused: %.1f us/frame" % (size, pipe, (time.perf_counter() - t0) / 500 * 1e6), flush=True)
s.profile_enable(True)
for i in range(100): step(0.0)
p = s.profile_read()
print({k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in p.items()})
s.close()
