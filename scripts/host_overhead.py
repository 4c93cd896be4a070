"""How long does the host take to enqueue a frame, versus the GPU to run it?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
mesh, texs = T.load_assets(find_assets("diablo"))
s = T.Scene(4096, 4096, mesh, texs, "phong")
cam, lt = camera(0.0), light(0.0)
def step():
    s.clear(); s.set_light_direction(lt); s.set_camera(*cam); s.render()
for _ in range(50): step()
s.sync()
K = 500
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
s.sync()
t2 = time.perf_counter()
print("host enqueue %.1f us/frame, total %.1f us/frame" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
# python-only overhead: the three cheap calls
t0 = time.perf_counter()
for _ in range(K):
    s.clear(); s.set_light_direction(lt); s.set_camera(*cam)
t1 = time.perf_counter()
print("clear+set_light+set_camera (no GPU work): %.1f us/frame" % ((t1 - t0) / K * 1e6))
