"""The 20-step timed region of bench.py on its own (for a kernel trace): warm-up, then three repetitions of
render_frames(steps) + sync with pauses between them.  python scripts/probe_steps.py [steps] [size] [pipeline]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
pipe = sys.argv[3] if len(sys.argv) > 3 else "phong"
mesh, texs = T.load_assets(find_assets("diablo"))
s = T.Scene(size, size, mesh, texs, pipe)
def params(n, a0):
    p = np.zeros((n, 12), np.float32)
    for i in range(n):
        p[i, 0:3] = light(a0 + 0.01 * i)
        p[i, 3:6], p[i, 6:9], p[i, 9:12] = camera(a0 + 0.01 * i)
    return p
s.render_frames(params(int(os.environ.get("WARM", "8")), 0.0)); s.sync()
for rep in range(3):
    time.sleep(0.02)
    p = params(steps, 0.1 * rep)
    t0 = time.perf_counter(); s.render_frames(p); t1 = time.perf_counter(); s.sync(); t2 = time.perf_counter()
    print("rep %d: %d steps  %.1f us/step  (enqueue %.1f us, sync %.1f us)" % (rep, steps, (t2 - t0) / steps * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6), flush=True)
s.close()
