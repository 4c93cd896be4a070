"""One-time cost of growing the frame groups: the first long tr_scene_render_frames call of a scene against the second.
    python scripts/probe_growth.py [frames]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mesh, texs = T.load_assets(find_assets("diablo"))
t0 = time.perf_counter()
s = T.Scene(4096, 4096, mesh, texs, "phong")
s.sync()
print("create %.1f ms" % ((time.perf_counter() - t0) * 1e3))
p = np.zeros((n, 12), np.float32)
p[:, 0:3] = light(0.0)
p[:, 3:6], p[:, 6:9], p[:, 9:12] = camera(0.0)
for label, k in (("8 frames", 8), ("8 frames", 8), ("%d frames" % n, n), ("%d frames" % n, n), ("%d frames" % n, n)):
    t0 = time.perf_counter(); s.render_frames(p[:k]); t1 = time.perf_counter(); s.sync(); t2 = time.perf_counter()
    print("%-12s %8.1f us total  %6.1f us/frame  (enqueue %.1f us)" % (label, (t2 - t0) * 1e6, (t2 - t0) / k * 1e6, (t1 - t0) * 1e6), flush=True)
s.close()
