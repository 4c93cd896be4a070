"""bench.py's timed region as the driver runs it (--steps 20 --warmup 5), piece by piece: one warm-up call, idle, then
ONE timed call of `steps` identical frames, flush, device-wide synchronisation.   python scripts/probe_driver.py [steps warmup]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiny_renderer_amd as T
from bench import find_assets, camera, light
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.cuda.set_device(0)
mesh, texs = T.load_assets(find_assets("diablo"))
s = T.Scene(4096, 4096, mesh, texs, "phong")
p = np.zeros((max(steps, warm), 12), np.float32)
p[:, 0:3] = light(0.0)
p[:, 3:6], p[:, 6:9], p[:, 9:12] = camera(0.0)
s.render_frames(p[:warm]); s.flush(); torch.cuda.synchronize(); s.sync()
s.flush(); torch.cuda.synchronize()
t0 = time.perf_counter(); s.render_frames(p[:steps]); t1 = time.perf_counter(); s.flush()
if os.environ.get("SCENE_SYNC"): s.sync()
t2 = time.perf_counter()
torch.cuda.synchronize(); t3 = time.perf_counter()
print("%d steps after %d: %.1f us/step | render_frames %.1f us, flush %.1f, synchronize %.1f" % (
    steps, warm, (t3 - t0) / steps * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6), flush=True)
s.close()
