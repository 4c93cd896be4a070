"""Diagnostic: when did each tile of one k_tile launch run?  (TR_OPT_TILE_STAMPS)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pipe = sys.argv[2] if len(sys.argv) > 2 else "phong"
adir = find_assets("diablo")
mesh, texs = T.load_assets(adir) if adir else T.synthetic_scene()
s = T.Scene(size, size, mesh, texs, pipe, tile_stamps=True)
for _ in range(5):
    s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()
s.sync()
st = s.debug_tile_stamps().astype(np.int64)
t0 = st[:, 0].min()
start, end, n = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, st[:, 2]   # us
dur = end - start
busy = n > 0
stage = (st[:, 4] - st[:, 0]) / 100.0; cover = (st[:, 5] - st[:, 4]) / 100.0; shade = (st[:, 1] - st[:, 5]) / 100.0
print("tiles", len(st), "busy", int(busy.sum()), "kernel span us %.1f" % end.max())
print("empty tiles: start median %.1f max %.1f ; dur median %.2f p99 %.2f max %.2f" % (
    np.median(start[~busy]), start[~busy].max(), np.median(dur[~busy]), np.percentile(dur[~busy], 99), dur[~busy].max()))
print("busy tiles : start median %.1f p90 %.1f max %.1f ; dur median %.1f p90 %.1f max %.1f ; end max %.1f" % (
    np.median(start[busy]), np.percentile(start[busy], 90), start[busy].max(), np.median(dur[busy]),
    np.percentile(dur[busy], 90), dur[busy].max(), end[busy].max()))
order = np.argsort(-dur)[:12]
for i in order:
    print("  tile %4d n=%3d start %.1f dur %.1f hw %x" % (i, n[i], start[i], dur[i], st[i, 3]))
# correlation of duration with bin size
for lo, hi in [(1, 5), (5, 10), (10, 20), (20, 40), (40, 1000)]:
    m = (n >= lo) & (n < hi)
    if m.any():
        print("  bin %3d..%3d: %4d tiles, dur mean %.1f max %.1f | stage %.1f cover %.1f shade+store %.1f" % (
            lo, hi, m.sum(), dur[m].mean(), dur[m].max(), stage[m].mean(), cover[m].mean(), shade[m].mean()))
# concurrency: busy tiles resident at time t
for t in (5, 10, 20, 40, 60, 80, 100, 120, 140):
    print("  t=%3d us: busy resident %4d, empty resident %4d" % (t, int(((start <= t) & (end > t) & busy).sum()), int(((start <= t) & (end > t) & ~busy).sum())))
np.save("gpurun_out/stamps_%d_%s.npy" % (size, pipe), st)
