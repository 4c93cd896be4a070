"""Diagnostic: when did each tile of one k_tile launch run?  (TR_OPT_TILE_STAMPS)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pipe = sys.argv[2] if len(sys.argv) > 2 else "phong"
model = sys.argv[3] if len(sys.argv) > 3 else "diablo"
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 1
adir = find_assets(model)
mesh, texs = T.load_assets(adir) if adir else T.synthetic_scene()
if grid > 1:
    mesh = T.instanced_grid(mesh, grid)
s = T.Scene(size, size, mesh, texs, pipe, tile_stamps=True)
for _ in range(5):
    s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()
s.sync()
st = s.debug_tile_stamps().astype(np.int64)
t0 = st[st[:, 2] > 0, 0].min()  # workgroups of empty tiles leave no stamp
start, end, n = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, st[:, 2]   # us
dur = end - start
busy = n > 0
stage = (st[:, 4] - st[:, 0]) / 100.0; cover = (st[:, 5] - st[:, 4]) / 100.0; shade = (st[:, 1] - st[:, 5]) / 100.0
print("tiles", len(st), "busy", int(busy.sum()), "busy span us %.1f" % end[busy].max())
print("busy tiles : start median %.1f p90 %.1f max %.1f ; dur median %.1f p90 %.1f max %.1f ; end max %.1f" % (
    np.median(start[busy]), np.percentile(start[busy], 90), start[busy].max(), np.median(dur[busy]),
    np.percentile(dur[busy], 90), dur[busy].max(), end[busy].max()))
order = np.argsort(-np.where(busy, dur, 0))[:12]
for i in order:
    print("  tile %4d n=%3d start %.1f dur %.1f hw %x" % (i, n[i], start[i], dur[i], st[i, 3]))
# correlation of duration with bin size
for lo, hi in [(1, 5), (5, 10), (10, 20), (20, 40), (40, 1000)]:
    m = (n >= lo) & (n < hi)
    if m.any():
        print("  bin %3d..%3d: %4d tiles, dur mean %.1f max %.1f | stage %.1f cover %.1f shade+store %.1f" % (
            lo, hi, m.sum(), dur[m].mean(), dur[m].max(), stage[m].mean(), cover[m].mean(), shade[m].mean()))
# concurrency: busy tiles resident at time t
for t in range(2, 60, 3):
    print("  t=%3d us: busy resident %4d  started %4d finished %4d" % (
        t, int(((start <= t) & (end > t) & busy).sum()), int(((start <= t) & busy).sum()), int(((end <= t) & busy).sum())))
# per compute unit: when did its last busy tile end, how many did it run
cu = st[:, 3]
ends = {}
for c in np.unique(cu[busy]):
    m = busy & (cu == c)
    ends[c] = (end[m].max(), int(m.sum()), float(dur[m].sum()))
e = np.array([v[0] for v in ends.values()]); k = np.array([v[1] for v in ends.values()]); w = np.array([v[2] for v in ends.values()])
print("CUs with busy tiles %d: last end min %.1f median %.1f max %.1f ; tiles per CU min %d median %d max %d ; sum of durations per CU median %.0f max %.0f" % (
    len(e), e.min(), np.median(e), e.max(), k.min(), np.median(k), k.max(), np.median(w), w.max()))
np.save("gpurun_out/stamps_%d_%s.npy" % (size, pipe), st)
