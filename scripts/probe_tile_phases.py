"""Where a busy tile's workgroup spends its life (TR_OPT_TILE_STAMPS: start, bin staged, coverage done, end in 10 ns ticks):
python scripts/probe_tile_phases.py [SIZE PIPE MODEL]   -- a lone frame, per-frame kernel (the stamps' scene has no groups)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pipe = sys.argv[2] if len(sys.argv) > 2 else "phong"
model = sys.argv[3] if len(sys.argv) > 3 else "diablo"
mesh, texs = T.load_assets(find_assets(model))
s = T.Scene(size, size, mesh, texs, pipe, tile_stamps=True)
for rep in range(3):
    s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render(); s.sync()
st = s.debug_tile_stamps().astype(np.int64)
busy = st[st[:, 2] > 0]
t0 = busy[:, 0].min()
life, stage, cover, shade = (busy[:, 1] - busy[:, 0]) / 100.0, (busy[:, 4] - busy[:, 0]) / 100.0, (busy[:, 5] - busy[:, 4]) / 100.0, (busy[:, 1] - busy[:, 5]) / 100.0
print("%d busy tiles, polygons per tile: mean %.1f median %d p90 %d max %d" % (len(busy), busy[:, 2].mean(), np.median(busy[:, 2]), np.percentile(busy[:, 2], 90), busy[:, 2].max()))
print("kernel span %.1f us (first start to last end)" % ((busy[:, 1].max() - t0) / 100.0))
for name, v in (("lifetime", life), ("start -> bin staged", stage), ("staged -> coverage done", cover), ("coverage done -> end (shade + store)", shade)):
    print("%-38s mean %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (name, v.mean(), np.median(v), np.percentile(v, 90), v.max()))
print("sum of lifetimes %.0f us = %.1f workgroup-slots busy over the span" % (life.sum(), life.sum() / ((busy[:, 1].max() - t0) / 100.0)))
