"""Randomised GPU-vs-oracle parity sweep (not part of the test suite): `python scripts/parity_sweep.py [n_seeds]`.
Polygon soups with exact depth ties and far-away / sliver polygons, five frame shapes, every tile
layout (waves per tile x column / shared resolve), five pipelines; prints every mismatch.  Round 1: 2 494 cases
(2 500 seeds), 0 mismatches; round 2: see profiles/r02_notes.md."""
import sys; sys.path.insert(0, '.')
import numpy as np, time
import tiny_renderer_amd as T
from tests.test_random_meshes import far_soup, soup, oracle_frame
from tests import helpers as H
bad = 0; ran = 0; t0 = time.time()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    rng = np.random.default_rng(seed)
    W, Hh = [(8192, 48), (4096, 130), (1000, 1000), (2048, 512), (333, 777)][seed % 5]
    waves = [4, 8, 16, 0][seed % 4]
    pipe = ["phong", "normal_map", "default", "darboux", "specular"][(seed // 2) % 5]
    mesh, texs = (far_soup(9000 + seed, int(rng.integers(20, 300))) if seed % 3 else soup(seed, int(rng.integers(20, 400)), int(rng.choice([2, 3, 5, 9]))))
    ca = float(rng.choice([0.0, 0.3, -1.2, 3.14159]))
    err, s = oracle_frame(W, Hh, mesh, texs, pipe, ca, 0.4)
    if err: continue
    g = T.Scene(W, Hh, mesh, texs, pipe, winner_tap=True, tile_waves=waves, tile_mode=[0, 1, 2][(seed // 5) % 3])
    g.clear(); g.set_light_direction(H.light(0.4)); g.set_camera(*H.camera(ca)); g.render()
    fb = g.get_frame_buffer()
    okw = np.array_equal(g.read_winner_u32(), s.winner_u32())
    okz = np.array_equal(g.read_z_f32().view(np.uint32), s.z_f32().view(np.uint32))
    d = np.abs(fb.astype(int) - s.get_frame_buffer().astype(int)).max()
    okf = d <= (1 if pipe == "specular" and not T.load_library().tr_specular_exact() else 0)
    ran += 1
    if not (okw and okz and okf):
        bad += 1
        print("MISMATCH seed", seed, W, Hh, waves, pipe, ca, okw, okz, d, flush=True)
    g.close()
print("ran", ran, "bad", bad, "secs %.0f" % (time.time() - t0))
