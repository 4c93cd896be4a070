"""Randomised GPU-vs-oracle parity sweep of the FRAME GROUPS (tr_scene_render_frames; not part of the test
suite): `python scripts/parity_sweep_groups.py [n_seeds]`.  Per seed: a random soup (exact depth ties, or
far-away / sliver polygons), a frame shape, a tile layout, a pipeline, a group size, 2..9 frames with random
cameras and lights in ONE call; every frame the call leaves behind is compared with the oracle (z bits, rgb,
shadow bits).  Prints every mismatch."""
import sys; sys.path.insert(0, '.')
import numpy as np, time
import tiny_renderer_amd as T
from tests.test_random_meshes import far_soup, soup, oracle_frame
from tests import helpers as H
bad = 0; ran = 0; frames_checked = 0; t0 = time.time()
exact_spec = bool(T.load_library().tr_specular_exact())
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    rng = np.random.default_rng(70000 + seed)
    W, Hh = [(2048, 48), (1024, 130), (640, 480), (512, 512), (333, 777)][seed % 5]
    waves = [4, 8, 16, 0][seed % 4]
    pipe = ["phong", "normal_map", "default", "darboux", "specular", "shadow", "occlusion"][(seed // 2) % 7]
    mesh, texs = (far_soup(19000 + seed, int(rng.integers(20, 300))) if seed % 3 else soup(500 + seed, int(rng.integers(20, 400)), int(rng.choice([2, 3, 5, 9]))))
    fpl = int(rng.choice([2, 3, 4, 8, 16, 32, 0]))
    n = int(rng.integers(2, 10))
    views = [(float(rng.choice([0.0, 0.3, -1.2, 3.14159])), float(rng.uniform(-1, 1))) for _ in range(n)]
    expect = []
    for ca, la in views:
        err, s = oracle_frame(W, Hh, mesh, texs, pipe, ca, la)
        if err: break
        expect.append(s)
    if len(expect) != n: continue
    p = np.zeros((n, 12), np.float32)
    for k, (ca, la) in enumerate(views):
        p[k, 0:3] = H.light(la); p[k, 3:6], p[k, 6:9], p[k, 9:12] = H.camera(ca)
    g = T.Scene(W, Hh, mesh, texs, pipe, tile_waves=waves, tile_mode=[0, 1, 2][(seed // 5) % 3], frames_per_launch=fpl)
    g.render_frames(p)
    ran += 1
    for back in range(g.frames_kept()):
        s = expect[n - 1 - back]
        g.select_frame(back)
        okz = np.array_equal(g.read_z_f32().view(np.uint32), s.z_f32().view(np.uint32))
        d = np.abs(g.get_frame_buffer().astype(int) - s.get_frame_buffer().astype(int)).max()
        okf = d <= (1 if pipe == "specular" and not exact_spec else 0)
        oks = True
        if pipe in ("shadow", "occlusion"):
            oks = np.array_equal(g.read_shadow_f32().view(np.uint32), s.shadow_f32().view(np.uint32))
        frames_checked += 1
        if not (okz and okf and oks):
            bad += 1
            print("MISMATCH seed", seed, W, Hh, waves, pipe, "fpl", fpl, "n", n, "back", back, okz, d, oks, flush=True)
    g.close()
print("ran", ran, "calls,", frames_checked, "frames checked, bad", bad, "secs %.0f" % (time.time() - t0))
