"""Random polygon soups with deliberate depth ties (duplicated, coplanar and edge-sharing polygons,
coarse coordinates): the order-independent resolve -- largest z, ties to the lowest polygon index
for colour passes and to the highest for depth passes -- must equal the reference's serial loop.
CPU: the emulation (bins visited in reverse order) against the oracle, driven by hypothesis.
GPU: the same generator with fixed seeds through the C ABI."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import oracle as O
from tests import emul_bind as E
from tests import helpers as H

F = np.float32


def soup(seed, n_tri, grid):
    """n_tri polygons on a coarse lattice (many exact ties), facing either way."""
    rng = np.random.default_rng(seed)
    pts = (rng.integers(-grid, grid + 1, size=(n_tri, 3, 3)).astype(F) / F(grid)) * F(0.9)
    pts[..., 2] = np.round(pts[..., 2] * 2) / 2 * F(0.5)           # few distinct depths
    dup = rng.random(n_tri) < 0.3                                   # exact duplicates of earlier polygons
    for i in np.nonzero(dup)[0]:
        if i:
            pts[i] = pts[rng.integers(0, i)]
    share = rng.random(n_tri) < 0.3                                 # share an edge with the previous polygon
    for i in np.nonzero(share)[0]:
        if i:
            pts[i, :2] = pts[i - 1, 1:]
    pos = pts.reshape(-1, 3)
    nrm = rng.standard_normal((n_tri * 3, 3)).astype(F)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tex = np.concatenate([rng.uniform(0.05, 0.95, (n_tri * 3, 2)).astype(F), np.zeros((n_tri * 3, 1), F)], 1)
    idx = np.arange(n_tri * 3, dtype=np.uint32).reshape(n_tri, 3).repeat(3, axis=1)
    texs = [rng.integers(0, 256, (32, 32, 3), dtype=np.uint8) for _ in range(4)]
    return {"pos": pos, "tex": tex, "nrm": nrm, "idx": idx}, texs


def oracle_frame(W, Hh, mesh, texs, pipe, ca, la):
    s = O.Scene(W, Hh, mesh, texs, pipe)
    s.clear()
    s.set_light_direction(H.light(la))
    s.set_camera(*H.camera(ca))
    err = s.render()
    return err, s


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), n_tri=st.integers(1, 60), grid=st.sampled_from([2, 3, 5, 9]),
       pipe=st.sampled_from(["default", "phong", "shadow", "occlusion"]),
       ca=st.sampled_from([0.0, 0.3, 3.14159]), size=st.sampled_from([(96, 64), (130, 50), (257, 33)]))
def test_resolve_matches_serial_order_cpu(seed, n_tri, grid, pipe, ca, size):
    mesh, texs = soup(seed, n_tri, grid)
    W, Hh = size
    err, s = oracle_frame(W, Hh, mesh, texs, pipe, ca, 0.4)
    if err:
        return  # the reference would have panicked (e.g. w == 0): no defined result
    e, z, sh, fb, win = E.render(W, Hh, mesh, texs, pipe, H.light(0.4), H.camera(ca))
    assert e == 0
    assert np.array_equal(win, s.winner_u32())
    assert np.array_equal(z.view(np.uint32), s.z_f32().view(np.uint32))
    assert np.array_equal(fb, s.get_frame_buffer())
    if pipe in ("shadow", "occlusion"):
        assert np.array_equal(sh.view(np.uint32), s.shadow_f32().view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("seed", range(12))
def test_resolve_matches_serial_order_gpu(built, seed, mode):
    """Polygon soups on a coarse vertex grid: many fragments of equal depth at the same pixel (shared
    vertices and edges, coplanar overlaps), where the survivor is decided by polygon order alone --
    in both depth-resolve forms of the tile kernel (1: private keys + index compare, 2: shared keys,
    the order packed under the depth in one 64-bit atomic maximum), and, with accumulating renders,
    against what the buffers held before."""
    import tiny_renderer_amd as T
    rng = np.random.default_rng(1000 + seed)
    n_tri = int(rng.integers(1, 400))
    grid = int(rng.choice([2, 3, 5, 9]))
    pipe = ["default", "phong", "shadow", "occlusion", "darboux", "normal_map"][seed % 6]
    W, Hh = [(96, 64), (130, 50), (257, 33), (640, 480)][seed % 4]
    mesh, texs = soup(seed, n_tri, grid)
    ca = [0.0, 0.3, 3.14159][seed % 3]
    err, s = oracle_frame(W, Hh, mesh, texs, pipe, ca, 0.4)
    if err:
        pytest.skip("the reference would panic on this soup")
    g = T.Scene(W, Hh, mesh, texs, pipe, winner_tap=True, tile_mode=mode, tile_waves=[0, 4, 8, 16][seed % 4])
    g.clear()
    g.set_light_direction(H.light(0.4))
    g.set_camera(*H.camera(ca))
    g.render()
    if seed % 2:   # the same frame again WITHOUT a clear: every fragment now ties with the buffer's content
        g.render()
        assert s.render() == 0
    fb = g.get_frame_buffer()
    assert np.array_equal(g.read_winner_u32(), s.winner_u32())
    assert np.array_equal(g.read_z_f32().view(np.uint32), s.z_f32().view(np.uint32))
    assert np.array_equal(fb, s.get_frame_buffer())
    if pipe in ("shadow", "occlusion"):
        assert np.array_equal(g.read_shadow_f32().view(np.uint32), s.shadow_f32().view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("seed", range(8))
def test_resolve_matches_serial_order_gpu_frame_groups(built, seed, mode):
    """The same tie soups through tr_scene_render_frames: five views of a soup rendered by one launch of each
    kernel (groups of 3 + 2), every frame against the oracle's serial loop."""
    import tiny_renderer_amd as T
    rng = np.random.default_rng(2000 + seed)
    n_tri = int(rng.integers(1, 400))
    grid = int(rng.choice([2, 3, 5, 9]))
    pipe = ["default", "phong", "shadow", "occlusion", "darboux", "normal_map", "specular", "phong"][seed % 8]
    W, Hh = [(96, 64), (130, 50), (257, 33), (640, 480)][seed % 4]
    mesh, texs = soup(seed + 50, n_tri, grid)
    views = [(ca, 0.4 + 0.3 * k) for k, ca in enumerate([0.0, 0.3, 3.14159, 0.3, 0.0])]
    expect = []
    for ca, la in views:
        err, s = oracle_frame(W, Hh, mesh, texs, pipe, ca, la)
        if err:
            pytest.skip("the reference would panic on this soup")
        expect.append(s)
    p = np.zeros((len(views), 12), np.float32)
    for k, (ca, la) in enumerate(views):
        p[k, 0:3] = H.light(la)
        p[k, 3:6], p[k, 6:9], p[k, 9:12] = H.camera(ca)
    g = T.Scene(W, Hh, mesh, texs, pipe, tile_mode=mode, tile_waves=[0, 4, 8, 16][seed % 4], frames_per_launch=3)
    g.render_frames(p)
    assert g.frames_kept() == 3
    exact = pipe != "specular" or bool(T.load_library().tr_specular_exact())
    for back in range(3):
        s = expect[len(views) - 1 - back]
        g.select_frame(back)
        assert np.array_equal(g.read_z_f32().view(np.uint32), s.z_f32().view(np.uint32))
        fg, fo = g.get_frame_buffer(), s.get_frame_buffer()
        if exact:
            assert np.array_equal(fg, fo)
        else:
            assert np.abs(fg.astype(np.int32) - fo.astype(np.int32)).max() <= 1  # tolerance: 1 LSB (device powf)
        if pipe in ("shadow", "occlusion"):
            assert np.array_equal(g.read_shadow_f32().view(np.uint32), s.shadow_f32().view(np.uint32))
    g.close()


def far_soup(seed, n_tri):
    """Polygons that stress the f32 rounding of the edge functions: vertices far outside the frame
    (raster coordinates up to ~3e8, products far beyond 2^24), long slivers crossing the screen,
    nearly collinear triples, mixed with ordinary small polygons."""
    rng = np.random.default_rng(seed)
    pts = np.zeros((n_tri, 3, 3), F)
    for i in range(n_tri):
        kind = i % 4
        reach = F(10.0 ** rng.uniform(0, 5))                       # 1 .. 1e5 object units
        if kind == 0:                                              # huge, vertices far away
            pts[i, :, :2] = rng.uniform(-1, 1, (3, 2)) * reach
        elif kind == 1:                                            # sliver: two near vertices, one far
            p = rng.uniform(-1, 1, 2)
            pts[i, 0, :2] = p
            pts[i, 1, :2] = p + rng.uniform(-1, 1, 2) * 10.0 ** rng.uniform(-4, -1)
            pts[i, 2, :2] = rng.uniform(-1, 1, 2) * reach
        elif kind == 2:                                            # nearly collinear
            a, b = rng.uniform(-1, 1, 2) * reach, rng.uniform(-1, 1, 2) * reach
            t = rng.uniform(0.2, 0.8)
            pts[i, 0, :2], pts[i, 1, :2] = a, b
            pts[i, 2, :2] = a + t * (b - a) + rng.uniform(-1, 1, 2) * 10.0 ** rng.uniform(-3, 0)
        else:                                                      # ordinary
            pts[i, :, :2] = rng.uniform(-1, 1, 2) + rng.uniform(-0.2, 0.2, (3, 2))
        pts[i, :, 2] = rng.uniform(-0.5, 0.5, 3)
    pos = pts.reshape(-1, 3)
    nrm = rng.standard_normal((n_tri * 3, 3)).astype(F)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tex = np.concatenate([rng.uniform(0.05, 0.95, (n_tri * 3, 2)).astype(F), np.zeros((n_tri * 3, 1), F)], 1)
    idx = np.arange(n_tri * 3, dtype=np.uint32).reshape(n_tri, 3).repeat(3, axis=1)
    texs = [rng.integers(0, 256, (32, 32, 3), dtype=np.uint8) for _ in range(4)]
    return {"pos": pos, "tex": tex, "nrm": nrm, "idx": idx}, texs


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(9))
def test_far_vertices_and_slivers_gpu(built, seed):
    """The tile kernel drops 8x8 blocks that lie outside an edge by more than a rounding margin
    before testing pixels: with far-away vertices the edge functions round in f32 (products up to
    ~1e17), so this is where a margin that is too small would lose fragments.  Wide frames keep the
    screen coordinates themselves large, too."""
    import tiny_renderer_amd as T
    W, Hh = [(8192, 48), (4096, 130), (1000, 1000)][seed % 3]
    waves = [4, 8, 16][(seed // 3) % 3]
    pipe = ["phong", "normal_map", "default"][seed % 3]  # (no shadow-buffer lookups: they would leave their range)
    mesh, texs = far_soup(5000 + seed, 160)
    err, s = oracle_frame(W, Hh, mesh, texs, pipe, 0.0, 0.4)
    if err:
        pytest.skip("the reference would panic on this soup")
    g = T.Scene(W, Hh, mesh, texs, pipe, winner_tap=True, tile_waves=waves, tile_mode=1 + seed % 2)
    g.clear()
    g.set_light_direction(H.light(0.4))
    g.set_camera(*H.camera(0.0))
    g.render()
    fb = g.get_frame_buffer()
    wo, wg = s.winner_u32(), g.read_winner_u32()
    assert (wo != 0xFFFFFFFF).sum() > 1000
    assert np.array_equal(wg, wo), "winner differs at %d pixels" % int((wg != wo).sum())
    assert np.array_equal(g.read_z_f32().view(np.uint32), s.z_f32().view(np.uint32))
    assert np.array_equal(fb, s.get_frame_buffer())
