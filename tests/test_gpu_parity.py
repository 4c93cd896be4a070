"""GPU parity: the HIP path through the C ABI against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): winner index and z bit-exact; rgb exact for every pipeline
whose arithmetic is +,-,*,/,sqrt only (default, phong, normal_map, darboux, shadow,
occlusion -- the occlusion sample offsets are computed on the host with the same libm as the
oracle), and for specular too when the library reproduces the host libm's powf (tr_specular_exact():
it does wherever the build found glibc's powf tables; otherwise the device library's powf is
used and the tolerance for specular is 1 LSB per channel).  PARITY UNPINNED upstream: the oracle is the normative restatement (oracle/tr_oracle.h).
"""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

EXACT = ("default", "phong", "normal_map", "darboux", "shadow", "occlusion")
ALL = EXACT + ("specular",)


def specular_exact():
    import tiny_renderer_amd as T
    return bool(T.load_library().tr_specular_exact())


def render_pair(W, Hh, mesh, texs, pipe, cam_angle, light_angle, **kw):
    import tiny_renderer_amd as T
    from oracle import oracle as O
    gpu = T.Scene(W, Hh, mesh, texs, pipe, winner_tap=True, **kw)
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    for s in (gpu, cpu):
        s.clear()
        s.set_light_direction(H.light(light_angle))
        s.set_camera(*H.camera(cam_angle))
    assert cpu.render() == 0
    gpu.render()
    return gpu, cpu


def assert_parity(gpu, cpu, pipe):
    zo, zg = cpu.z_f32().view(np.uint32), gpu.read_z_f32().view(np.uint32)
    assert np.array_equal(zg, zo), "z bits differ at %d pixels" % int((zg != zo).sum())
    wo, wg = cpu.winner_u32(), gpu.read_winner_u32()
    assert np.array_equal(wg, wo), "winner differs at %d pixels" % int((wg != wo).sum())
    if pipe in ("shadow", "occlusion"):
        so, sg = cpu.shadow_f32().view(np.uint32), gpu.read_shadow_f32().view(np.uint32)
        assert np.array_equal(sg, so), "shadow bits differ at %d pixels" % int((sg != so).sum())
    fo, fg = cpu.get_frame_buffer(), gpu.get_frame_buffer()
    if pipe in EXACT or specular_exact():
        assert np.array_equal(fg, fo), "rgb differs at %d pixels" % int((fg != fo).any(-1).sum())
    else:
        d = np.abs(fg.astype(np.int32) - fo.astype(np.int32))
        assert d.max() <= 1, "specular rgb differs by %d" % int(d.max())  # tolerance: 1 LSB
    assert (wo != 0xFFFFFFFF).sum() > 0


@pytest.mark.parametrize("pipe", ALL)
def test_synthetic_all_pipelines(synthetic, pipe):
    mesh, texs = synthetic
    gpu, cpu = render_pair(640, 480, mesh, texs, pipe, 0.4, -0.7)
    assert_parity(gpu, cpu, pipe)


@pytest.mark.parametrize("pipe", ALL)
@pytest.mark.parametrize("waves", [4, 8, 16])
@pytest.mark.parametrize("mode", [1, 2])
def test_tile_layouts(synthetic, pipe, waves, mode):
    """tr_options.tile_waves / tile_mode: four, eight or sixteen wavefronts per screen tile, each owning a
    column of the tile (1) or a share of its bin with the depth resolve through LDS atomics (2); the
    automatic choice goes by tile count.  Results must not depend on any of it."""
    mesh, texs = synthetic
    gpu, cpu = render_pair(801, 603, mesh, texs, pipe, -0.3, 0.9, tile_waves=waves, tile_mode=mode)
    assert_parity(gpu, cpu, pipe)
    for s in (gpu, cpu):  # and an accumulating render on top
        s.set_camera(*H.camera(0.8))
        s.render()
    assert_parity(gpu, cpu, pipe)


@pytest.mark.parametrize("pipe", ALL)
@pytest.mark.parametrize("angles", [(0.0, 0.0), (0.7, -1.1)])
def test_diablo_800(diablo, pipe, angles):
    mesh, texs = diablo
    gpu, cpu = render_pair(800, 800, mesh, texs, pipe, *angles)
    assert_parity(gpu, cpu, pipe)


def test_african_head_default_800(african_head):
    """BASELINE.json configs[0]."""
    mesh, texs = african_head
    gpu, cpu = render_pair(800, 800, mesh, texs, "default", 0.0, 0.0)
    assert_parity(gpu, cpu, "default")
    assert cpu.stats()[0]["tri_kept"] == 1841


def test_diablo_phong_2048(diablo):
    """BASELINE.json configs[1]."""
    mesh, texs = diablo
    gpu, cpu = render_pair(2048, 2048, mesh, texs, "phong", 0.0, 0.0)
    assert_parity(gpu, cpu, "phong")


@pytest.mark.parametrize("size", [(801, 603), (130, 70), (64, 64), (1, 1), (4100, 36)])
def test_ragged_sizes(small_synthetic, size):
    """Widths that are not multiples of 4 / 16 / the tile take the byte-store paths."""
    mesh, texs = small_synthetic
    gpu, cpu = render_pair(size[0], size[1], mesh, texs, "phong", 0.2, 0.3)
    zo, zg = cpu.z_f32().view(np.uint32), gpu.read_z_f32().view(np.uint32)
    assert np.array_equal(zg, zo)
    assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer())
    assert np.array_equal(gpu.read_winner_u32(), cpu.winner_u32())


@pytest.mark.parametrize("pipe", ["phong", "shadow"])
def test_render_twice_without_clear_accumulates(small_synthetic, pipe):
    """scene.rs:151: render does not clear; a second render depth-tests against the first."""
    mesh, texs = small_synthetic
    gpu, cpu = render_pair(320, 256, mesh, texs, pipe, 0.0, 0.0)
    for s in (gpu, cpu):
        s.set_camera(*H.camera(0.5))
        s.set_light_direction(H.light(0.9))
        s.render()
    assert_parity(gpu, cpu, pipe)


@pytest.mark.parametrize("peek", [False, True])
@pytest.mark.parametrize("band", [None, (100, 420)])
def test_fast_depth_clear(small_synthetic, peek, band):
    """Empty tiles of a cleared frame keep their z behind a per-tile flag instead of in memory.  An
    accumulating render that lands on such tiles, a clear that is read before any render, and the
    z getters (with the flags still up, or already lowered by an earlier read) must all see f32::MIN."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh = 1024, 512
    kw = {"band_rows": band} if band else {}
    gpu = T.Scene(W, Hh, mesh, texs, "phong", winner_tap=True, **kw)
    cpu = O.Scene(W, Hh, mesh, texs, "phong")
    rows = slice(*band) if band else slice(None)  # output rows (row 0 = top) this scene owns
    zrows = slice(Hh - band[1], Hh - band[0]) if band else slice(None)  # z rows count from the bottom

    def check():
        zo = cpu.z_f32().view(np.uint32).reshape(Hh, W)[zrows]
        zg = gpu.read_z_f32().view(np.uint32).reshape(Hh, W)[zrows]
        assert np.array_equal(zg, zo)
        assert np.array_equal(gpu.get_frame_buffer()[rows], cpu.get_frame_buffer()[rows])
        assert np.array_equal(gpu.get_z_buffer()[rows], cpu.get_z_buffer()[rows])

    views = [([0.0, 0.0, 1.0], [0.0, 0.0, 0.0]), ([0.9, 0.3, 1.0], [0.9, 0.3, 0.0]), ([-0.8, -0.4, 1.0], [-0.8, -0.4, 0.0])]
    for s in (gpu, cpu):
        s.clear()
        s.set_light_direction(H.light(0.3))
    for k, (frm, at) in enumerate(views):  # three renders without a clear, each on other tiles
        for s in (gpu, cpu):
            s.set_camera(frm, at, [0.0, 1.0, 0.0])
            s.render()
        if peek or k == len(views) - 1:
            check()
    for s in (gpu, cpu):  # a clear read back before any render, then a render on top of it
        s.clear()
    if peek:
        check()
    for s in (gpu, cpu):
        s.set_camera(views[1][0], views[1][1], [0.0, 1.0, 0.0])
        s.render()
    check()


def test_initial_state_and_clear_only(small_synthetic):
    """Buffer::new zero-fills (shader.rs:46-47); clear sets f32::MIN / 0 (scene.rs:128-137)."""
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    s = T.Scene(96, 40, mesh, texs, "phong", winner_tap=True)
    assert np.all(s.read_z_f32() == 0.0) and np.all(s.read_shadow_f32() == 0.0)
    assert np.all(s.get_frame_buffer() == 0)
    s.clear()
    assert np.all(s.read_z_f32().view(np.uint32) == 0xFF7FFFFF)
    assert np.all(s.read_shadow_f32().view(np.uint32) == 0xFF7FFFFF)
    assert np.all(s.get_frame_buffer() == 0) and np.all(s.get_z_buffer() == 0)


def test_degenerate_inputs_and_lifetimes(small_synthetic):
    """No polygons at all, a model entirely off screen, tear-down with frames still in flight, two
    scenes interleaved on one device."""
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    empty = dict(mesh, idx=np.zeros((0, 9), np.uint32))
    for pipe in ("phong", "shadow"):
        s = T.Scene(300, 200, empty, texs, pipe, winner_tap=True)
        s.clear()
        s.render()
        assert not s.get_frame_buffer().any()
        assert (s.read_z_f32().view(np.uint32) == 0xFF7FFFFF).all() and (s.read_winner_u32() == 0xFFFFFFFF).all()
        s.render()
        assert s.sync() == 0
        s.close()
    far = dict(mesh, pos=mesh["pos"] + np.float32(1000.0))
    s = T.Scene(300, 200, far, texs, "phong")
    s.clear()
    s.render()
    assert not s.get_frame_buffer().any() and s.sync() == 0
    s.close()
    s = T.Scene(1024, 1024, mesh, texs, "phong")
    for i in range(100):
        s.clear()
        s.set_camera(*H.camera(0.01 * i))
        s.render()
    s.close()  # frames still in flight
    a, b = T.Scene(512, 512, mesh, texs, "phong"), T.Scene(640, 360, mesh, texs, "darboux")
    for i in range(10):
        for sc in (a, b):
            sc.clear()
            sc.set_camera(*H.camera(0.1 * i))
            sc.render()
    alone = T.Scene(512, 512, mesh, texs, "phong")
    alone.clear()
    alone.set_camera(*H.camera(0.9))
    alone.render()
    assert np.array_equal(a.get_frame_buffer(), alone.get_frame_buffer())


def test_async_frame_readback(small_synthetic):
    """tr_scene_get_frame_buffer_async: read-backs queued behind their frames, no host wait between
    frames; every copy holds its own frame after one sync."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh = 640, 360
    gpu = T.Scene(W, Hh, mesh, texs, "phong")
    cpu = O.Scene(W, Hh, mesh, texs, "phong")
    outs = [gpu.pinned_frame() for _ in range(3)]
    for k, out in enumerate(outs):
        gpu.clear()
        gpu.set_light_direction(H.light(0.2 * k))
        gpu.set_camera(*H.camera(0.5 * k))
        gpu.render()
        gpu.get_frame_buffer_async(out)
    assert gpu.sync() == 0
    for k, out in enumerate(outs):
        cpu.clear()
        cpu.set_light_direction(H.light(0.2 * k))
        cpu.set_camera(*H.camera(0.5 * k))
        assert cpu.render() == 0
        assert np.array_equal(out, cpu.get_frame_buffer()), "frame %d" % k
    gpu.clear()   # a clear that is read back before any render
    gpu.get_frame_buffer_async(outs[0])
    assert gpu.sync() == 0 and not outs[0].any()
    gpu.close()


def test_sparse_read_back_orbit(diablo):
    """The reference hands every frame to its window (app.rs:213-218).  Into page-locked buffers only the tiles
    that are not zeros on both sides cross PCIe (k_read_back): the camera orbits over twelve frames read back
    alternately into two buffers -- tiles that fill, tiles that EMPTY again (they must be re-zeroed in the buffer
    that still holds an older frame), a frame read into a buffer the caller has scribbled on -- and every host
    frame is the oracle's."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = diablo
    W, Hh, n = 1024, 1024, 12
    gpu = T.Scene(W, Hh, mesh, texs, "phong")
    cpu = O.Scene(W, Hh, mesh, texs, "phong")
    pinned = [gpu.pinned_frame() for _ in range(2)]
    for b in pinned:
        b[...] = 99                                   # unknown content to begin with
    kept = []
    for f in range(n):
        ca = 0.55 * f
        for s in (gpu, cpu):
            s.clear(), s.set_light_direction(H.light(0.3)), s.set_camera(*H.camera(ca)), s.render()
        if f == 7:
            pinned[f % 2][100:200] = 55               # the caller writes into its buffer ...
            gpu.host_buffer_written(pinned[f % 2])    # ... and says so
        gpu.get_frame_buffer_async(pinned[f % 2])
        if f % 2 == 1 or f == n - 1:
            assert gpu.sync() == 0                    # two frames in flight at a time
            for k in ((f - 1, f) if f % 2 == 1 else (f,)):
                kept.append((k, pinned[k % 2].copy()))
        want = cpu.get_frame_buffer()
        if f % 2 == 1 or f == n - 1:
            got = dict(kept)[f]
            assert np.array_equal(got, want), "frame %d: %d pixels differ" % (f, int((got != want).any(-1).sum()))
        else:
            first_of_pair = want
    # (the first frame of each pair is checked through the second buffer one frame later: above only the second
    # and the last are compared with the oracle at the time; do all of them now)
    cpu2 = O.Scene(W, Hh, mesh, texs, "phong")
    for k, got in kept:
        cpu2.clear(), cpu2.set_light_direction(H.light(0.3)), cpu2.set_camera(*H.camera(0.55 * k)), cpu2.render()
        assert np.array_equal(got, cpu2.get_frame_buffer()), "frame %d" % k
    assert len(kept) == n
    gpu.close()


def test_band_scene_reads_back_the_whole_buffer(small_synthetic):
    """A band scene (tr_options.band_row0/1) renders some rows of a frame buffer whose other rows belong to somebody
    else (other ranks' bands in an all-gather buffer).  tr_scene_get_frame_buffer_async into page-locked memory must
    deliver the COMPLETE buffer -- what tr_scene_get_frame_buffer returns -- not only the band's tiles: the sparse
    tile path is for scenes that render the whole frame."""
    import torch
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W, Hh = 512, 256
    full = torch.full((Hh * W * 3,), 7, dtype=torch.uint8, device="cuda")   # "other ranks' rows": sevens
    torch.cuda.synchronize()
    gpu = T.Scene(W, Hh, mesh, texs, "phong", band_rows=(64, 176), frame_buffer_device=full.data_ptr())
    pinned = gpu.pinned_frame()
    for k in range(3):
        pinned[...] = 0xAB                            # garbage the read-back must replace everywhere
        gpu.clear(), gpu.set_light_direction(H.light(0.2)), gpu.set_camera(*H.camera(0.4 * k)), gpu.render()
        gpu.get_frame_buffer_async(pinned)
        assert gpu.sync() == 0
        want = gpu.get_frame_buffer()
        assert (want[:64] == 7).all() and (want[176:] == 7).all() and want[64:176].any()
        assert np.array_equal(pinned, want), "read-back %d" % k
    gpu.close()


def test_two_scenes_share_one_pinned_buffer(small_synthetic):
    """The record of which tiles of a page-locked buffer hold zeros belongs to the scene that wrote the buffer last:
    when another scene has read back into the same buffer in between, nothing may be skipped."""
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W, Hh = 1024, 512
    a = T.Scene(W, Hh, mesh, texs, "phong")
    # the second scene's model fills tiles the first one leaves empty (the same sphere, off to the right)
    moved = dict(mesh)
    moved["pos"] = (mesh["pos"] * np.float32(0.5) + np.array([0.45, 0.3, 0.0], np.float32)).astype(np.float32)
    b = T.Scene(W, Hh, moved, texs, "default")
    pinned = a.pinned_frame()
    frames = {}
    for name, s in (("a", a), ("b", b)):
        s.clear(), s.set_light_direction(H.light(0.1)), s.set_camera(*H.camera(0.0)), s.render()
        frames[name] = s.get_frame_buffer()
    assert (frames["b"].any(-1) & ~frames["a"].any(-1)).sum() > 1000    # b lights pixels a leaves black
    for k, (name, s) in enumerate((("a", a), ("b", b), ("a", a), ("a", a), ("b", b), ("b", b), ("a", a))):
        s.clear(), s.set_light_direction(H.light(0.1)), s.set_camera(*H.camera(0.0)), s.render()
        s.get_frame_buffer_async(pinned)
        assert s.sync() == 0
        assert np.array_equal(pinned, frames[name]), "read-back %d (scene %s)" % (k, name)
    b.close()
    a.close()


def test_max_frame_slots_bounds_the_groups(small_synthetic):
    """tr_options.max_frame_slots caps the frame slots (and with them the frames per launch); results do not change."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh = 320, 200
    for cap in (1, 3):
        gpu = T.Scene(W, Hh, mesh, texs, "shadow", max_frame_slots=cap)
        assert gpu.frames_per_launch == cap
        frames = np.zeros((11, 12), np.float32)
        for i in range(len(frames)):
            frames[i, 0:3] = H.light(0.3 * i)
            frames[i, 3:6], frames[i, 6:9], frames[i, 9:12] = H.camera(0.25 * i)
        gpu.render_frames(frames)
        assert gpu.frames_kept() == min(cap, len(frames))
        cpu = O.Scene(W, Hh, mesh, texs, "shadow")
        for back in range(gpu.frames_kept()):
            q = frames[len(frames) - 1 - back]
            cpu.clear(), cpu.set_light_direction(q[0:3]), cpu.set_camera(q[3:6], q[6:9], q[9:12]), cpu.render()
            gpu.select_frame(back)
            assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer()), (cap, back)
        cpu.close()
        gpu.close()
    with pytest.raises(T.TinyRendererError):
        T.Scene(W, Hh, mesh, texs, "phong", max_frame_slots=2, frames_per_launch=4)


def test_depth_views(small_synthetic):
    """get_z_buffer / get_shadow_buffer (scene.rs:101-125)."""
    mesh, texs = small_synthetic
    gpu, cpu = render_pair(320, 256, mesh, texs, "shadow", 0.3, 0.8)
    assert np.array_equal(gpu.get_z_buffer(), cpu.get_z_buffer())
    assert np.array_equal(gpu.get_shadow_buffer(), cpu.get_shadow_buffer())


def test_instanced_grid_specular(synthetic):
    """BASELINE.json configs[4] in small: the n x n instanced grid (SURVEY 8d defines the instancing),
    -s specular, many small polygons per tile; whole frame and two of its row bands."""
    import tiny_renderer_amd as T
    mesh, texs = synthetic
    grid = T.instanced_grid(mesh, 4)
    assert grid["idx"].shape[0] == 16 * mesh["idx"].shape[0]
    W, Hh = 2048, 1024
    gpu, cpu = render_pair(W, Hh, grid, texs, "specular", 0.0, 0.0)
    assert_parity(gpu, cpu, "specular")
    ref = cpu.get_frame_buffer()
    for band in ((0, 256), (512, 1024)):
        b = T.Scene(W, Hh, grid, texs, "specular", band_rows=band)
        b.clear()
        b.set_light_direction(H.light(0.0))
        b.set_camera(*H.camera(0.0))
        b.render()
        d = np.abs(b.get_frame_buffer()[band[0]:band[1]].astype(np.int32) - ref[band[0]:band[1]].astype(np.int32))
        assert d.max() <= (0 if specular_exact() else 1)  # tolerance: 1 LSB only with the device library's powf
        b.close()


def test_determinism(synthetic):
    import tiny_renderer_amd as T
    mesh, texs = synthetic
    s = T.Scene(1024, 1024, mesh, texs, "darboux")
    frames = []
    for _ in range(3):
        s.clear()
        s.set_light_direction(H.light(0.3))
        s.set_camera(*H.camera(1.0))
        s.render()
        frames.append(s.get_frame_buffer())
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[1], frames[2])


@pytest.mark.parametrize("n_bands", [2, 4, 8])
def test_band_shards_reassemble(small_synthetic, n_bands):
    """row bands rendered by separate scenes concatenate to the single-scene frame."""
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W, Hh = 512, 384
    full = T.Scene(W, Hh, mesh, texs, "shadow")
    parts = []
    for s in [full] + [T.Scene(W, Hh, mesh, texs, "shadow", band_rows=(b * Hh // n_bands, (b + 1) * Hh // n_bands))
                       for b in range(n_bands)]:
        s.clear()
        s.set_light_direction(H.light(0.6))
        s.set_camera(*H.camera(-0.4))
        s.render()
        parts.append(s.get_frame_buffer())
    out = np.zeros_like(parts[0])
    for b in range(n_bands):
        r0, r1 = b * Hh // n_bands, (b + 1) * Hh // n_bands
        out[r0:r1] = parts[1 + b][r0:r1]
    assert np.array_equal(out, parts[0])


def test_unknown_pipeline_and_alias(small_synthetic):
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    with pytest.raises(T.TinyRendererError) as e:
        T.Scene(64, 64, mesh, texs, "nope")
    assert e.value.code == -2  # TR_E_UNKNOWN_PIPELINE, shader.rs:108
    T.Scene(64, 64, mesh, texs, "true_normal").close()


def test_full_size_properties_4096(synthetic):
    """BASELINE full size (4096x4096): size-independent properties instead of an oracle frame:
    the lit-pixel mask and z are identical across single-pass pipelines that share cull +
    transform; unlit pixels hold the cleared values; the frame is idempotent."""
    import tiny_renderer_amd as T
    mesh, texs = synthetic
    zs, masks = [], []
    for pipe in ("default", "phong", "darboux"):
        s = T.Scene(4096, 4096, mesh, texs, pipe, winner_tap=True)
        s.clear()
        s.set_light_direction(H.light(0.0))
        s.set_camera(*H.camera(0.0))
        s.render()
        z = s.read_z_f32()
        w = s.read_winner_u32()
        fb = s.get_frame_buffer()
        lit = w != 0xFFFFFFFF
        assert np.all(z.view(np.uint32)[~lit] == 0xFF7FFFFF)
        assert np.all(fb[::-1][~lit] == 0)
        zs.append(z)
        masks.append(w)
        s.close()
    assert np.array_equal(zs[0].view(np.uint32), zs[1].view(np.uint32))
    assert np.array_equal(zs[1].view(np.uint32), zs[2].view(np.uint32))
    assert np.array_equal(masks[0], masks[1]) and np.array_equal(masks[1], masks[2])


@pytest.mark.parametrize("cfg", [("phong", 4096, 1), ("darboux", 4096, 1), ("shadow", 4096, 1), ("specular", 8192, 8)])
def test_baseline_configs_at_full_size(diablo, cfg):
    """BASELINE.json configs[2..4] and the metric's own workload (diablo / phong / 4096^2) compared with
    the oracle's complete frame, z and winner at full size (the oracle needs 0.1-1.7 s per frame)."""
    pipe, size, grid = cfg
    import tiny_renderer_amd as T
    mesh, texs = diablo
    if grid > 1:
        mesh = T.instanced_grid(mesh, grid)
    gpu, cpu = render_pair(size, size, mesh, texs, pipe, 0.0, 0.0)
    assert_parity(gpu, cpu, pipe)
    gpu.close()


@pytest.mark.parametrize("pipe", ["normal_map", "specular"])
def test_lit_texel_path_on_small_frames(synthetic, pipe, monkeypatch):
    """The normal-map and specular closures are functions of the texel alone, so a scene whose frame has many more pixels
    than its images have texels runs them ONCE PER TEXEL and frame (k_lit) and lets the fragment stage fetch the result
    (FS_LIT) -- by default from sixteen pixels per texel (4096^2 frames of the reference's 1024^2 images:
    test_baseline_configs_at_full_size covers the x64 grid at 8192^2).  Forced here on a small frame, whose images are
    also not a multiple of the 8x4 / 4x2 blocks: single frames with a moving camera and light, an accumulating render,
    and a group call, all against the oracle; and the kernel really runs."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    monkeypatch.setenv("TR_LIT", "1")
    mesh, texs = synthetic
    texs = [np.ascontiguousarray(t[:250, :203]) for t in texs]
    W, Hh = 520, 390
    gpu = T.Scene(W, Hh, mesh, texs, pipe, winner_tap=True)
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    gpu.profile_enable(True)
    for i, (ca, la, clear) in enumerate([(0.3, 0.2, True), (1.4, -0.6, True), (2.0, 0.9, False), (0.1, 2.2, True)]):
        for s_ in (gpu, cpu):
            if clear:
                s_.clear()
            s_.set_light_direction(H.light(la))
            s_.set_camera(*H.camera(ca))
        assert cpu.render() == 0
        gpu.render()
        assert_parity(gpu, cpu, pipe)
    assert gpu.profile_read()["k_lit"]["launches"] == 4
    gpu.close()
    # a group call (no winner tap: the fused launches)
    gpu = T.Scene(W, Hh, mesh, texs, pipe, frames_per_launch=4)
    p = np.zeros((6, 12), np.float32)
    for i in range(6):
        p[i, 0:3] = H.light(0.4 * i - 1.0)
        f, a, u = H.camera(0.7 * i)
        p[i, 3:6], p[i, 6:9], p[i, 9:12] = f, a, u
    gpu.render_frames(p)
    for back in range(gpu.frames_kept()):
        gpu.select_frame(back)
        q = p[5 - back]
        cpu.clear(); cpu.set_light_direction(q[0:3]); cpu.set_camera(q[3:6], q[6:9], q[9:12])
        assert cpu.render() == 0
        assert np.array_equal(gpu.read_z_f32().view(np.uint32), cpu.z_f32().view(np.uint32))
        fg, fo = gpu.get_frame_buffer(), cpu.get_frame_buffer()
        if pipe in EXACT or specular_exact():
            assert np.array_equal(fg, fo), "frame -%d" % back
        else:
            assert np.abs(fg.astype(np.int32) - fo.astype(np.int32)).max() <= 1  # tolerance: 1 LSB (device powf)
    gpu.close()


@pytest.mark.parametrize("pipe", ["phong", "shadow"])
def test_bin_overflow_grows_and_rerenders(synthetic, pipe):
    """More (polygon, tile) pairs in a pass than the record pool holds (a pool of 64, far below the automatic
    size): the library grows the pools and renders the frame again, transparently for a frame that started
    from clear()."""
    mesh, texs = synthetic  # 5 022 polygons on a 256x256 frame: hundreds per 128x16 tile
    gpu, cpu = render_pair(256, 256, mesh, texs, pipe, 0.3, 0.2, bin_capacity=64)
    assert_parity(gpu, cpu, pipe)
    # and the grown bins serve the next frame directly
    for s in (gpu, cpu):
        s.clear()
        s.set_camera(*H.camera(1.3))
        s.render()
    assert_parity(gpu, cpu, pipe)


@pytest.mark.parametrize("pipe", ["phong", "shadow"])
def test_many_polygons_in_one_tile_render_the_first_time(built, pipe):
    """No tile has a capacity: every tile gets exactly the records its polygons need from the pass's pool
    (k_setup counts, k_order places, k_bin fills).  A 20 088-polygon sphere shrunk to a dozen pixels in the middle
    of a 512x512 frame -- the four 128x16 tiles that meet there get ten times the 256 records that used to be a
    bin, each -- renders with the default options in ONE launch of each kernel per pass (a second tile-kernel
    launch would be the frame rendered again after an overflow), bit-identical to the oracle."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = T.synthetic_scene(n_lat=62, n_lon=162, tex_size=256)
    mesh = dict(mesh, pos=(mesh["pos"] * np.float32(0.04)).astype(np.float32))
    W = Hh = 512
    gpu = T.Scene(W, Hh, mesh, texs, pipe, winner_tap=True)
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    gpu.profile_enable(True)
    for s_ in (gpu, cpu):
        s_.clear()
        s_.set_light_direction(H.light(0.2))
        s_.set_camera(*H.camera(0.3))
    assert cpu.render() == 0
    gpu.render()
    gpu.sync()
    # how many polygons the busiest tile holds: winners alone show thousands of distinct polygons in a few tiles
    win = cpu.winner_u32()
    drawn = np.unique(win[win != 0xFFFFFFFF]).size
    ys, xs = np.nonzero(win != 0xFFFFFFFF)
    tiles = {(int(x) // 128, int(y) // 16) for x, y in zip(xs, ys)}
    assert len(tiles) <= 6 and mesh["idx"].shape[0] // 2 // len(tiles) > 1500, (len(tiles), drawn)
    prof = gpu.profile_read()
    n_passes = 2 if pipe == "shadow" else 1
    assert sum(v["launches"] for k, v in prof.items() if k.startswith("k_tile")) == n_passes, prof  # (the depth pass has its own entry)
    assert prof["k_bin"]["launches"] == n_passes, prof
    assert_parity(gpu, cpu, pipe)
    gpu.close()


def test_shadow_fetch_through_flags_on_the_device(built):
    """The DEVICE's shadow_fetch through the shadow buffer's fast-clear flags (the form that once took the flag's
    tile from (x, y) and faulted: a row k * 2^32 / W + r wraps, as upstream, to a valid flat index) against the
    plain lookup in the materialised buffer: coordinates inside the frame, columns beyond the row, rows that wrap
    around 2^32 back into the buffer, out of range, negative, NaN (shader.rs:774-778, 909-912, 932-935)."""
    import tiny_renderer_amd as T
    from tests import emul_bind as E
    L = T.load_library()
    rng = np.random.default_rng(11)
    for W, Hh in ((512, 512), (640, 100), (130, 70), (4096, 64)):
        ntx, nty = (W + 127) // 128, (Hh + 15) // 16
        plain = rng.standard_normal(W * Hh).astype(np.float32)
        flags = (rng.random(ntx * nty) < 0.5).astype(np.uint32) * np.uint32(0xFFFFFFFF)
        tile_of = (np.arange(Hh)[:, None] // 16) * ntx + (np.arange(W)[None, :] // 128)
        clean = flags[tile_of].astype(bool).reshape(-1)
        plain[clean] = np.float32(np.finfo(np.float32).min)
        stale = plain.copy()
        stale[clean] = np.float32(123.0)
        xs = [rng.uniform(-3, W + 3, 4000), rng.uniform(W, 40 * W, 2000), rng.uniform(0, W, 3000), rng.uniform(0, W, 500)]
        ys = [rng.uniform(-3, Hh + 3, 4000), rng.uniform(0, Hh / 2, 2000),
              (rng.integers(1, 64, 3000) * (2.0 ** 32 / W)) + rng.integers(0, Hh, 3000),   # wraps around 2^32
              rng.uniform(Hh, 1e9, 500)]
        x = np.ascontiguousarray(np.concatenate(xs + [[np.nan, 0.0, -1e30, 1e30]]).astype(np.float32))
        y = np.ascontiguousarray(np.concatenate(ys + [[0.0, np.nan, 5.0, 1e30]]).astype(np.float32))
        n = len(x)
        out = [np.zeros(n, np.uint32) for _ in range(4)]
        T._lib.check(L.tr_selftest_shadow_fetch(0, W, Hh, plain.ctypes.data, stale.ctypes.data, flags.ctypes.data, n,
                                                x.ctypes.data, y.ctypes.data, *[o.ctypes.data for o in out]))
        assert np.array_equal(out[0], out[1]), (W, Hh, int((out[0] != out[1]).sum()))
        assert np.array_equal(out[2], out[3])
        assert (out[2] == 0).sum() > 5000 and (out[2] != 0).sum() > 100   # lookups inside the buffer, and flagged ones
        assert (out[2][6000:9000] == 0).sum() > 100                        # the wrapped rows do land inside the buffer
        # ... and the host's emulation of the same function agrees with itself on these inputs
        assert E.lib().tr_emul_shadow_fetch_mismatches(plain.ctypes.data, stale.ctypes.data, flags.ctypes.data, W, Hh,
                                                       x.ctypes.data, y.ctypes.data, n) == 0


def test_device_math_selftest(built):
    """The instruction-level substitutions on the device: v_cvt_{u32,i32}_f32 as Rust `as` casts
    and the shared-reciprocal division against the device's own '/' and the host's."""
    import ctypes as C
    import tiny_renderer_amd as T
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 0.999, -0.999, 254.999, 255.0, 255.5, 256.0, 1e10,
                        -1e10, 4294967040.0, 4294967296.0, 2147483520.0, 2147483648.0, -2147483648.0, -2147483904.0,
                        1e-40, -1e-40, 3.5, -3.5], np.float32)
    x = np.concatenate([special, np.trunc((rng.standard_normal(100000) * 2.0 ** rng.integers(0, 40, 100000)))
                        .astype(np.float32)])
    d = np.concatenate([np.ones(special.size, np.float32),
                        np.trunc(rng.integers(1, 1 << 26, 100000).astype(np.float32)) * rng.choice([-1, 1], 100000)
                        ]).astype(np.float32)
    n = x.size
    u32, i32, u8 = np.zeros(n, np.uint32), np.zeros(n, np.int32), np.zeros(n, np.uint32)
    dv, dref = np.zeros(n, np.float32), np.zeros(n, np.float32)
    fp = C.POINTER(C.c_float)
    T._lib.check(T.load_library().tr_selftest_device_math(0, x.ctypes.data_as(fp), d.ctypes.data_as(fp), n,
                                                         u32.ctypes.data, i32.ctypes.data, u8.ctypes.data,
                                                         dv.ctypes.data, dref.ctypes.data))
    L = O.lib()
    for k in range(special.size + 2000):
        assert int(u32[k]) == L.tro_f32_to_u32(float(x[k])), x[k]
        assert int(i32[k]) == L.tro_f32_to_i32(float(x[k])), x[k]
        assert int(u8[k]) == L.tro_f32_to_u8(float(x[k])), x[k]
    fin = np.isfinite(x)
    with np.errstate(all="ignore"):
        host = (x / d).astype(np.float32)
    assert np.array_equal(dref[fin].view(np.uint32), host[fin].view(np.uint32))   # device '/' is IEEE
    assert np.array_equal(dv[fin].view(np.uint32), host[fin].view(np.uint32))     # shared reciprocal too


@pytest.mark.parametrize("pipe", ["phong", "occlusion"])
def test_every_tile_heavy(synthetic, pipe):
    """Every tile holds many polygons: the work list consists of its heaviest buckets only."""
    mesh, texs = synthetic
    gpu, cpu = render_pair(256, 64, mesh, texs, pipe, 0.0, 0.0)
    assert_parity(gpu, cpu, pipe)
    gpu, cpu = render_pair(384, 200, mesh, texs, pipe, 0.9, -0.4)
    assert_parity(gpu, cpu, pipe)


@pytest.mark.parametrize("pipe", ["phong", "shadow"])
def test_frames_in_flight(small_synthetic, pipe):
    """Consecutive frames are pipelined (the setup kernel of frame f+1 overlaps the tile kernel of
    frame f, with triple-buffered counters and double-buffered bins): a burst of frames with a
    moving camera and no readback in between must end with exactly the last frame."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh = 640, 400
    gpu = T.Scene(W, Hh, mesh, texs, pipe, winner_tap=True)
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    for burst in (1, 2, 3, 7):
        for f in range(burst):
            gpu.clear()
            gpu.set_light_direction(H.light(0.1 * f - 0.3 * burst))
            gpu.set_camera(*H.camera(0.37 * f + burst))
            gpu.render()
        cpu.clear()
        cpu.set_light_direction(H.light(0.1 * (burst - 1) - 0.3 * burst))
        cpu.set_camera(*H.camera(0.37 * (burst - 1) + burst))
        assert cpu.render() == 0
        assert_parity(gpu, cpu, pipe)


@pytest.mark.parametrize("seed", range(6))
def test_random_api_sequences(small_synthetic, seed):
    """Random interleavings of the whole API surface (clear / setters / render / every getter / the
    asynchronous read-back / sync) against the oracle driven by the same calls: renders are held back
    and submitted in batches, clears are lazy, z lives behind per-tile flags -- none of which may show."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    rng = np.random.default_rng(700 + seed)
    # (occlusion is left out: a random light can hit the antiparallel case in which the reference panics)
    pipe = ["phong", "shadow", "default", "normal_map", "darboux", "specular"][seed % 6]
    W, Hh = [(320, 200), (1100, 700), (640, 64)][seed % 3]
    gpu = T.Scene(W, Hh, mesh, texs, pipe, winner_tap=True, tile_waves=[0, 4, 8, 16][seed % 4], tile_mode=[0, 2, 1][seed % 3])
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    pinned = gpu.pinned_frame()
    pending_async = False
    for step in range(45):
        op = rng.choice(["clear", "camera", "light", "render", "render", "render", "frame", "z", "zview", "shadow",
                         "winner", "async", "sync"])
        if op == "clear":
            gpu.clear(); cpu.clear()
        elif op == "camera":
            a = float(rng.uniform(-3.0, 3.0))
            off = [float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.3, 0.3))]
            frm = [float(np.sin(np.float32(a))) + off[0], off[1], float(np.cos(np.float32(a)))]
            for s in (gpu, cpu):
                s.set_camera(frm, [off[0], off[1], 0.0], [0.0, 1.0, 0.0])
        elif op == "light":
            v = H.light(float(rng.uniform(-2.0, 2.0)))
            gpu.set_light_direction(v); cpu.set_light_direction(v)
        elif op == "render":
            gpu.render()
            assert cpu.render() == 0
        elif op == "frame":
            assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer()), (step, op)
        elif op == "z":
            assert np.array_equal(gpu.read_z_f32().view(np.uint32), cpu.z_f32().view(np.uint32)), (step, op)
        elif op == "zview":
            assert np.array_equal(gpu.get_z_buffer(), cpu.get_z_buffer()), (step, op)
        elif op == "shadow":
            assert np.array_equal(gpu.read_shadow_f32().view(np.uint32), cpu.shadow_f32().view(np.uint32)), (step, op)
        elif op == "winner":
            assert np.array_equal(gpu.read_winner_u32(), cpu.winner_u32()), (step, op)
        elif op == "async":
            gpu.get_frame_buffer_async(pinned)
            expect = cpu.get_frame_buffer()
            pending_async = True
        if op == "sync" or (pending_async and op in ("render", "clear")):
            if pending_async:   # a later render may not disturb a queued read-back
                assert gpu.sync() == 0
                assert np.array_equal(pinned, expect), (step, "async")
                pending_async = False
            elif op == "sync":
                assert gpu.sync() == 0
    assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer())
    assert np.array_equal(gpu.read_z_f32().view(np.uint32), cpu.z_f32().view(np.uint32))
    gpu.close()


@pytest.mark.parametrize("first", ["zview", "shadowview", "z", "winner", "frame"])
def test_bin_overflow_first_getter(synthetic, first):
    """Whatever getter is the first synchronising call after a render that overflowed the bins must
    see the frame rendered again with grown bins, not a view derived from the truncated one."""
    mesh, texs = synthetic
    gpu, cpu = render_pair(256, 256, mesh, texs, "shadow", 0.3, 0.2, bin_capacity=64)
    if first == "zview":
        assert np.array_equal(gpu.get_z_buffer(), cpu.get_z_buffer())
    elif first == "shadowview":
        assert np.array_equal(gpu.get_shadow_buffer(), cpu.get_shadow_buffer())
    elif first == "z":
        assert np.array_equal(gpu.read_z_f32().view(np.uint32), cpu.z_f32().view(np.uint32))
    elif first == "winner":
        assert np.array_equal(gpu.read_winner_u32(), cpu.winner_u32())
    assert_parity(gpu, cpu, "shadow")
    gpu.close()


def test_bin_overflow_with_reads_in_flight(synthetic):
    """Frames whose bins overflowed and that were already copied out asynchronously cannot be
    repaired behind the caller's back: sync must say TR_E_BIN_OVERFLOW (never TR_OK with a
    truncated copy); after it the bins fit and the same sequence gives three correct frames."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = synthetic
    W, Hh = 256, 256
    gpu = T.Scene(W, Hh, mesh, texs, "phong", bin_capacity=64)
    cpu = O.Scene(W, Hh, mesh, texs, "phong")
    outs = [gpu.pinned_frame() for _ in range(3)]

    def burst():
        for k, out in enumerate(outs):
            gpu.clear()
            gpu.set_light_direction(H.light(0.3 * k))
            gpu.set_camera(*H.camera(0.9 * k))
            gpu.render()
            gpu.get_frame_buffer_async(out)

    burst()
    with pytest.raises(T.TinyRendererError) as e:
        gpu.sync()
    assert e.value.code == -9  # TR_E_BIN_OVERFLOW
    burst()
    assert gpu.sync() == 0
    for k, out in enumerate(outs):
        cpu.clear()
        cpu.set_light_direction(H.light(0.3 * k))
        cpu.set_camera(*H.camera(0.9 * k))
        assert cpu.render() == 0
        assert np.array_equal(out, cpu.get_frame_buffer()), "frame %d" % k
    # frames that nobody observed are still repaired silently: only the last one can be seen
    gpu2 = T.Scene(W, Hh, mesh, texs, "phong", bin_capacity=64)
    for k in range(3):
        gpu2.clear()
        gpu2.set_light_direction(H.light(0.3 * k))
        gpu2.set_camera(*H.camera(0.9 * k))
        gpu2.render()
    assert gpu2.sync() == 0
    assert np.array_equal(gpu2.get_frame_buffer(), cpu.get_frame_buffer())
    gpu.close()
    gpu2.close()


def test_caller_stream_consumes_frames_without_sync(small_synthetic):
    """The multi-GPU bench's pattern on one GPU: the scene renders on a caller-provided (torch) side
    stream into caller-provided, double-buffered frame tensors, and every frame is consumed ON THAT
    STREAM (a device copy standing in for the all-gather) without any host synchronisation in
    between; the camera moves, so a consumer that ran ahead of the render would keep a stale frame."""
    import torch
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh, n = 1024, 640, 12
    side = torch.cuda.Stream()
    assert side.cuda_stream != 0
    bufs = [torch.zeros(Hh * W * 3, dtype=torch.uint8, device="cuda") for _ in range(2)]
    kept = torch.zeros(n, Hh * W * 3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu = T.Scene(W, Hh, mesh, texs, "phong", stream=side.cuda_stream, frame_buffer_device=bufs[0].data_ptr())
    with torch.cuda.stream(side):
        for f in range(n):
            gpu.set_frame_buffer_device(bufs[f % 2].data_ptr())
            gpu.clear()
            gpu.set_light_direction(H.light(0.1 * f))
            gpu.set_camera(*H.camera(0.45 * f))
            gpu.render()
            kept[f].copy_(bufs[f % 2], non_blocking=True)
    assert gpu.sync() == 0
    torch.cuda.synchronize()
    cpu = O.Scene(W, Hh, mesh, texs, "phong")
    for f in range(n):
        cpu.clear()
        cpu.set_light_direction(H.light(0.1 * f))
        cpu.set_camera(*H.camera(0.45 * f))
        assert cpu.render() == 0
        assert np.array_equal(kept[f].cpu().numpy().reshape(Hh, W, 3), cpu.get_frame_buffer()), "frame %d" % f
    gpu.close()


def test_a_callers_buffer_rewritten_behind_the_scenes_back(small_synthetic):
    """Colour-clean flags of a CALLER's buffer are forgotten whenever the buffer is handed over again: a cleared
    render then produces every pixel of the frame, whatever wrote the buffer in between (a post-process, a memset,
    an allocator giving the address to another tensor).  With TR_OPT_TRUST_FRAME_BUFFERS the caller promises not
    to, and the empty tiles' zeros are not stored a second time."""
    import torch
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh = 640, 384
    cpu = O.Scene(W, Hh, mesh, texs, "phong")
    cpu.clear(), cpu.set_light_direction(H.light(0.2)), cpu.set_camera(*H.camera(0.3)), cpu.render()
    want = cpu.get_frame_buffer()
    assert (want == 0).all(-1).mean() > 0.3          # a good part of the frame is empty tiles
    for trust in (False, True):
        buf = torch.zeros(Hh * W * 3, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        gpu = T.Scene(W, Hh, mesh, texs, "phong", frame_buffer_device=buf.data_ptr(), trust_frame_buffers=trust)
        for rep in range(3):
            gpu.set_frame_buffer_device(buf.data_ptr())
            gpu.clear(), gpu.set_light_direction(H.light(0.2)), gpu.set_camera(*H.camera(0.3)), gpu.render()
            assert gpu.sync() == 0
            torch.cuda.synchronize()
            got = buf.cpu().numpy().reshape(Hh, W, 3)
            if rep < 2 or not trust:
                assert np.array_equal(got, want), "trust %s, render %d" % (trust, rep)
            else:
                # the promise was broken below: the scene did not store the empty tiles again (that is the saving)
                assert (got == 77).any() and np.array_equal(got[want.any(-1)], want[want.any(-1)])
            if rep == 1:
                buf.fill_(77)                         # somebody else writes the buffer
                torch.cuda.synchronize()
        gpu.close()
    cpu.close()


def test_bin_overflow_in_an_older_frame_of_a_call_with_callers_buffers(synthetic):
    """tr_scene_render_frames with more frames than a launch holds, every frame into a buffer of the caller's, bins
    too small: the frames older than the last group cannot be rendered again, so the sync says TR_E_BIN_OVERFLOW
    (never TR_OK with truncated frames in the caller's hands); rendered again with the grown bins, every buffer
    holds its frame."""
    import torch
    import tiny_renderer_amd as T
    mesh, texs = synthetic
    W, Hh, n = 512, 384, 6
    p = np.zeros((n, 12), np.float32)
    for f in range(n):
        p[f, 0:3] = H.light(0.1 * f)
        p[f, 3:6], p[f, 6:9], p[f, 9:12] = H.camera(0.4 * f)
    bufs = [torch.zeros(Hh * W * 3, dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    gpu = T.Scene(W, Hh, mesh, texs, "phong", bin_capacity=64, frames_per_launch=2)
    gpu.render_frames(p, [b.data_ptr() for b in bufs])
    with pytest.raises(T.TinyRendererError) as e:
        gpu.sync()
    assert e.value.code == -9
    gpu.render_frames(p, [b.data_ptr() for b in bufs])
    assert gpu.sync() == 0
    torch.cuda.synchronize()
    for f in range(n):
        err, s = H_oracle(W, Hh, mesh, texs, "phong", p[f])
        assert np.array_equal(bufs[f].cpu().numpy().reshape(Hh, W, 3), s), "frame %d" % f
    gpu.close()


def H_oracle(W, Hh, mesh, texs, pipe, q):
    from oracle import oracle as O
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    cpu.clear(), cpu.set_light_direction(q[0:3]), cpu.set_camera(q[3:6], q[6:9], q[9:12])
    err = cpu.render()
    fb = cpu.get_frame_buffer()
    cpu.close()
    return err, fb


def test_cli_time_based_loop(synthetic, tmp_path):
    """`--seconds`: the reference's time-based frame loop (app.rs:166-247: angles advance by 3 rad/s times the frame
    time, `FPS --- n` about once a second), every frame read back; the written frame is a frame of the orbit."""
    import os
    import subprocess
    import sys
    from PIL import Image
    out = str(tmp_path / "loop.png")
    r = subprocess.run([sys.executable, "-m", "tiny_renderer_amd.cli", "--synthetic", "-s", "phong", "--width", "320",
                        "--height", "240", "--seconds", "1.3", "--out", out], cwd=H.REPO, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    fps = [int(l.split("---")[1]) for l in r.stdout.splitlines() if l.startswith("FPS ---")]
    assert fps and all(f > 20 for f in fps), r.stdout[-500:]
    got = np.array(Image.open(out).convert("RGB"))
    assert got.shape == (240, 320, 3) and got.any()


def test_pair_rcp_sqrt_exhaustive(built):
    """rcp2 / sqrt2 (tr_pk.h: v_rcp_f32 / v_rsq_f32 + fused residual corrections in packed arithmetic) against the
    compiler's correctly rounded 1.0f / x and sqrtf for EVERY f32 of the ranges the two-pixel closures'
    guard admits (|d| in [2^-40, 2^41] -> exponents -42..42 tested; sums of squares in [2^-80, 2^82] ->
    -84..84): 7.1e8 + 1.4e9 arguments, none may differ."""
    import ctypes as C
    import tiny_renderer_amd as T
    L = T.load_library()
    # which = 2: pack_u8 (v_cvt_pk_u8_f32, the colour channels' `as u8` with its byte insertion) against f32_to_u8 for
    # EVERY f32 -- subnormals, infinities and NaNs of both signs included (biased exponents 0..255, x and -x)
    for which, lo, hi in ((0, -42, 42), (1, -84, 84), (2, -127, 128)):
        nt, nb = C.c_uint64(), C.c_uint64()
        bits = (C.c_uint32 * 16)()
        T._lib.check(L.tr_selftest_device_unary(0, which, lo, hi, C.byref(nt), C.byref(nb), bits))
        assert nt.value == (hi - lo + 1) << 23
        assert nb.value == 0, [hex(b) for b in bits if b]


def _oracle_frame(W, Hh, mesh, texs, pipe, ca, la):
    from oracle import oracle as O
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    cpu.clear()
    cpu.set_light_direction(H.light(la))
    cpu.set_camera(*H.camera(ca))
    assert cpu.render() == 0
    return cpu.get_frame_buffer()


@pytest.mark.parametrize("pipe,ext", [("phong", "png"), ("shadow", "tga"), ("darboux", "ppm")])
def test_cli_writes_the_oracles_frame(synthetic, tmp_path, pipe, ext):
    """The headless CLI (main.rs:9-39's -p / -s plus explicit size and angles): the file it writes is the
    oracle's frame, through all three writers."""
    from PIL import Image
    import tiny_renderer_amd as T
    from tiny_renderer_amd import cli
    mesh, texs = synthetic
    out = str(tmp_path / ("frame." + ext))
    rc = cli.main(["--synthetic", "-s", pipe, "--width", "640", "--height", "360", "--camera-angle", "0.4",
                   "--light-angle", "-0.3", "--out", out])
    assert rc == 0
    got = T.load_tga(out) if ext == "tga" else np.array(Image.open(out).convert("RGB"))
    assert np.array_equal(got, _oracle_frame(640, 360, mesh, texs, pipe, 0.4, -0.3))


def test_cli_many_frames_uses_frame_groups(synthetic, tmp_path):
    """`--frames 7` on one GPU goes through tr_scene_render_frames (camera orbiting over the frames); the
    file written is the oracle's last frame, and the z view of the same run is the oracle's z view."""
    from PIL import Image
    from tiny_renderer_amd import cli
    from oracle import oracle as O
    mesh, texs = synthetic
    angle = float(np.float32(0.2 + 2.0 * np.pi * 6 / 7))
    for view in ("frame", "z"):
        out = str(tmp_path / ("orbit_%s.png" % view))
        assert cli.main(["--synthetic", "-s", "normal_map", "--width", "500", "--height", "300", "--frames", "7",
                         "--camera-angle", "0.2", "--light-angle", "0.6", "--view", view, "--out", out]) == 0
        got = np.array(Image.open(out).convert("RGB"))
        if view == "frame":
            assert np.array_equal(got, _oracle_frame(500, 300, mesh, texs, "normal_map", angle, 0.6))
        else:
            s = O.Scene(500, 300, mesh, texs, "normal_map")
            s.clear(), s.set_light_direction(H.light(0.6)), s.set_camera(*H.camera(angle)), s.render()
            assert np.array_equal(got, s.get_z_buffer())


def test_cli_sharded_single_rank(synthetic, tmp_path):
    """`--gpus` code path of the CLI (ShardedScene: band scene on a torch side stream, two frame tensors,
    RCCL all-gather on a second stream) with a one-rank process group, in its own process; several
    frames with a moving camera, the last one must be the oracle's."""
    import os
    import subprocess
    import sys
    from PIL import Image
    mesh, texs = synthetic
    out = str(tmp_path / "sharded.png")
    env = dict(os.environ, TR_CLI_FORCE_DIST="1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "tiny_renderer_amd.cli", "--synthetic", "-s", "shadow", "--width", "512",
                        "--height", "384", "--frames", "5", "--light-angle", "0.5", "--gpus", "1", "--out", out],
                       env=env, cwd=H.REPO, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.array(Image.open(out).convert("RGB"))
    angle = float(np.float32(2.0 * np.pi * 4 / 5))
    assert np.array_equal(got, _oracle_frame(512, 384, mesh, texs, "shadow", angle, 0.5))


def test_sharded_scene_render_frames(built):
    """ShardedScene.render_frames (a rank's band of a GROUP of frames per kernel launch, the bands exchanged frame by
    frame on the second stream) in a one-rank RCCL group, in its own process: tests/sharded_groups_worker.py."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MASTER_PORT="29541")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "tests", "sharded_groups_worker.py")], env=env, cwd=H.REPO,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("exchange", ["peer", "peer-sparse"])
def test_sharded_scene_two_ranks_one_gpu(built, exchange):
    """The multi-rank path with a REAL peer: two rank processes of a ShardedScene share this box's one GPU (RCCL refuses two
    ranks on one device, so the bands travel through the library's peer transport -- dense: pulled by the DMA engines;
    sparse: the product's k_push_tiles) through render, render_frames, the collective repair of a bin overflow that only
    ONE rank had, and an error in ONE band that every rank must raise; the assembled frame is compared with the oracle's
    on BOTH ranks: tests/sharded_ranks_worker.py."""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(H.REPO, "tests", "sharded_ranks_worker.py"), exchange],
                       env=env, cwd=H.REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rank 0 OK" in r.stdout and "rank 1 OK" in r.stdout, (r.stdout[-1500:] + r.stderr[-4000:])


@pytest.mark.parametrize("exchange", ["rccl", "librccl"])
def test_bench_sharded_path_over_rccl_with_one_rank(built, exchange):
    """bench.py's N > 1 code path as the driver's multi-GPU run takes it -- process group, ShardedScene over torch's RCCL
    collective (the default) or the library's own RCCL communicator, frame groups, per-rank timings, the scale_config
    leg -- with the ONE rank RCCL allows on a one-GPU box (TR_BENCH_FORCE_DIST / TR_BENCH_FORCE_SCALE).  Two real ranks
    run the same render side through the peer transports: test_sharded_scene_two_ranks_one_gpu."""
    import json
    import os
    import subprocess
    import sys
    env = dict(os.environ, TR_BENCH_FORCE_DIST="1", TR_BENCH_FORCE_SCALE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MASTER_PORT=str(29560 + (0 if exchange == "rccl" else 1)))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "bench.py"), "--gpus", "1", "--exchange", exchange, "--size", "1024",
                        "--steps", "12", "--warmup", "4", "--no-cpu", "--scale-size", "1024", "--scale-grid", "2", "--scale-steps", "8"],
                       env=env, cwd=H.REPO, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["parity_vs_oracle"]["ok"] and j["group_ranks"] == 1 and "RCCL all-gather" in j["config"]["sharding"]
    assert j["per_rank"][0]["band_rows"] == [0, 1024] and j["per_rank"][0]["render_us"] and j["per_rank"][0]["gather_us"]
    sc = j["scale_config"]
    assert sc["parity_vs_oracle"]["ok"] and "x4 grid" in sc["workload"] and sc["exchange"] == ("torch" if exchange == "rccl" else "rccl")


def test_two_ranks_on_one_gpu_under_rccl_says_so(built):
    """`bench.py --gpus 2` with both ranks on this box's one GPU and the default (RCCL) exchange: one clear sentence, not
    RCCL's "Duplicate GPU detected" from the bottom of a traceback."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, TR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "bench.py"), "--gpus", "2", "--size", "512", "--steps", "4", "--warmup", "2",
                        "--no-cpu"], env=env, cwd=H.REPO, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "RCCL needs a device per rank" in r.stderr and "--exchange peer" in r.stderr, r.stderr[-2000:]
    assert "Duplicate GPU" not in r.stderr


def test_rccl_exchange_behind_the_c_abi(built):
    """The RCCL backend of tr_exchange_* (the library owns the communicator: SURVEY.md 8b / 8e) with one rank, in
    its own process and without torch: tests/rccl_exchange_worker.py."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "tests", "rccl_exchange_worker.py")], env=env, cwd=H.REPO,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("sparse", [False, True])
def test_peer_exchange_two_processes_one_gpu(diablo, sparse):
    """The library's own frame exchange (tr_exchange_*: HIP IPC mapped frame slots, concurrent DMA-engine
    band copies, generation flags) with TWO rank processes sharing this box's one GPU: bench.py's N = 2
    code path end to end -- band scenes, two frame slots, exchange on a second stream, a moving-camera
    leg -- and its closing comparison of the assembled frame with the oracle (rank 0 exits non-zero on a
    mismatch, e.g. a band that arrived late or stale).  Cross-GPU coherence cannot show on one device;
    that run is the driver's.  `sparse`: the band goes tile by tile (tr_exchange_all_gather_tiles) and the tiles that
    are the cleared colour on both sides stay home -- the assembled frames must be the same, with fewer bytes pushed."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, TR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "bench.py"), "--gpus", "2", "--exchange", "peer", "--size", "1024",
                        "--steps", "40", "--warmup", "5", "--no-cpu", "--scale-size", "2048", "--scale-grid", "4", "--scale-steps", "12"]
                       + (["--sparse"] if sparse else []), env=env, cwd=H.REPO, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["group_ranks"] == 2 and j["parity_vs_oracle"]["ok"]
    assert "peer-to-peer" in j["config"]["sharding"] and len(j["per_rank"]) == 2
    assert j["per_rank"][0]["band_rows"] == [0, 512] and j["per_rank"][1]["band_rows"] == [512, 1024]
    dense = 512 * 1024 * 3
    assert j["exchange_dense_bytes_per_frame"] == dense
    if sparse:
        # the model covers the middle of the frame: most tiles of the band are the cleared colour and stay home
        assert 0 < j["exchange_bytes_per_frame"] < dense // 2, j["exchange_bytes_per_frame"]
    else:
        assert j["exchange_bytes_per_frame"] == dense
    # BASELINE configs[4]'s shape (an n x n grid of the model, specular) sharded over the same two ranks, beside the headline
    sc = j["scale_config"]
    assert sc["n_gpus"] == 2 and sc["parity_vs_oracle"]["ok"] and "x16 grid" in sc["workload"] and "2048x2048" in sc["workload"]
    assert len(sc["per_rank"]) == 2 and all(r_["render_us"] and r_["gather_us"] and r_["k_tile_us"] for r_ in sc["per_rank"])
    assert sc["exchange_bound_estimate_us"]["direct"] > 0 and sc["exchange_dense_bytes_per_frame"] == 1024 * 2048 * 3


def _exchange_rank(rank, q_in, q_out, participate):
    import ctypes as C
    import os
    os.environ["TR_EXCHANGE_TIMEOUT_MS"] = "1500"
    import torch
    import tiny_renderer_amd as T
    from tiny_renderer_amd._lib import TR_EXCHANGE_HANDLE_BYTES
    from tiny_renderer_amd.sharded import _DeviceBytes
    L = T.load_library()
    h = C.c_void_p()
    n = 1 << 20
    assert L.tr_exchange_create(0, 2, rank, 1, n, C.byref(h)) == 0
    # every rank's half of the frame is known to everybody: the dense exchange pulls (nothing but flags crosses into a peer)
    off, siz = (C.c_size_t * 2)(0, n // 2), (C.c_size_t * 2)(n // 2, n // 2)
    assert L.tr_exchange_set_ranges(h, off, siz) == 0
    slot = torch.as_tensor(_DeviceBytes(L.tr_exchange_frame(h, 0), n), device="cuda:0")
    slot[rank * (n // 2):(rank + 1) * (n // 2)] = 7 + rank          # this rank's band: sevens / eights
    torch.cuda.synchronize()
    rec = C.create_string_buffer(TR_EXCHANGE_HANDLE_BYTES)
    assert L.tr_exchange_export(h, rec) == 0
    q_out.put((rank, bytes(rec.raw)))
    records = q_in.get(timeout=120)
    assert L.tr_exchange_connect(h, b"".join(records)) == 0
    code = 0
    if participate:
        assert L.tr_exchange_all_gather(h, 0, rank * (n // 2), n // 2, None) == 0
        out = (C.c_uint8 * n)()
        code = L.tr_exchange_read(h, 0, out, n)
    q_out.put(("done", rank, code))
    q_in.get(timeout=120)   # stay alive (mappings valid) until told to leave: by now the other rank has given up
    torch.cuda.synchronize()
    mine = slot.cpu().numpy()
    other = mine[(1 - rank) * (n // 2):(2 - rank) * (n // 2)]
    # the rank that never joined: the half of ITS slot that belongs to the peer is as it was created (zeros) -- the
    # peer's copy engines wrote nothing into a slot its owner had not opened
    untouched = bool((other == 0).all()) if not participate else None
    q_out.put(("slot", rank, untouched))
    L.tr_exchange_destroy(h)


def test_peer_exchange_reports_a_missing_rank(built):
    """A rank whose peer never joins the all-gather must not hang the GPU: the device-side waits give up
    (TR_EXCHANGE_TIMEOUT_MS = 1.5 s here, ten seconds by default) and the status is TR_E_EXCHANGE -- and, the ranges
    being declared (tr_exchange_set_ranges: the pull form), the absent rank's slot is left exactly as it was."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    q_ins = [ctx.Queue(), ctx.Queue()]
    procs = [ctx.Process(target=_exchange_rank, args=(r, q_ins[r], q_out, r == 0)) for r in range(2)]
    for p in procs:
        p.start()
    recs = dict(q_out.get(timeout=180) for _ in range(2))
    for q in q_ins:
        q.put([recs[0], recs[1]])
    done = {}
    for _ in range(2):
        tag, rank, code = q_out.get(timeout=180)
        done[rank] = code
    for q in q_ins:
        q.put("leave")
    slots = {}
    for _ in range(2):
        tag, rank, untouched = q_out.get(timeout=180)
        assert tag == "slot"
        slots[rank] = untouched
    for p in procs:
        p.join(60)
    assert done[0] == -11, done    # TR_E_EXCHANGE on the rank that waited alone
    assert done[1] == 0
    assert slots[1] is True, "the absent rank's slot was written by its peer"
    assert all(p.exitcode == 0 for p in procs)
