"""Frame groups (tr_scene_render_frames): several cleared frames rendered by ONE launch of each kernel,
every frame into render targets of its own.  Frame i of a call must be bit for bit what the reference's
per-frame protocol (clear -> set_light_direction -> set_camera -> render, app.rs:170-213) gives -- i.e.
what the CPU oracle renders for that light and camera -- whatever the group size, the position of the
frame in its group, the tile layout, or what the scene did before and does afterwards.
PARITY UNPINNED upstream: the oracle is the normative restatement (oracle/tr_oracle.h)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

EXACT = ("default", "phong", "normal_map", "darboux", "shadow", "occlusion")
ALL = EXACT + ("specular",)


def specular_exact():
    import tiny_renderer_amd as T
    return bool(T.load_library().tr_specular_exact())


def params(n, cam0=0.0, cam_step=0.37, light0=-0.5, light_step=0.23):
    """[n, 12]: light, look_from, look_at, up per frame -- every frame differs from its neighbours."""
    out = np.zeros((n, 12), np.float32)
    for i in range(n):
        out[i, 0:3] = H.light(light0 + light_step * i)
        f, a, u = H.camera(cam0 + cam_step * i)
        out[i, 3:6], out[i, 6:9], out[i, 9:12] = f, a, u
    return out


def oracle_frames(W, Hh, mesh, texs, pipe, p, band=None):
    from oracle import oracle as O
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    if band is not None:
        cpu.set_output_band(*band)
    out = []
    for q in p:
        cpu.clear()
        cpu.set_light_direction(q[0:3])
        cpu.set_camera(q[3:6], q[6:9], q[9:12])
        assert cpu.render() == 0
        out.append((cpu.get_frame_buffer().copy(), cpu.z_f32().view(np.uint32).copy(),
                    cpu.shadow_f32().view(np.uint32).copy() if pipe in ("shadow", "occlusion") else None))
    return out


def check_kept(gpu, expect, pipe, n):
    """Every frame the call left behind, newest first, against the oracle's frames."""
    kept = gpu.frames_kept()
    assert kept == min(n, gpu.frames_per_launch)
    for back in range(kept):
        gpu.select_frame(back)
        fo, zo, so = expect[n - 1 - back]
        zg = gpu.read_z_f32().view(np.uint32)
        assert np.array_equal(zg, zo), "frame -%d: z bits differ at %d pixels" % (back, int((zg != zo).sum()))
        if so is not None:
            sg = gpu.read_shadow_f32().view(np.uint32)
            assert np.array_equal(sg, so), "frame -%d: shadow bits differ at %d pixels" % (back, int((sg != so).sum()))
        fg = gpu.get_frame_buffer()
        if pipe in EXACT or specular_exact():
            assert np.array_equal(fg, fo), "frame -%d: rgb differs at %d pixels" % (back, int((fg != fo).any(-1).sum()))
        else:
            assert np.abs(fg.astype(np.int32) - fo.astype(np.int32)).max() <= 1  # tolerance: 1 LSB (device powf)
        assert fo.any(), "an empty frame proves nothing"
    gpu.select_frame(0)


@pytest.mark.parametrize("pipe", ALL)
def test_group_frames_match_the_oracle(synthetic, pipe):
    import tiny_renderer_amd as T
    mesh, texs = synthetic
    W, Hh, n = 640, 480, 7
    p = params(n)
    gpu = T.Scene(W, Hh, mesh, texs, pipe, frames_per_launch=4)   # groups of 4 + 3
    assert gpu.frames_per_launch == 4
    gpu.render_frames(p)
    assert gpu.sync() == 0
    check_kept(gpu, oracle_frames(W, Hh, mesh, texs, pipe, p), pipe, n)
    gpu.close()


@pytest.mark.parametrize("fpl", [1, 2, 3, 8, 16, 32, 0])
@pytest.mark.parametrize("n", [1, 5, 16, 19, 70])
def test_group_sizes_and_counts(small_synthetic, fpl, n):
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W, Hh = 520, 300
    p = params(n, cam_step=0.21)
    gpu = T.Scene(W, Hh, mesh, texs, "phong", frames_per_launch=fpl)
    gpu.render_frames(p)
    check_kept(gpu, oracle_frames(W, Hh, mesh, texs, "phong", p), "phong", n)
    gpu.close()


@pytest.mark.parametrize("n,launches", [(70, 4), (20, 3), (9, 2)])
def test_group_sizes_of_long_and_short_calls(small_synthetic, n, launches):
    """Automatic group sizes at 4096^2 (the usual group: 4 frames).  A call of sixteen groups or more grows its groups
    (70 frames: 4 + 16 + 32 + 18), a shorter one starts with the usual group and doubles up to twelve (20 frames:
    4 + 8 + 8; 9 frames: 4 + 5) -- and whatever the sizes, the frames the call leaves behind are the oracle's."""
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W = Hh = 4096
    p = params(n, cam_step=0.09)
    gpu = T.Scene(W, Hh, mesh, texs, "phong")
    assert gpu.frames_per_launch == 4
    gpu.profile_enable(True)
    gpu.render_frames(p)
    assert gpu.sync() == 0
    prof = gpu.profile_read()
    gpu.profile_enable(False)
    assert prof["k_tile"]["launches"] == launches and prof["k_tile"]["frames"] == n, prof
    # (the oracle renders only the frames that are left: 0.1 s each at this size)
    kept = gpu.frames_kept()
    assert kept == min(n, 4)
    expect = [None] * (n - kept) + oracle_frames(W, Hh, mesh, texs, "phong", p[n - kept:])
    check_kept(gpu, expect, "phong", n)
    gpu.close()


@pytest.mark.parametrize("pipe", ["phong", "darboux", "shadow"])
@pytest.mark.parametrize("waves", [4, 8, 16])
@pytest.mark.parametrize("mode", [1, 2])
def test_group_tile_layouts(small_synthetic, pipe, waves, mode):
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W, Hh, n = 400, 200, 5
    p = params(n)
    gpu = T.Scene(W, Hh, mesh, texs, pipe, tile_waves=waves, tile_mode=mode, frames_per_launch=4)
    gpu.render_frames(p)
    check_kept(gpu, oracle_frames(W, Hh, mesh, texs, pipe, p), pipe, n)
    gpu.close()


@pytest.mark.parametrize("pipe", ["phong", "shadow"])
def test_groups_between_per_frame_renders(small_synthetic, pipe):
    """Per-frame renders before a group call, an accumulating render and a cleared one after it, a second
    group call after those: the per-frame path and the groups share streams, bins' capacity and targets."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh = 384, 256
    gpu = T.Scene(W, Hh, mesh, texs, pipe, frames_per_launch=4)
    cpu = O.Scene(W, Hh, mesh, texs, pipe)

    def both(fn):
        fn(gpu), fn(cpu)

    def same():
        assert np.array_equal(gpu.read_z_f32().view(np.uint32), cpu.z_f32().view(np.uint32))
        assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer())

    for f in range(3):  # frames in flight on the per-frame path when the group call arrives
        both(lambda s: (s.clear(), s.set_light_direction(H.light(0.1 * f)), s.set_camera(*H.camera(0.5 * f)), s.render()))
    p = params(6, cam0=1.0)
    gpu.render_frames(p)
    expect = oracle_frames(W, Hh, mesh, texs, pipe, p)
    check_kept(gpu, expect, pipe, 6)
    # the oracle scene follows: its state after the call is the last frame's
    cpu.clear(), cpu.set_light_direction(p[-1, 0:3]), cpu.set_camera(p[-1, 3:6], p[-1, 6:9], p[-1, 9:12]), cpu.render()
    same()
    # accumulate into the last frame (no clear): light and camera are the last frame's unless set again
    both(lambda s: (s.set_camera(*H.camera(2.2)), s.render()))
    same()
    both(lambda s: (s.clear(), s.set_light_direction(H.light(0.9)), s.render()))
    same()
    # an older frame of the call, selected and accumulated into
    p2 = params(4, cam0=-1.0, cam_step=0.5)
    gpu.render_frames(p2)
    gpu.select_frame(2)
    cpu.clear(), cpu.set_light_direction(p2[1, 0:3]), cpu.set_camera(p2[1, 3:6], p2[1, 6:9], p2[1, 9:12]), cpu.render()
    same()
    both(lambda s: (s.set_camera(*H.camera(0.3)), s.render()))
    same()
    assert gpu.frames_kept() == 0  # a per-frame render ends the call's selection
    with pytest.raises(T.TinyRendererError):
        gpu.select_frame(0)
    gpu.close()


@pytest.mark.parametrize("pipe", ["phong", "shadow"])
def test_group_bin_overflow_renders_the_kept_frames_again(synthetic, pipe):
    """Bins too small for the frames of a group: found at the first getter, the bins grow and the frames
    the call left behind are rendered again -- every one of them, not only the last."""
    import tiny_renderer_amd as T
    mesh, texs = synthetic  # 5 022 polygons on a 256x256 frame: hundreds per tile
    W, Hh, n = 256, 256, 6
    p = params(n)
    gpu = T.Scene(W, Hh, mesh, texs, pipe, bin_capacity=64, frames_per_launch=4)
    gpu.render_frames(p)
    expect = oracle_frames(W, Hh, mesh, texs, pipe, p)
    check_kept(gpu, expect, pipe, n)
    # the grown bins serve the next call directly
    p2 = params(3, cam0=2.0)
    gpu.render_frames(p2)
    assert gpu.sync() == 0
    check_kept(gpu, oracle_frames(W, Hh, mesh, texs, pipe, p2), pipe, 3)
    gpu.close()


@pytest.mark.parametrize("pipe", ["phong", "shadow", "occlusion"])
def test_group_on_a_callers_stream_into_callers_buffers(small_synthetic, pipe):
    """The multi-GPU pattern: a caller's stream, the caller's frame tensors as colour targets, the frames
    consumed on that stream without host synchronisation; a band scene writes its rows only."""
    import torch
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W, Hh, n = 768, 512, 10
    band = (128, 384)
    side = torch.cuda.Stream()
    bufs = [torch.full((Hh * W * 3,), 77, dtype=torch.uint8, device="cuda") for _ in range(4)]
    kept = torch.zeros(n, Hh * W * 3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu = T.Scene(W, Hh, mesh, texs, pipe, stream=side.cuda_stream, frame_buffer_device=bufs[0].data_ptr(),
                  band_rows=band, frames_per_launch=4)
    p = params(n)
    with torch.cuda.stream(side):
        for i0 in range(0, n, 4):
            g = min(4, n - i0)
            gpu.render_frames(p[i0:i0 + g], [bufs[j].data_ptr() for j in range(g)])
            for j in range(g):
                kept[i0 + j].copy_(bufs[j], non_blocking=True)
    assert gpu.sync() == 0
    torch.cuda.synchronize()
    expect = oracle_frames(W, Hh, mesh, texs, pipe, p, band=band)
    for i in range(n):
        got = kept[i].cpu().numpy().reshape(Hh, W, 3)
        assert np.array_equal(got[band[0]:band[1]], expect[i][0][band[0]:band[1]]), "frame %d" % i
        assert (got[:band[0]] == 77).all() and (got[band[1]:] == 77).all(), "rows outside the band were written"
    gpu.close()


def test_group_with_winner_tap_and_profile(small_synthetic):
    """The winner tap is one buffer: a call then goes frame by frame (same results, same slots).  The profile
    counts the frames a launch covered."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh, n = 300, 200, 6
    p = params(n)
    gpu = T.Scene(W, Hh, mesh, texs, "phong", winner_tap=True, frames_per_launch=4)
    assert gpu.frames_per_launch == 1
    gpu.render_frames(p)
    cpu = O.Scene(W, Hh, mesh, texs, "phong")
    cpu.clear(), cpu.set_light_direction(p[-1, 0:3]), cpu.set_camera(p[-1, 3:6], p[-1, 6:9], p[-1, 9:12]), cpu.render()
    assert np.array_equal(gpu.read_winner_u32(), cpu.winner_u32())
    assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer())
    gpu.close()

    gpu = T.Scene(W, Hh, mesh, texs, "shadow", frames_per_launch=4)
    gpu.render_frames(p[:2])
    gpu.sync()
    gpu.profile_enable(True)
    gpu.render_frames(p)          # groups of 4 + 2
    prof = gpu.profile_read()
    assert prof["k_tile"]["launches"] == 2 and prof["k_tile"]["frames"] == 6
    assert prof["k_tile_depth"]["launches"] == 2 and prof["k_tile_depth"]["frames"] == 6
    assert prof["k_setup"]["launches"] == 4 and prof["k_setup"]["frames"] == 12
    assert len(gpu.profile_frame_intervals()) >= 2
    gpu.profile_enable(False)
    check_kept(gpu, oracle_frames(W, Hh, mesh, texs, "shadow", p), "shadow", n)
    gpu.close()


def test_full_size_group_equals_per_frame_path(diablo):
    """4096^2 (BASELINE configs[2] geometry), automatic group size: every kept frame of a group call equals the
    per-frame path's frame for the same light and camera (itself checked against the oracle at 800^2 /
    2048^2 and by properties at this size)."""
    import tiny_renderer_amd as T
    mesh, texs = diablo
    n = 6
    p = params(n, cam_step=0.11)
    gpu = T.Scene(4096, 4096, mesh, texs, "phong")
    assert gpu.frames_per_launch == 4
    gpu.render_frames(p)
    one = T.Scene(4096, 4096, mesh, texs, "phong")
    for back in range(gpu.frames_kept()):
        q = p[n - 1 - back]
        one.clear(), one.set_light_direction(q[0:3]), one.set_camera(q[3:6], q[6:9], q[9:12]), one.render()
        gpu.select_frame(back)
        assert np.array_equal(gpu.read_z_f32().view(np.uint32), one.read_z_f32().view(np.uint32))
        assert np.array_equal(gpu.get_frame_buffer(), one.get_frame_buffer())
    gpu.close()
    one.close()


def test_tiles_that_empty_after_an_orbit_are_cleared(diablo):
    """An orbit leaves colour in tiles that are empty in the next view; their workgroups must store the
    cleared colour although the tile's "already clean" flag is consulted by four waves that do not run in
    lockstep (a wave that raised the flag before its siblings had read it left 4-228 stale pixels per frame
    at this size).  4096^2 darboux: the frames after the orbit equal a fresh scene's frame."""
    import tiny_renderer_amd as T
    mesh, texs = diablo
    W = Hh = 4096
    fresh = T.Scene(W, Hh, mesh, texs, "darboux")
    head = params(8, cam_step=0.0, light_step=0.0)
    fresh.clear(), fresh.set_light_direction(head[0, 0:3]), fresh.set_camera(head[0, 3:6], head[0, 6:9], head[0, 9:12])
    fresh.render()
    z1, f1 = fresh.read_z_f32().view(np.uint32), fresh.get_frame_buffer()
    fresh.close()
    gpu = T.Scene(W, Hh, mesh, texs, "darboux")
    for lap in range(2):
        gpu.render_frames(params(200, cam_step=2.0 * np.pi / 200, light_step=0.0))
        gpu.render_frames(head)
        assert gpu.sync() == 0
        for back in range(gpu.frames_kept()):
            gpu.select_frame(back)
            assert np.array_equal(gpu.read_z_f32().view(np.uint32), z1)
            f = gpu.get_frame_buffer()
            assert np.array_equal(f, f1), "lap %d frame -%d: %d stale pixels" % (lap, back, int((f != f1).any(-1).sum()))
        # the per-frame path after the same orbit (its empty tiles take the same branch)
        for i in range(0, 200, 7):
            q = params(1, cam0=2.0 * np.pi * i / 200, light_step=0.0)[0]
            gpu.clear(), gpu.set_light_direction(q[0:3]), gpu.set_camera(q[3:6], q[6:9], q[9:12]), gpu.render()
        gpu.clear(), gpu.set_light_direction(head[0, 0:3]), gpu.set_camera(head[0, 3:6], head[0, 6:9], head[0, 9:12])
        gpu.render()
        assert np.array_equal(gpu.get_frame_buffer(), f1)
    gpu.close()


def test_render_frames_arguments_and_a_singular_camera(small_synthetic):
    """Zero frames are a no-op; a null frame list is refused; a frame the reference would panic on (occlusion
    with the light along -z: Rotation3::rotation_between(..).unwrap(), shader.rs:921) inside a call returns
    TR_E_SINGULAR, leaves nothing selectable, and the scene renders normally afterwards."""
    import ctypes as C
    import tiny_renderer_amd as T
    from tiny_renderer_amd import _lib
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh = 320, 200
    PIPE = "occlusion"
    gpu = T.Scene(W, Hh, mesh, texs, PIPE, frames_per_launch=4)
    L = T.load_library()
    assert L.tr_scene_render_frames(gpu._h, 0, None, None) == 0
    assert L.tr_scene_render_frames(gpu._h, 3, None, None) == _lib.TR_E_INVALID
    p = params(9)
    fbs = (C.c_void_p * 9)()          # a list with null entries
    assert L.tr_scene_render_frames(gpu._h, 9, p.ctypes.data, fbs) == _lib.TR_E_INVALID
    gpu.render_frames(p)
    assert gpu.frames_kept() == 4
    bad = p.copy()
    bad[6, 0:3] = (0.0, 0.0, -1.0)     # frame 6 (second group): the sample rotation does not exist
    ref = O.Scene(W, Hh, mesh, texs, PIPE)
    ref.clear(), ref.set_light_direction(bad[6, 0:3]), ref.set_camera(bad[6, 3:6], bad[6, 6:9], bad[6, 9:12])
    assert ref.render() != 0           # the oracle reports the reference's panic too
    with pytest.raises(T.TinyRendererError) as e:
        gpu.render_frames(bad)
    assert e.value.code == _lib.TR_E_SINGULAR
    assert gpu.frames_kept() == 0
    with pytest.raises(T.TinyRendererError):
        gpu.sync()                     # the failure is the frame's status, like a failed render()
    gpu.render_frames(p[:5])
    assert gpu.sync() == 0
    check_kept(gpu, oracle_frames(W, Hh, mesh, texs, PIPE, p[:5]), PIPE, 5)
    gpu.clear(), gpu.set_light_direction(p[2, 0:3]), gpu.set_camera(p[2, 3:6], p[2, 6:9], p[2, 9:12]), gpu.render()
    cpu = O.Scene(W, Hh, mesh, texs, PIPE)
    cpu.clear(), cpu.set_light_direction(p[2, 0:3]), cpu.set_camera(p[2, 3:6], p[2, 6:9], p[2, 9:12]), cpu.render()
    assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer())
    gpu.close()


def test_group_then_callers_buffer_for_single_frames(small_synthetic):
    """After a group call into the scene's own frame slots the caller swaps in its own colour buffer
    (tr_scene_set_frame_buffer_device) and renders single frames into it: the slot's z, the caller's colour."""
    import torch
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W, Hh = 384, 208
    p = params(6)
    gpu = T.Scene(W, Hh, mesh, texs, "shadow", frames_per_launch=4)
    gpu.render_frames(p)
    buf = torch.full((Hh * W * 3,), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu.set_frame_buffer_device(buf.data_ptr())
    q = p[3]
    gpu.clear(), gpu.set_light_direction(q[0:3]), gpu.set_camera(q[3:6], q[6:9], q[9:12]), gpu.render()
    assert gpu.sync() == 0
    torch.cuda.synchronize()
    want = oracle_frames(W, Hh, mesh, texs, "shadow", p[3:4])[0]
    assert np.array_equal(buf.cpu().numpy().reshape(Hh, W, 3), want[0])
    assert np.array_equal(gpu.read_z_f32().view(np.uint32), want[1])
    assert np.array_equal(gpu.read_shadow_f32().view(np.uint32), want[2])
    gpu.set_frame_buffer_device(None)
    gpu.clear(), gpu.render()
    assert np.array_equal(gpu.get_frame_buffer(), want[0])
    gpu.close()


# ---- automatic frame groups: the per-frame calls, held back and fused by the library ------------------------

def _per_frame(s, q, clear=True):
    if clear:
        s.clear()
    s.set_light_direction(q[0:3])
    s.set_camera(q[3:6], q[6:9], q[9:12])
    s.render()


@pytest.mark.parametrize("pipe", ["phong", "shadow", "darboux"])
def test_auto_groups_read_after_k_frames(small_synthetic, pipe):
    """clear -> set_* -> render, k times without a read, then a getter: the frame returned is the oracle's k-th
    frame whatever k is (one frame alone takes the ordinary path, 2..G a partial group, more than G full groups
    plus a remainder), z and shadow buffer too; an accumulating render and a pending clear survive the hold-back."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh = 352, 224
    p = params(48, cam_step=0.29)
    gpu = T.Scene(W, Hh, mesh, texs, pipe, frames_per_launch=4)
    cpu = O.Scene(W, Hh, mesh, texs, pipe)
    i = 0
    for k in (1, 2, 3, 4, 5, 7, 9, 1, 6):
        for _ in range(k):
            _per_frame(gpu, p[i]); _per_frame(cpu, p[i])
            i += 1
        assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer()), "after %d frames" % k
        assert np.array_equal(gpu.read_z_f32().view(np.uint32), cpu.z_f32().view(np.uint32))
        if pipe == "shadow":
            assert np.array_equal(gpu.read_shadow_f32().view(np.uint32), cpu.shadow_f32().view(np.uint32))
    # held-back frames, then a render WITHOUT clear (accumulates onto the last of them), then a clear that stays pending
    for _ in range(2):
        _per_frame(gpu, p[i]); _per_frame(cpu, p[i]); i += 1
    _per_frame(gpu, p[i], clear=False); _per_frame(cpu, p[i], clear=False); i += 1
    _per_frame(gpu, p[i]); _per_frame(cpu, p[i]); i += 1
    gpu.clear(); cpu.clear()
    assert not gpu.get_frame_buffer().any() and not cpu.get_frame_buffer().any()   # the clear, materialised
    _per_frame(gpu, p[i], clear=False); _per_frame(cpu, p[i], clear=False)
    assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer())
    gpu.close()


def test_auto_groups_equal_the_unfused_path_and_can_be_switched_off(synthetic):
    import tiny_renderer_amd as T
    mesh, texs = synthetic
    W, Hh, n = 640, 400, 11
    p = params(n)
    frames = []
    for auto in (True, False):
        s = T.Scene(W, Hh, mesh, texs, "normal_map", auto_group=auto)
        s.profile_enable(True)
        for q in p:
            _per_frame(s, q)
        prof = s.profile_read()
        s.profile_enable(False)
        assert prof["k_tile"]["frames"] == n
        if auto:
            assert prof["k_tile"]["launches"] < n       # fused launches ...
        else:
            assert prof["k_tile"]["launches"] == n      # ... or one per frame
        frames.append((s.get_frame_buffer(), s.read_z_f32().view(np.uint32).copy()))
        s.close()
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1])
    want = oracle_frames(W, Hh, mesh, texs, "normal_map", p[-1:])[0]
    assert np.array_equal(frames[0][0], want[0]) and np.array_equal(frames[0][1], want[1])


def test_auto_groups_with_alternating_callers_buffers_and_async_reads(small_synthetic):
    """A caller on the library's own stream that alternates two colour buffers (tr_scene_set_frame_buffer_device)
    finds the last two frames in them after a sync although the frames were fused; asynchronous read-backs queued
    between the renders each get their own frame."""
    import torch
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W, Hh, n = 320, 192, 9
    p = params(n)
    want = oracle_frames(W, Hh, mesh, texs, "phong", p)
    bufs = [torch.full((Hh * W * 3,), 5, dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    gpu = T.Scene(W, Hh, mesh, texs, "phong", frames_per_launch=4)
    for i in range(n):
        gpu.set_frame_buffer_device(bufs[i % 2].data_ptr())
        _per_frame(gpu, p[i])
    assert gpu.sync() == 0
    torch.cuda.synchronize()
    for i in (n - 2, n - 1):
        assert np.array_equal(bufs[i % 2].cpu().numpy().reshape(Hh, W, 3), want[i][0]), "frame %d" % i
    gpu.set_frame_buffer_device(None)
    pinned = [gpu.pinned_frame() for _ in range(n)]
    for i in range(n):
        _per_frame(gpu, p[i])
        if i % 3 != 1:
            gpu.get_frame_buffer_async(pinned[i])
    assert gpu.sync() == 0
    for i in range(n):
        if i % 3 != 1:
            assert np.array_equal(pinned[i], want[i][0]), "read-back of frame %d" % i
    gpu.close()


def test_auto_groups_bin_overflow_and_teardown(synthetic):
    """Bins too small for frames that were fused behind the caller's back: the getter still returns the right
    frame (the last render is replayed); a scene destroyed with frames held back renders them on the way out."""
    import tiny_renderer_amd as T
    mesh, texs = synthetic
    W, Hh = 256, 256
    p = params(7)
    gpu = T.Scene(W, Hh, mesh, texs, "phong", bin_capacity=64, frames_per_launch=4)
    for q in p:
        _per_frame(gpu, q)
    want = oracle_frames(W, Hh, mesh, texs, "phong", p[-1:])[0]
    assert np.array_equal(gpu.read_z_f32().view(np.uint32), want[1])
    assert np.array_equal(gpu.get_frame_buffer(), want[0])
    for q in p[:3]:
        _per_frame(gpu, q)
    gpu.close()   # three frames still on the host


@pytest.mark.parametrize("pipe", ["phong", "shadow", "specular"])
def test_transient_depth_is_made_real_on_demand(small_synthetic, pipe):
    """The fused launches leave a cleared frame's depth on the chip (TileArgs::store); whoever wants the z buffer gets it
    from a repeat of the colour pass for the depth alone: the z getters (bits and u8 view), a render WITHOUT a clear that
    depth-tests against a kept frame (scene.rs:151), each kept frame in turn -- always the oracle's z, the colour
    untouched, one repeat per frame (not per read) -- and TR_OPT_STORE_DEPTH switches the whole thing off."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    mesh, texs = small_synthetic
    W, Hh, n = 384, 256, 7
    p = params(n)
    expect = oracle_frames(W, Hh, mesh, texs, pipe, p)
    exact = pipe in EXACT or specular_exact()
    for store_depth in (False, True):
        gpu = T.Scene(W, Hh, mesh, texs, pipe, frames_per_launch=4, store_depth=store_depth)
        gpu.render_frames(p)
        gpu.sync()
        gpu.profile_enable(True)
        for back in (2, 0, 1):                                    # kept frames, in no particular order
            gpu.select_frame(back)
            fo, zo, _ = expect[n - 1 - back]
            assert np.array_equal(gpu.read_z_f32().view(np.uint32), zo), (pipe, store_depth, back)
            assert np.array_equal(gpu.read_z_f32().view(np.uint32), zo)          # ... and again: already real
            if exact:
                assert np.array_equal(gpu.get_frame_buffer(), fo), "the depth-only repeat touched the colour"
        launches = gpu.profile_read().get("k_tile", {"launches": 0})["launches"]
        assert launches == (0 if store_depth else 3), (store_depth, launches)   # one repeat per frame, none when stored
        gpu.profile_enable(False)
        # the u8 view of the z buffer, straight after a group (nothing made real yet for the newest frame of a new call)
        gpu.render_frames(p[:5])
        cpu = O.Scene(W, Hh, mesh, texs, pipe)
        q = p[4]
        cpu.clear(), cpu.set_light_direction(q[0:3]), cpu.set_camera(q[3:6], q[6:9], q[9:12]), cpu.render()
        assert np.array_equal(gpu.get_z_buffer(), cpu.get_z_buffer())
        # an accumulating render on top of a kept frame whose depth was never stored
        gpu.render_frames(p[:6])
        gpu.select_frame(1)                                        # frame 4 again
        for s in (gpu, cpu):
            s.set_light_direction(H.light(0.9)), s.set_camera(*H.camera(2.2)), s.render()   # no clear
        assert np.array_equal(gpu.read_z_f32().view(np.uint32), cpu.z_f32().view(np.uint32)), (pipe, store_depth, "accumulate")
        if exact:
            assert np.array_equal(gpu.get_frame_buffer(), cpu.get_frame_buffer())
        # a clear after a deferred frame: the z buffer is the cleared value, nothing is repeated
        gpu.render_frames(p[:3])
        gpu.clear(), cpu.clear()
        assert np.array_equal(gpu.read_z_f32().view(np.uint32), cpu.z_f32().view(np.uint32))
        cpu.close()
        gpu.close()


@pytest.mark.parametrize("pipe", ["phong", "darboux"])   # (shadow's lookups leave the buffer under these cameras: upstream panics)
def test_work_units_leave_no_tile_behind(small_synthetic, pipe):
    """The tile kernel's grid is one workgroup per tile WITH polygons and one per 32 empty tiles, sized by the list
    lengths of the frame of the group that needs most (k_bin_group reports them; the host launches behind a chain it has
    seen complete) -- and one per tile when the lengths are not known (a call's first group).  Frames whose models sit
    in different corners of the screen, so that every group holds frames with very different numbers of busy tiles and
    most tiles flip between busy and empty from one frame of a slot to the next (an empty unit must clear what the
    slot's previous frame left there): every frame the call leaves behind is the oracle's, in a call long enough for
    exact grids (five groups) and again after per-frame renders."""
    import tiny_renderer_amd as T
    mesh, texs = small_synthetic
    W = Hh = 1536   # 12 x 96 tiles: 36 chunks of empty tiles
    n = 19
    p = params(n, cam_step=0.31)
    corners = [(0.0, 0.0), (0.9, 0.7), (-0.9, -0.7), (0.9, -0.7), (-1.6, 0.0), (0.0, 1.3)]
    for i in range(n):
        cx, cy = corners[i % len(corners)]
        p[i, 6:9] = (cx, cy, 0.0)          # look_at: the model leaves the middle of the frame (partly the frame itself)
    gpu = T.Scene(W, Hh, mesh, texs, pipe, frames_per_launch=4)
    expect = oracle_frames(W, Hh, mesh, texs, pipe, p)
    for rep in range(2):
        gpu.render_frames(p)
        assert gpu.sync() == 0
        check_kept(gpu, expect, pipe, n)
        q = p[3]
        gpu.clear(), gpu.set_light_direction(q[0:3]), gpu.set_camera(q[3:6], q[6:9], q[9:12]), gpu.render()
        assert np.array_equal(gpu.get_frame_buffer(), expect[3][0])
    gpu.close()
