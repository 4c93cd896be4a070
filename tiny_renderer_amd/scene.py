"""Python mirror of the reference's `Scene` (src/scene.rs:25-269) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, load_library

# shader.rs:100-109
PIPELINES = ("default", "phong", "normal_map", "specular", "darboux", "shadow", "occlusion")


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _mesh_struct(mesh, keep):
    pos = np.ascontiguousarray(mesh["pos"], np.float32).reshape(-1, 3)
    tex = np.ascontiguousarray(mesh["tex"], np.float32).reshape(-1, 3)
    nrm = np.ascontiguousarray(mesh["nrm"], np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(mesh["idx"], np.uint32).reshape(-1, 9)
    keep += [pos, tex, nrm, idx]
    fp = C.POINTER(C.c_float)
    return _lib.Mesh(pos.ctypes.data_as(fp), tex.ctypes.data_as(fp), nrm.ctypes.data_as(fp),
                     idx.ctypes.data_as(C.POINTER(C.c_uint32)), pos.shape[0], tex.shape[0],
                     nrm.shape[0], idx.shape[0])


class Scene:
    """Scene::new(width, height, obj, texture, normal_map, normal_map_tangent, specular_map,
    shader_pipeline_name) -- scene.rs:47-56.

    mesh: dict of float32 pos[n,3], tex[n,3], nrm[n,3] and uint32 idx[n_tri,9]
          (p0,t0,n0,p1,t1,n1,p2,t2,n2 -- obj::raw::RawObj as the path reads it).
    textures: uint8 [h,w,3] arrays in Scene::new's order.
    Extra keyword options are the tr_options of include/tiny_renderer.h.
    """

    def __init__(self, width, height, mesh, textures, shader_pipeline_name, *, device=-1,
                 winner_tap=False, tile_stamps=False, band_rows=None, stream=None, frame_buffer_device=None,
                 bin_capacity=0, tile_waves=0, tile_mode=0, frames_per_launch=0, auto_group=True,
                 trust_frame_buffers=False, max_frame_slots=0, store_depth=False):
        L = load_library()
        self.width, self.height = int(width), int(height)
        keep = []
        m = _mesh_struct(mesh, keep)
        imgs = (_lib.ImageRgb8 * 4)()
        if len(textures) != 4:
            raise ValueError("four textures are required (texture, normal_map, normal_map_tangent, specular_map)")
        for k, t in enumerate(textures):
            t = np.ascontiguousarray(t, np.uint8)
            keep.append(t)
            imgs[k] = _lib.ImageRgb8(t.ctypes.data_as(C.POINTER(C.c_uint8)), t.shape[1], t.shape[0])
        o = _lib.Options()
        o.struct_size = C.sizeof(_lib.Options)
        o.device = device
        o.flags = ((_lib.TR_OPT_WINNER_TAP if winner_tap else 0) | (_lib.TR_OPT_TILE_STAMPS if tile_stamps else 0)
                   | (0 if auto_group else _lib.TR_OPT_NO_AUTO_GROUP)
                   | (_lib.TR_OPT_TRUST_FRAME_BUFFERS if trust_frame_buffers else 0)
                   | (_lib.TR_OPT_STORE_DEPTH if store_depth else 0))
        if band_rows is not None:
            o.band_row0, o.band_row1 = int(band_rows[0]), int(band_rows[1])
        o.stream = stream
        o.frame_buffer_device = frame_buffer_device
        o.bin_capacity = int(bin_capacity)
        o.tile_waves = int(tile_waves)
        o.tile_mode = int(tile_mode)
        o.frames_per_launch = int(frames_per_launch)
        o.max_frame_slots = int(max_frame_slots)
        h = C.c_void_p()
        self._h = None
        self._pinned = []
        check(L.tr_scene_create(self.width, self.height, C.byref(m), imgs,
                                shader_pipeline_name.encode(), C.byref(o), C.byref(h)))
        self._h = h
        self.pipeline = shader_pipeline_name

    def close(self):
        if getattr(self, "_h", None):
            load_library().tr_scene_destroy(self._h)   # waits for queued work, pending read-backs included
            self._h = None
            for p in getattr(self, "_pinned", []):
                load_library().tr_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- the reference's methods -------------------------------------------------------------
    def clear(self):
        check(load_library().tr_scene_clear(self._h))

    def set_light_direction(self, light_direction):
        check(load_library().tr_scene_set_light_direction(self._h, _f3(light_direction)))

    def set_camera(self, look_from, look_at, up):
        check(load_library().tr_scene_set_camera(self._h, _f3(look_from), _f3(look_at), _f3(up)))

    def render(self):
        check(load_library().tr_scene_render(self._h))

    def set_auto_group(self, on):
        """Whether render() may hold cleared frames back to fuse them (default on; tr_scene_set_auto_group)."""
        check(load_library().tr_scene_set_auto_group(self._h, 1 if on else 0))

    def render_frames(self, frames, frame_buffers_device=None):
        """tr_scene_render_frames: `frames` is an [n, 12] float32 array (or a list of (light, look_from,
        look_at, up) tuples): per frame light direction, look_from, look_at, up.  Frame i is what
        clear(); set_light_direction; set_camera; render() produces; the frames of a group are rendered by
        one launch per kernel.  frame_buffers_device: optional list of n device pointers (colour targets)."""
        if not isinstance(frames, np.ndarray):
            frames = np.asarray([np.concatenate([np.asarray(v, np.float32).reshape(3) for v in f]) for f in frames],
                                np.float32)
        frames = np.ascontiguousarray(frames, np.float32).reshape(-1, 12)
        fbs = None
        if frame_buffers_device is not None:
            if len(frame_buffers_device) != len(frames):
                raise ValueError("one frame buffer per frame")
            fbs = (C.c_void_p * len(frames))(*[int(q) for q in frame_buffers_device])
        check(load_library().tr_scene_render_frames(self._h, len(frames), frames.ctypes.data, fbs))

    @property
    def frames_per_launch(self):
        return check(load_library().tr_scene_frames_per_launch(self._h))

    def frames_kept(self):
        return check(load_library().tr_scene_frames_kept(self._h))

    def select_frame(self, back):
        """Makes the frame `back` frames before the last one of the last render_frames call current."""
        check(load_library().tr_scene_select_frame(self._h, int(back)))

    def _image(self, fn, strict):
        out = np.empty((self.height, self.width, 3), np.uint8)
        code = getattr(load_library(), fn)(self._h, out.ctypes.data)
        if strict:
            check(code)
        self.last_status = code
        return out

    def get_frame_buffer(self, strict=True):
        return self._image("tr_scene_get_frame_buffer", strict)

    def pinned_frame(self):
        """A page-locked [H, W, 3] uint8 array for get_frame_buffer_async (freed with the scene)."""
        L = load_library()
        n = self.width * self.height * 3
        p = L.tr_host_alloc(n)
        if not p:
            raise MemoryError("tr_host_alloc(%d)" % n)
        self._pinned.append(p)
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(p)).reshape(self.height, self.width, 3)

    def get_frame_buffer_async(self, out):
        """Enqueue the read-back of the frame into `out` (see pinned_frame); valid after sync()."""
        if out.nbytes != self.width * self.height * 3 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a contiguous [H, W, 3] uint8 array")
        check(load_library().tr_scene_get_frame_buffer_async(self._h, out.ctypes.data))
        return out

    def host_buffer_written(self, out):
        """Tells the scene that the caller has written into a pinned_frame() array (it then assumes nothing about
        the array's content at the next get_frame_buffer_async)."""
        check(load_library().tr_scene_host_buffer_written(self._h, out.ctypes.data))

    def get_z_buffer(self, strict=True):
        return self._image("tr_scene_get_z_buffer", strict)

    def get_shadow_buffer(self, strict=True):
        return self._image("tr_scene_get_shadow_buffer", strict)

    # --- parity taps / device-resident access -------------------------------------------------
    def _raw(self, fn, dtype):
        out = np.empty((self.height, self.width), dtype)
        check(getattr(load_library(), fn)(self._h, out.ctypes.data))
        return out

    def read_z_f32(self):
        return self._raw("tr_scene_read_z_f32", np.float32)

    def read_shadow_f32(self):
        return self._raw("tr_scene_read_shadow_f32", np.float32)

    def read_winner_u32(self):
        return self._raw("tr_scene_read_winner_u32", np.uint32)

    def sync(self):
        return check(load_library().tr_scene_sync(self._h))

    def flush(self):
        """Hand every render issued so far to the device without waiting (renders may be held back in batches)."""
        return check(load_library().tr_scene_flush(self._h))

    def frame_buffer_device(self):
        return load_library().tr_scene_frame_buffer_device(self._h)

    def set_frame_buffer_device(self, ptr):
        """Renders issued from now on write the device buffer at `ptr` (None: the library's own)."""
        check(load_library().tr_scene_set_frame_buffer_device(self._h, ptr))

    def set_stream(self, stream):
        check(load_library().tr_scene_set_stream(self._h, stream))

    def debug_tile_stamps(self):
        """[n_tiles, 8] uint64: start, end (100 MHz ticks), polygons in the bin, hardware id, bin staged, coverage done."""
        cap = ((self.width + 127) // 128) * ((self.height + 7) // 8)
        out = np.zeros((cap, 8), np.uint64)
        n = check(load_library().tr_scene_debug_tile_stamps(self._h, out.ctypes.data, cap))
        return out[:n]

    def band_tiles(self, frame_buffer_device=None):
        """tr_scene_band_tiles: the tiles of one of the frame buffers this scene has rendered into (None: the current
        one) with their cleared-colour flags -- what PeerExchange.all_gather_tiles sends by."""
        out = _lib.BandTiles()
        check(load_library().tr_scene_band_tiles(self._h, frame_buffer_device, C.byref(out)))
        return out

    def profile_enable(self, on=True):
        check(load_library().tr_scene_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        buf = (_lib.KernelTime * 16)()
        n = check(load_library().tr_scene_profile_read(self._h, buf, 16))
        return {buf[i].name.decode(): {"launches": int(buf[i].launches), "total_ms": float(buf[i].total_ms),
                                       "frames": int(buf[i].frames)}
                for i in range(n)}


def _frame_intervals(self, cap=1 << 16):
    """Microseconds between the completions of consecutive profiled frames (float32 array)."""
    out = np.zeros(cap, np.float32)
    n = check(load_library().tr_scene_profile_frame_intervals(self._h, out.ctypes.data, cap))
    return out[:n]


Scene.profile_frame_intervals = _frame_intervals


def band_rows(height, n_ranks, rank):
    """Output rows [row0, row1) rank `rank` of `n_ranks` renders (tr_band_rows, include/tiny_renderer.h)."""
    r0, r1 = C.c_uint32(), C.c_uint32()
    check(load_library().tr_band_rows(height, n_ranks, rank, C.byref(r0), C.byref(r1)))
    return int(r0.value), int(r1.value)


def prepare_uniforms(kind, width, height, light, look_from, look_at, up, uniforms=None):
    """shader.rs:183-279 on the host (no GPU needed).  Returns (status, Uniforms)."""
    u = uniforms if uniforms is not None else _lib.Uniforms()
    st = load_library().tr_prepare_uniforms(kind, C.byref(u), width, height, _f3(light),
                                            _f3(look_from), _f3(look_at), _f3(up))
    return st, u
