"""ctypes binding of the CPU oracle (oracle/tr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  PARITY UNPINNED (see tr_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtr_oracle.so")

E_UNKNOWN_PIPELINE = 1 << 0
E_W_ZERO = 1 << 1
E_SINGULAR = 1 << 2
E_TEX_OOB = 1 << 3
E_SHADOW_OOB = 1 << 4
E_ROTATION = 1 << 5
E_VEC_W_NONZERO = 1 << 6
E_INDEX_OOB = 1 << 7


class Mesh(C.Structure):
    _fields_ = [("pos", C.POINTER(C.c_float)), ("tex", C.POINTER(C.c_float)),
                ("nrm", C.POINTER(C.c_float)), ("idx", C.POINTER(C.c_uint32)),
                ("n_pos", C.c_uint32), ("n_tex", C.c_uint32), ("n_nrm", C.c_uint32),
                ("n_tri", C.c_uint32)]


class Image(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("w", C.c_uint32), ("h", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("tri_total", C.c_uint64), ("tri_kept", C.c_uint64), ("bbox_px", C.c_uint64),
                ("frag_covered", C.c_uint64), ("frag_accept", C.c_uint64),
                ("shadow_upd", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Uniforms(C.Structure):
    _fields_ = [("camera_direction", C.c_float * 3), ("t_light_direction", C.c_float * 3),
                ("vpmv", C.c_float * 16), ("i_vpmv", C.c_float * 16), ("m", C.c_float * 16),
                ("i_m", C.c_float * 16), ("it_m", C.c_float * 16),
                ("shadow_matrix", C.c_float * 16)]

    def as_dict(self):
        return {k: np.array(getattr(self, k), dtype=np.float32) for k, _ in self._fields_}


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "tr_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libtr_oracle.so"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None
_variant_path = None


def use_library(path=None):
    """scripts/oracle_variants.py only: bind the module to another build of tr_oracle.c (a TRO_VARIANT).
    Scenes created before the switch must not be used after it.  None = back to the normative oracle."""
    global _lib, _variant_path
    _lib = None
    _variant_path = path


def lib():
    global _lib
    if _lib is None:
        if _variant_path is None:
            build()
        L = C.CDLL(_variant_path or _LIB_PATH)
        L.tro_scene_new.restype = C.c_void_p
        L.tro_scene_new.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Mesh), C.POINTER(Image), C.c_char_p]
        L.tro_scene_free.argtypes = [C.c_void_p]
        L.tro_scene_clear.argtypes = [C.c_void_p]
        L.tro_scene_set_light_direction.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.tro_scene_set_camera.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 3
        L.tro_scene_render.argtypes = [C.c_void_p]
        L.tro_scene_render.restype = C.c_int
        L.tro_scene_set_output_band.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.tro_scene_set_output_band.restype = C.c_int
        for n in ("tro_scene_get_frame_buffer", "tro_scene_get_z_buffer", "tro_scene_get_shadow_buffer"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
        L.tro_scene_z_f32.restype = C.POINTER(C.c_float)
        L.tro_scene_z_f32.argtypes = [C.c_void_p]
        L.tro_scene_shadow_f32.restype = C.POINTER(C.c_float)
        L.tro_scene_shadow_f32.argtypes = [C.c_void_p]
        L.tro_scene_winner_u32.restype = C.POINTER(C.c_uint32)
        L.tro_scene_winner_u32.argtypes = [C.c_void_p]
        L.tro_scene_frame_raw.restype = C.POINTER(C.c_uint8)
        L.tro_scene_frame_raw.argtypes = [C.c_void_p]
        L.tro_scene_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.tro_scene_uniforms.argtypes = [C.c_void_p, C.POINTER(Uniforms)]
        L.tro_prepare.restype = C.c_int
        L.tro_prepare.argtypes = [C.c_int, C.POINTER(Uniforms), C.c_uint32, C.c_uint32] + \
            [C.POINTER(C.c_float)] * 4
        fp = C.POINTER(C.c_float)
        L.tro_mat4_mul.argtypes = [fp, fp, fp]
        L.tro_mat4_mul_vec4.argtypes = [fp, fp, fp]
        L.tro_mat4_inverse.argtypes = [fp, fp]
        L.tro_mat4_inverse.restype = C.c_int
        L.tro_mat3_inverse.argtypes = [fp, fp]
        L.tro_mat3_inverse.restype = C.c_int
        L.tro_barycentric.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, fp]
        L.tro_f32_to_i32.argtypes = [C.c_float]
        L.tro_f32_to_i32.restype = C.c_int32
        L.tro_f32_to_u32.argtypes = [C.c_float]
        L.tro_f32_to_u32.restype = C.c_uint32
        L.tro_f32_to_u8.argtypes = [C.c_float]
        L.tro_f32_to_u8.restype = C.c_uint8
        L.tro_color_blend.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_float,
                                      C.POINTER(C.c_uint8)]
        L.tro_occlusion_steps.argtypes = [fp, fp]
        L.tro_occlusion_steps.restype = C.c_int
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Scene:
    """Mirror of the reference's `Scene` (scene.rs:25-269) on the CPU oracle.

    mesh: dict with float32 arrays pos[n,3], tex[n,3], nrm[n,3] and uint32 idx[n_tri,9].
    textures: four uint8 arrays [h,w,3]: texture, normal_map, normal_map_tangent, specular_map.
    """

    def __init__(self, width, height, mesh, textures, pipeline):
        L = lib()
        self.width, self.height = int(width), int(height)
        self._keep = [np.ascontiguousarray(mesh["pos"], np.float32),
                      np.ascontiguousarray(mesh["tex"], np.float32),
                      np.ascontiguousarray(mesh["nrm"], np.float32),
                      np.ascontiguousarray(mesh["idx"], np.uint32)]
        pos, tex, nrm, idx = self._keep
        m = Mesh(_fptr(pos), _fptr(tex), _fptr(nrm), idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                 pos.shape[0], tex.shape[0], nrm.shape[0], idx.shape[0])
        imgs = (Image * 4)()
        for k, t in enumerate(textures):
            t = np.ascontiguousarray(t, np.uint8)
            self._keep.append(t)
            imgs[k] = Image(t.ctypes.data_as(C.POINTER(C.c_uint8)), t.shape[1], t.shape[0])
        self._h = L.tro_scene_new(self.width, self.height, C.byref(m), imgs, pipeline.encode())
        if not self._h:
            raise ValueError("Provided pipeline name is not supported!")  # shader.rs:108

    def close(self):
        if self._h:
            lib().tro_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        lib().tro_scene_clear(self._h)

    def set_light_direction(self, v):
        lib().tro_scene_set_light_direction(self._h, _f3(v))

    def set_camera(self, look_from, look_at, up):
        lib().tro_scene_set_camera(self._h, _f3(look_from), _f3(look_at), _f3(up))

    def set_output_band(self, row0, row1):
        """the colour pass's clamp rectangle cut to output rows [row0, row1) (row 0 = top)."""
        if lib().tro_scene_set_output_band(self._h, int(row0), int(row1)) != 0:
            raise ValueError("band rows must satisfy row0 < row1 <= height")

    def render(self):
        """Returns the TRO_E_* bitmask (0 = the reference would not have panicked)."""
        return lib().tro_scene_render(self._h)

    def _img(self, fn):
        out = np.empty((self.height, self.width, 3), np.uint8)
        getattr(lib(), fn)(self._h, out.ctypes.data)
        return out

    def get_frame_buffer(self):
        return self._img("tro_scene_get_frame_buffer")

    def get_z_buffer(self):
        return self._img("tro_scene_get_z_buffer")

    def get_shadow_buffer(self):
        return self._img("tro_scene_get_shadow_buffer")

    def _n(self):
        return self.width * self.height

    def z_f32(self):
        return np.ctypeslib.as_array(lib().tro_scene_z_f32(self._h), (self._n(),)).reshape(
            self.height, self.width).copy()

    def shadow_f32(self):
        return np.ctypeslib.as_array(lib().tro_scene_shadow_f32(self._h), (self._n(),)).reshape(
            self.height, self.width).copy()

    def winner_u32(self):
        return np.ctypeslib.as_array(lib().tro_scene_winner_u32(self._h), (self._n(),)).reshape(
            self.height, self.width).copy()

    def frame_raw(self):
        return np.ctypeslib.as_array(lib().tro_scene_frame_raw(self._h), (self._n() * 3,)).reshape(
            self.height, self.width, 3).copy()

    def stats(self):
        st = (Stats * 2)()
        lib().tro_scene_stats(self._h, st)
        return [st[0].as_dict(), st[1].as_dict()]

    def uniforms(self):
        u = Uniforms()
        lib().tro_scene_uniforms(self._h, C.byref(u))
        return u.as_dict()


def prepare(kind, width, height, light, look_from, look_at, up, uniforms=None):
    u = uniforms if uniforms is not None else Uniforms()
    err = lib().tro_prepare(kind, C.byref(u), width, height, _f3(light), _f3(look_from),
                            _f3(look_at), _f3(up))
    return err, u
