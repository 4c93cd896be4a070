"""ctypes binding of the TEST-ONLY host emulation (tests/emul/emul.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "emul", "libtr_emul.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "emul")])
        L = C.CDLL(_LIB)
        L.tr_emul_render.restype = C.c_uint32
        L.tr_emul_covers.restype = C.c_int
        L.tr_emul_covers.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_float)]
        L.tr_emul_div.restype = None
        L.tr_emul_div.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.tr_emul_decode_normal_mismatches.restype = C.c_uint32
        L.tr_emul_decode_normal_mismatches.argtypes = [C.c_uint32, C.c_uint32]
        L.tr_emul_shadow_fetch_mismatches.restype = C.c_uint32
        L.tr_emul_shadow_fetch_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                      C.c_void_p, C.c_void_p, C.c_uint32]
        L.tr_emul_pair_counts.restype = None
        L.tr_emul_pair_counts.argtypes = [C.POINTER(C.c_uint64)]
        L.tr_emul_mask_counts.restype = None
        L.tr_emul_mask_counts.argtypes = [C.POINTER(C.c_uint64)]
        L.tr_emul_blend_mismatches.restype = C.c_uint64
        L.tr_emul_blend_mismatches.argtypes = [C.c_void_p, C.c_uint64]
        L.tr_emul_blend.restype = C.c_uint32
        L.tr_emul_blend.argtypes = [C.c_uint32, C.c_float, C.c_int]
        L.tr_emul_powf.restype = C.c_int
        L.tr_emul_powf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _lib = L
    return _lib


def mask_counts():
    """{covered pixels found outside pair_masks' cells (must stay 0), live cells, box cells} so far."""
    out = (C.c_uint64 * 3)()
    lib().tr_emul_mask_counts(out)
    return int(out[0]), int(out[1]), int(out[2])


def render(W, Hh, mesh, texs, pipe, light, cam, fresh=1, bufs=None, band=(0, 0)):
    """Runs the product's stage functions through the kernels' decomposition on the CPU.
    Returns (err, z, shadow, fb, winner)."""
    from tiny_renderer_amd import _lib as TL
    from tiny_renderer_amd.scene import _mesh_struct
    keep = []
    m = _mesh_struct(mesh, keep)
    imgs = (TL.ImageRgb8 * 4)()
    for k, t in enumerate(texs):
        t = np.ascontiguousarray(t, np.uint8)
        keep.append(t)
        imgs[k] = TL.ImageRgb8(t.ctypes.data_as(C.POINTER(C.c_uint8)), t.shape[1], t.shape[0])
    if bufs is None:
        z = np.zeros((Hh, W), np.float32)
        sh = np.zeros((Hh, W), np.float32)
        fb = np.zeros((Hh, W, 3), np.uint8)
        win = np.full((Hh, W), 0xFFFFFFFF, np.uint32)
    else:
        z, sh, fb, win = bufs

    def f3(v):
        return (C.c_float * 3)(*[float(x) for x in v])

    err = lib().tr_emul_render(C.c_uint32(W), C.c_uint32(Hh), C.byref(m), imgs, pipe.encode(), f3(light),
                               f3(cam[0]), f3(cam[1]), f3(cam[2]), C.c_int(fresh),
                               z.ctypes.data_as(C.c_void_p), sh.ctypes.data_as(C.c_void_p),
                               fb.ctypes.data_as(C.c_void_p), win.ctypes.data_as(C.c_void_p),
                               C.c_uint32(band[0]), C.c_uint32(band[1]))
    return err, z, sh, fb, win
