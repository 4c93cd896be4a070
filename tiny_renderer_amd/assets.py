"""Asset loading through the C ABI (app.rs:87-131: model.obj + four TGA files)."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check, load_library

# app.rs:87-91
ASSET_FILES = ("model.obj", "texture.tga", "normal_map.tga", "normal_map_tangent.tga", "specular_map.tga")


def load_obj(path):
    L = load_library()
    mp = C.POINTER(_lib.Mesh)()
    check(L.tr_load_obj(os.fsencode(path), C.byref(mp)))
    try:
        m = mp.contents

        def arr(ptr, n, k, dt):
            if n == 0:
                return np.zeros((0, k), dt)
            return np.ctypeslib.as_array(ptr, (n, k)).astype(dt, copy=True)

        return {"pos": arr(m.pos, m.n_pos, 3, np.float32), "tex": arr(m.tex, m.n_tex, 3, np.float32),
                "nrm": arr(m.nrm, m.n_nrm, 3, np.float32), "idx": arr(m.idx, m.n_tri, 9, np.uint32)}
    finally:
        L.tr_free_mesh(mp)


def load_tga(path):
    L = load_library()
    img = _lib.ImageRgb8()
    check(L.tr_load_tga_rgb8(os.fsencode(path), C.byref(img)))
    try:
        return np.ctypeslib.as_array(img.rgb, (img.h, img.w, 3)).copy()
    finally:
        L.tr_free_image(C.byref(img))


def save_tga(path, rgb):
    """Write an [H, W, 3] uint8 frame (row 0 = top, as get_frame_buffer returns it) as a 24-bit TGA."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    check(load_library().tr_save_tga_rgb8(os.fsencode(path), rgb.ctypes.data, rgb.shape[1], rgb.shape[0]))


def save_png(path, rgb):
    """Write an [H, W, 3] uint8 frame (row 0 = top) as an 8-bit RGB PNG."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    check(load_library().tr_save_png_rgb8(os.fsencode(path), rgb.ctypes.data, rgb.shape[1], rgb.shape[0]))


def load_assets(asset_path):
    """-p <asset dir>: returns (mesh, [texture, normal_map, normal_map_tangent, specular_map])."""
    mesh = load_obj(os.path.join(asset_path, ASSET_FILES[0]))
    texs = [load_tga(os.path.join(asset_path, f)) for f in ASSET_FILES[1:]]
    return mesh, texs
