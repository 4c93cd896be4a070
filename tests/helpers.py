"""Test-side helpers: independent asset loaders (PIL for TGA, a small Python OBJ reader),
asset discovery and image dumps.  Independent of both the oracle and the product loaders so
that it can cross-check them."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

TEX_FILES = ("texture.tga", "normal_map.tga", "normal_map_tangent.tga", "specular_map.tga")


def asset_root():
    """Directory holding <model>/model.obj etc.: $TR_ASSETS, the git-ignored copy made by
    __graft_entry__.build() (assets/_ref), or the reference checkout when present."""
    for p in (os.environ.get("TR_ASSETS"), os.path.join(REPO, "assets", "_ref"),
              "/root/reference/assets"):
        if p and os.path.isdir(p):
            return p
    return None


def asset_dir(name):
    root = asset_root()
    if root is None:
        return None
    d = os.path.join(root, name)
    return d if os.path.isfile(os.path.join(d, "model.obj")) else None


def load_obj_py(path):
    """obj-rs `parse_obj` semantics for the subset the path uses: v / vt / vn / f with
    v/vt/vn triples; indices zero based; only the first three vertices of a face are used
    (scene.rs:224-226)."""
    pos, tex, nrm, idx = [], [], [], []
    with open(path, "r") as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v":
                pos.append([np.float32(x) for x in parts[1:4]])
            elif parts[0] == "vt":
                v = [np.float32(x) for x in parts[1:4]]
                while len(v) < 3:
                    v.append(np.float32(0.0))
                tex.append(v)
            elif parts[0] == "vn":
                nrm.append([np.float32(x) for x in parts[1:4]])
            elif parts[0] == "f":
                tri = []
                for tok in parts[1:4]:
                    a, b, c = tok.split("/")
                    tri += [int(a) - 1, int(b) - 1, int(c) - 1]
                idx.append(tri)
    return {"pos": np.array(pos, np.float32).reshape(-1, 3),
            "tex": np.array(tex, np.float32).reshape(-1, 3),
            "nrm": np.array(nrm, np.float32).reshape(-1, 3),
            "idx": np.array(idx, np.uint32).reshape(-1, 9)}


def load_tga_pil(path):
    """image::open(path)?.into_rgb8(): rgb8, row 0 = top."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.array(im.convert("RGB"), dtype=np.uint8))


def load_assets_py(name):
    d = asset_dir(name)
    if d is None:
        return None
    mesh = load_obj_py(os.path.join(d, "model.obj"))
    texs = [load_tga_pil(os.path.join(d, f)) for f in TEX_FILES]
    return mesh, texs


def camera(angle):
    """app.rs:200-202"""
    a = np.float32(angle)
    return ([float(np.sin(a)), 0.0, float(np.cos(a))], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0])


def light(angle):
    """app.rs:203-207"""
    a = np.float32(angle)
    return [float(np.sin(a)), 0.0, float(np.cos(a))]


def save_png(path, rgb):
    from PIL import Image
    Image.fromarray(rgb).save(path)
