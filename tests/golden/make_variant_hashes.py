"""Generates tests/golden/variant_hashes.json: for ONE small scene per closure family, the sha256 fingerprints (as in
oracle_hashes.json) of the frame and the z bits under the normative oracle AND under each alternative reading of
nalgebra's operation order (TRO_VARIANT 1..6, oracle/tr_oracle.c; scripts/oracle_variants.py builds the libraries).
A maintainer with a Rust toolchain renders the same scene upstream ("Pinning the oracle upstream")
and looks the two hashes up here: the entry that matches names the reading upstream really has.

    python scripts/oracle_variants.py            # builds oracle/_variants/*.so (CPU, a few minutes)
    python tests/golden/make_variant_hashes.py   # rewrites the json next to this script
"""
import hashlib, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import oracle as O      # noqa: E402
from tests import helpers as H      # noqa: E402

VARIANTS = {0: "normative (what the GPU path reproduces)", 1: "dot3 associates right", 2: "4x4 gemv sums pairwise",
            3: "gemv accumulates from +0", 4: "normalize = v * (1/n)", 5: "3x3 inverse = cofactor * (1/det)",
            6: "4x4 inverse = cofactor / det"}
W, Hh, CA, LA = 640, 480, 0.7, -1.1   # a view where no matrix entry is exact (at angle 0 the gemv variants coincide)
PIPES = ("phong", "specular", "darboux", "shadow")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def main():
    inputs = H.load_assets_py("diablo")
    if inputs is None:
        raise SystemExit("the reference's assets are not available here")
    mesh, texs = inputs
    out = {"scene": "diablo.obj + its four textures, %dx%d, camera angle %.2f, light angle %.2f (tests/helpers.py camera / light: "
                    "app.rs:200-207)" % (W, Hh, CA, LA),
           "hash": "first 24 hex digits of sha256 over the raw bytes: fb = get_frame_buffer() rgb8 rows top to bottom; "
                   "z = the z buffer's f32 values as stored (index x + y * width, row 0 = bottom), little endian",
           "variants": {}}
    for k, name in VARIANTS.items():
        O.use_library(None if k == 0 else os.path.join(REPO, "oracle", "_variants", "libtr_oracle_v%d.so" % k))
        entry = {"reading": name}
        for pipe in PIPES:
            s = O.Scene(W, Hh, mesh, texs, pipe)
            s.clear(); s.set_light_direction(H.light(LA)); s.set_camera(*H.camera(CA))
            s.render()
            entry[pipe] = {"fb": sha(s.get_frame_buffer()), "z": sha(s.z_f32().view(np.uint32))}
            s.close()
        out["variants"][str(k)] = entry
    O.use_library(None)
    path = os.path.join(HERE, "variant_hashes.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
