"""Generates tests/golden/oracle_hashes.json: sha256 fingerprints of the CPU oracle's outputs
(frame buffer, z bits, winner indices, shadow bits) for fixed scenes.

The reference ships no golden vectors and cannot be run here (Rust, no toolchain), so these are
regression pins of OUR restatement, not upstream data (PARITY UNPINNED, oracle/tr_oracle.h).
They let `-m "not gpu"` tests detect any drift of the oracle, and `-m gpu` tests check the GPU
against the same fingerprints without re-deriving them.

    python tests/golden/make_golden.py          # rewrites the json next to this script
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

PIPES = ("default", "phong", "normal_map", "specular", "darboux", "shadow", "occlusion")
# (scene, width, height, camera angle, light angle)
CASES = [("synthetic", 320, 240, 0.4, -0.7), ("synthetic", 257, 129, -1.2, 0.9),
         ("diablo", 800, 800, 0.0, 0.0), ("diablo", 640, 480, 0.7, -1.1), ("african_head", 800, 800, 0.0, 0.0)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def scene_inputs(name):
    import tiny_renderer_amd as T
    from tests import helpers as H
    if name == "synthetic":
        return T.synthetic_scene(n_lat=12, n_lon=24, tex_size=256)
    return H.load_assets_py(name)


def fingerprints(name, W, Hh, ca, la, pipe, inputs):
    from oracle import oracle as O
    from tests import helpers as H
    mesh, texs = inputs
    s = O.Scene(W, Hh, mesh, texs, pipe)
    s.clear()
    s.set_light_direction(H.light(la))
    s.set_camera(*H.camera(ca))
    err = s.render()
    out = {"err": err, "fb": sha(s.get_frame_buffer()), "z": sha(s.z_f32().view(np.uint32)),
           "winner": sha(s.winner_u32()), "accepts": s.stats()[-1 if pipe in ("shadow", "occlusion") else 0]["frag_accept"]}
    if pipe in ("shadow", "occlusion"):
        out["shadow"] = sha(s.shadow_f32().view(np.uint32))
    return out


def key(name, W, Hh, ca, la, pipe):
    return "%s/%dx%d/cam%+.2f/light%+.2f/%s" % (name, W, Hh, ca, la, pipe)


def main():
    res = {}
    for name, W, Hh, ca, la in CASES:
        inputs = scene_inputs(name)
        if inputs is None:
            print("skipping %s: assets not available" % name)
            continue
        for pipe in PIPES:
            res[key(name, W, Hh, ca, la, pipe)] = fingerprints(name, W, Hh, ca, la, pipe, inputs)
    path = os.path.join(HERE, "oracle_hashes.json")
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print("wrote %d fingerprints to %s" % (len(res), path))


if __name__ == "__main__":
    main()
