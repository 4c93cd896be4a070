"""Committed fingerprints (tests/golden/oracle_hashes.json, made by tests/golden/make_golden.py):
the oracle must keep reproducing them (CPU), and the GPU must produce frames with the same
fingerprints (GPU; specular's RGB only when the library reproduces the host libm's powf,
tr_specular_exact(): with the device library's powf it is within 1 LSB and checked elsewhere)."""
import json
import os

import numpy as np
import pytest

from tests import helpers as H
from tests.golden import make_golden as G

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_hashes.json")))


def _cases():
    for name, W, Hh, ca, la in G.CASES:
        for pipe in G.PIPES:
            yield name, W, Hh, ca, la, pipe


@pytest.mark.parametrize("name,W,Hh,ca,la,pipe", list(_cases()))
def test_oracle_reproduces_fingerprints(built, name, W, Hh, ca, la, pipe):
    inputs = G.scene_inputs(name)
    if inputs is None:
        pytest.skip("reference assets not available")
    assert G.fingerprints(name, W, Hh, ca, la, pipe, inputs) == GOLD[G.key(name, W, Hh, ca, la, pipe)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,W,Hh,ca,la,pipe", list(_cases()))
def test_gpu_matches_fingerprints(built, name, W, Hh, ca, la, pipe):
    import tiny_renderer_amd as T
    inputs = G.scene_inputs(name)
    if inputs is None:
        pytest.skip("reference assets not available")
    gold = GOLD[G.key(name, W, Hh, ca, la, pipe)]
    mesh, texs = inputs
    s = T.Scene(W, Hh, mesh, texs, pipe, winner_tap=True)
    s.clear()
    s.set_light_direction(H.light(la))
    s.set_camera(*H.camera(ca))
    s.render()
    fb = s.get_frame_buffer()
    assert G.sha(s.read_z_f32().view(np.uint32)) == gold["z"]
    assert G.sha(s.read_winner_u32()) == gold["winner"]
    if "shadow" in gold:
        assert G.sha(s.read_shadow_f32().view(np.uint32)) == gold["shadow"]
    if pipe != "specular" or T.load_library().tr_specular_exact():
        assert G.sha(fb) == gold["fb"]
    s.close()


def test_variant_hashes_are_consistent_with_the_oracle_fingerprints():
    """tests/golden/variant_hashes.json (what every alternative reading of nalgebra's operation order would hash to for
    the scene of INTEGRATION.md's Rust test): its normative entry is the committed oracle fingerprint of that scene, the
    two readings the Rust test can tell apart really hash differently, and the text of INTEGRATION.md quotes the
    normative hashes."""
    here = os.path.dirname(os.path.abspath(__file__))
    v = json.load(open(os.path.join(here, "golden", "variant_hashes.json")))["variants"]
    for pipe in ("phong", "specular", "darboux", "shadow"):
        gold = GOLD["diablo/640x480/cam+0.70/light-1.10/%s" % pipe]
        assert v["0"][pipe]["fb"] == gold["fb"] and v["0"][pipe]["z"] == gold["z"], pipe
    assert v["1"]["phong"]["z"] != v["0"]["phong"]["z"] and v["1"]["phong"]["fb"] == v["0"]["phong"]["fb"]
    assert v["2"]["phong"]["fb"] != v["0"]["phong"]["fb"] and v["2"]["phong"]["z"] != v["0"]["phong"]["z"]
    text = open(os.path.join(os.path.dirname(here), "INTEGRATION.md")).read()
    assert v["0"]["phong"]["fb"] in text and v["0"]["phong"]["z"] in text
    assert v["1"]["phong"]["z"][:8] in text and v["2"]["phong"]["fb"][:8] in text
