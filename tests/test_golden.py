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
