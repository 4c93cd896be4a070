"""The C oracle against a SECOND restatement of the reference (tests/golden/second_opinion.py: numpy, written from the
Rust source, not from the C): frame, z bits and winning polygon per pixel must be identical.

Guards against a misreading of scene.rs / shader.rs / util.rs that the oracle and the kernels would share; it cannot pin
nalgebra's operation order (both apply SURVEY.md Appendix A's readings) -- DESIGN.md section 2."""
import numpy as np
import pytest

from tests import helpers as H
from tests.golden import second_opinion as S

PIPELINES = ("default", "phong", "normal_map", "specular", "darboux", "shadow", "occlusion")
ANGLES = ((0.0, 0.0), (0.7, -1.1))   # (camera, light): the reference's first frame, and one where no matrix is exact


def _both(W, Hh, mesh, texs, pipe, cam_angle, light_angle, twice=False):
    from oracle import oracle as O
    a, b = O.Scene(W, Hh, mesh, texs, pipe), S.Scene(W, Hh, mesh, texs, pipe)
    for s in (a, b):
        s.clear()
        s.set_light_direction(H.light(light_angle))
        s.set_camera(*H.camera(cam_angle))
    assert a.render() == 0
    b.render()
    if twice:   # render() does not clear: a second render from another angle depth-tests against the first (scene.rs:151)
        for s in (a, b):
            s.set_camera(*H.camera(cam_angle + 0.5))
            s.render()
    return a, b


def _compare(a, b, what):
    za, zb = a.z_f32().view(np.uint32).ravel(), b.buf["z"].view(np.uint32)
    assert np.array_equal(za, zb), "%s: z bits differ at %d pixels" % (what, int((za != zb).sum()))
    wa, wb = a.winner_u32().ravel(), b.winner
    assert np.array_equal(wa, wb), "%s: winner differs at %d pixels" % (what, int((wa != wb).sum()))
    fa, fb = a.get_frame_buffer(), b.get_frame_buffer()
    assert np.array_equal(fa, fb), "%s: rgb differs at %d pixels (max %d)" % (
        what, int((fa != fb).any(-1).sum()), int(np.abs(fa.astype(int) - fb.astype(int)).max()))
    assert (wa != 0xFFFFFFFF).sum() > 50, what


@pytest.mark.parametrize("pipe", PIPELINES)
@pytest.mark.parametrize("angles", ANGLES)
def test_oracle_agrees_with_the_second_restatement(small_synthetic, pipe, angles):
    mesh, texs = small_synthetic
    for (W, Hh) in ((64, 64), (201, 150)):
        a, b = _both(W, Hh, mesh, texs, pipe, *angles)
        _compare(a, b, "%s %dx%d angles %s" % (pipe, W, Hh, angles))
        if pipe in ("shadow", "occlusion"):
            sa, sb = a.shadow_f32().view(np.uint32).ravel(), b.buf["shadow"].view(np.uint32)
            assert np.array_equal(sa, sb), "shadow buffer bits differ at %d pixels" % int((sa != sb).sum())
        a.close()


@pytest.mark.parametrize("pipe", ("phong", "darboux"))
def test_second_restatement_accumulates_like_the_oracle(small_synthetic, pipe):
    mesh, texs = small_synthetic
    a, b = _both(96, 96, mesh, texs, pipe, 0.3, 0.2, twice=True)
    _compare(a, b, "two renders without a clear, " + pipe)
    a.close()


@pytest.mark.parametrize("pipe", PIPELINES)
def test_second_restatement_on_the_reference_model(diablo, pipe):
    """The reference's own model and maps (5 022 polygons, 1024^2 textures), small frame."""
    mesh, texs = diablo
    a, b = _both(96, 96, mesh, texs, pipe, 0.7, -1.1)
    _compare(a, b, "diablo " + pipe)
    a.close()
