"""The product's arithmetic substitutions, checked on the CPU against the dividing forms:
  * division-free inside tests (tr_shaders.h `covers`, and `covers_oriented`: the orientation-
    normalised three-way-minimum form of the tile kernel) vs the reference's sign tests on the
    divided barycentric coordinates (scene.rs:192-196, 245) as restated by the oracle;
  * shared-reciprocal division (tr_math.h `div_by`) vs IEEE '/';
  * depth_order_key is monotone and folds -0.0 onto +0.0."""
import ctypes as C

import numpy as np

from oracle import oracle as O
from tests import emul_bind as E


def _check_triangles(tris, pts):
    Lo, Le = O.lib(), E.lib()
    bo = (C.c_float * 3)()
    be = (C.c_float * 3)()
    n_cov = 0
    for t in tris:
        r = (C.c_int32 * 6)(*[int(v) for v in t])
        for (px, py) in pts(t):
            Lo.tro_barycentric(r, int(px), int(py), bo)
            ref_inside = not (bo[0] < 0.0 or bo[1] < 0.0 or bo[2] < 0.0)
            got = Le.tr_emul_covers(r, int(px), int(py), be)
            if got < 0:   # degenerate: the oracle reports (-1, 1, 1)
                assert list(bo) == [-1.0, 1.0, 1.0]
                continue
            # bit 0: `covers`; bit 1: `covers_oriented`, the form the tile kernel evaluates
            assert (got & 1) == ref_inside and (got >> 1) == ref_inside, (list(t), px, py, list(bo), got)
            assert np.array_equal(np.array(bo, np.float32).view(np.uint32), np.array(be, np.float32).view(np.uint32))
            n_cov += got & 1
    return n_cov


def test_covers_matches_divided_sign_tests_small():
    rng = np.random.default_rng(7)
    tris = rng.integers(-40, 140, size=(300, 6))

    def pts(t):
        return [(x, y) for x in range(0, 100, 7) for y in range(0, 100, 9)] + \
               [(t[0], t[1]), (t[2], t[3]), (t[4], t[5]), ((t[0] + t[2]) // 2, (t[1] + t[3]) // 2)]
    assert _check_triangles(tris, pts) > 1000


def test_covers_matches_divided_sign_tests_large_coordinates():
    """Products beyond 2^24 round in f32 (8192^2 frames, off-screen vertices)."""
    rng = np.random.default_rng(8)
    tris = rng.integers(-20000, 30000, size=(300, 6))

    def pts(t):
        p = [(int(rng.integers(0, 8192)), int(rng.integers(0, 8192))) for _ in range(150)]
        # points on and next to the edges, where the sign tests are decided by one ulp
        for a, b in ((0, 2), (2, 4), (4, 0)):
            for f in (0.25, 0.5, 0.75):
                x = int(t[a] + f * (t[b] - t[a]))
                y = int(t[a + 1] + f * (t[b + 1] - t[a + 1]))
                p += [(x, y), (x + 1, y), (x, y + 1), (x - 1, y - 1)]
        return p
    assert _check_triangles(tris, pts) > 1000


def test_shared_reciprocal_division_is_correctly_rounded():
    L = E.lib()
    rng = np.random.default_rng(9)
    n = 200000
    # coverage operands: integer valued, |d| >= 1; include all-ones mantissas and near-midpoint quotients
    d = np.concatenate([rng.integers(1, 1 << 26, n // 2).astype(np.float32),
                        (rng.integers(1, 1 << 24, n // 4) * 2 - 1).astype(np.float32) * np.float32(1 << 10),
                        np.float32(2.0) ** rng.integers(0, 40, n // 4) * np.float32(1.9999999)])
    d = np.trunc(d) * rng.choice([-1.0, 1.0], d.size).astype(np.float32)
    x = np.trunc((rng.standard_normal(d.size) * 2.0 ** rng.integers(0, 40, d.size)).astype(np.float32))
    want = (x / d).astype(np.float32)
    got = np.array([L.tr_emul_div_by(float(a), float(b)) for a, b in zip(x, d)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_depth_order_key():
    L = E.lib()
    vals = np.array([-np.inf, -3.4028235e38, -1e10, -1.0, -1e-30, -0.0, 0.0, 1e-30, 1.0, 254.5, 1e10, np.inf], np.float32)
    keys = [L.tr_emul_depth_order_key(float(v)) for v in vals]
    assert keys[5] == keys[6]                       # -0.0 == +0.0 for `z <= zbuf`
    ks = keys[:5] + keys[6:]
    assert all(a < b for a, b in zip(ks, ks[1:]))


def test_powf_reproduces_the_host_libm():
    """tr_powf.h restates glibc's powf with the constants read out of the installed libm
    (csrc/gen_powf_tables.py); the specular closure uses it so that the device returns what the
    reference's f32::powf (= the platform powf) returns.  Domain of the closure: base in [0, 1]
    (max(r.z, 0) of a unit vector), exponent 0..255; special values on top."""
    L = E.lib()
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(5)
    n = 400000
    x = np.concatenate([rng.random(n), 1.0 - rng.random(n) * 2.0 ** -rng.integers(1, 24, n), 2.0 ** -rng.uniform(0, 149, n),
                        [0.0, 1.0, 1.0000001, 1e-45, 1.1754942e-38, 0.5, np.inf, np.nan, 2.0, -0.0]]).astype(np.float32)
    y = np.concatenate([rng.integers(0, 256, 3 * n), [0, 0, 255, 7, 255, 0, 3, 2, 200, 5]]).astype(np.float32)
    out = np.zeros(x.size, np.float32)
    exact = L.tr_emul_powf(x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
    if not exact:
        import pytest
        pytest.skip("the installed libm's powf tables were not recognised: the device library's powf is used (1 ulp)")
    ref = np.array([libm.powf(float(a), float(b)) for a, b in zip(x[::7], y[::7])], np.float32)
    got = out[::7]
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), list(zip(x[::7][~same][:5], y[::7][~same][:5], got[~same][:5], ref[~same][:5]))
