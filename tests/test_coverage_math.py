"""The product's arithmetic substitutions, checked on the CPU against the dividing forms:
  * division-free inside tests (tr_shaders.h `covers`, and `covers_oriented`: the orientation-
    normalised three-way-minimum form of the tile kernel) vs the reference's sign tests on the
    divided barycentric coordinates (scene.rs:192-196, 245) as restated by the oracle;
  * shared-reciprocal division vs IEEE '/': `div_by` (tr_math.h) on the coverage loop's integer-valued
    operands and `div_by2_nonzero` (tr_pk.h) on general operands inside the two-pixel closures' guard
    range; decode_normal's two-texel form for all 2^24 texels.
On the host rcp2 / sqrt2 are '/' and sqrtf; their device forms are checked exhaustively on the GPU
(tests/test_gpu_parity.py::test_pair_rcp_sqrt_exhaustive)."""
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
            # both bits: `covers_oriented` on the orientation-normalised polygon, the form the tile kernel
            # evaluates; the barycentrics are the shading phase's (`barycentric2`)
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


def _div(which, x, d):
    x = np.ascontiguousarray(x, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    out = np.empty_like(x)
    E.lib().tr_emul_div(which, x.ctypes.data, d.ctypes.data, out.ctypes.data, x.size)
    return out


def test_shared_reciprocal_division_is_correctly_rounded():
    """div_by (coverage / barycentrics): integer-valued operands, |d| >= 1, zero numerators keep their sign."""
    rng = np.random.default_rng(9)
    n = 2000000
    d = np.concatenate([rng.integers(1, 1 << 26, n // 2).astype(np.float32),
                        (rng.integers(1, 1 << 24, n // 4) * 2 - 1).astype(np.float32) * np.float32(1 << 10),
                        np.float32(2.0) ** rng.integers(0, 40, n // 4) * np.float32(1.9999999)])
    d = np.trunc(d) * rng.choice([-1.0, 1.0], d.size).astype(np.float32)
    x = np.trunc((rng.standard_normal(d.size) * 2.0 ** rng.integers(0, 40, d.size)).astype(np.float32))
    x[::97] = 0.0
    x[1::97] = -0.0
    want = (x / d).astype(np.float32)
    assert np.array_equal(_div(0, x, d).view(np.uint32), want.view(np.uint32))


def test_pair_closure_division_on_general_operands():
    """div_by2_nonzero (the two-pixel closures: normalisations, the 3x3 inverse): ANY f32 operands inside
    the guard range 2^-40 <= |x|, |d| <= 2^41, x != 0 -- not only integers.  1.1e8 pairs, three
    quarters of them adversarial: quotients placed within a few ulp of a rounding boundary
    (x = RN(d * m) for m a float or a midpoint between two floats, nudged by -2..2 ulp), divisors with
    all-ones or single-bit significands.  Must equal IEEE x / d bit for bit."""
    rng = np.random.default_rng(21)
    chunk, total = 4000000, 0

    def mant(k):   # random significands in [1, 2), a share of them all-ones / single-bit
        m = (1.0 + rng.integers(0, 1 << 23, k) * 2.0 ** -23)
        sel = rng.integers(0, 8, k)
        m = np.where(sel == 0, 2.0 - 2.0 ** -23 * rng.integers(1, 4, k), m)
        m = np.where(sel == 1, 1.0 + 2.0 ** -rng.integers(1, 24, k).astype(np.float64), m)
        return m

    for it in range(30):
        d = (mant(chunk) * 2.0 ** rng.integers(-40, 41, chunk) * rng.choice([-1.0, 1.0], chunk)).astype(np.float32)
        if it % 4 == 3:   # every fourth chunk: unrelated random operands
            x = (mant(chunk) * 2.0 ** rng.integers(-40, 41, chunk) * rng.choice([-1.0, 1.0], chunk)).astype(np.float32)
        else:
            q = (mant(chunk) * 2.0 ** rng.integers(-20, 21, chunk)).astype(np.float32)
            # a quotient target: q itself or the midpoint between q and its successor; x = RN(target * d) +- ulps
            mid = rng.integers(0, 2, chunk).astype(bool)
            target = np.where(mid, q.astype(np.float64) * (1.0 + 2.0 ** -24), q.astype(np.float64))
            x = (target * d.astype(np.float64)).astype(np.float32)
            x = (x.view(np.int32) + rng.integers(-2, 3, chunk).astype(np.int32)).view(np.float32)
        ok = (np.abs(x) >= 2.0 ** -40) & (np.abs(x) <= 2.0 ** 41) & np.isfinite(x)
        x, d = x[ok], d[ok]
        with np.errstate(all="ignore"):
            want = (x / d).astype(np.float32)   # numpy float32 division is IEEE
        got = _div(1, x, d)
        bad = got.view(np.uint32) != want.view(np.uint32)
        assert not bad.any(), (x[bad][:3], d[bad][:3], got[bad][:3], want[bad][:3])
        total += int(x.size)
    assert total >= 100000000


def test_decode_normal_pair_form_all_texels():
    """decode_normal for two texels at once (channel / 255 through the constant's reciprocal, shared
    reciprocal of the norm) against the plain form for every one of the 2^24 rgb values."""
    assert E.lib().tr_emul_decode_normal_mismatches(0, 1 << 24) == 0


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


def test_shadow_lookup_through_fast_clear_flags_equals_plain_lookup():
    """shadow_fetch with the shadow buffer's per-tile fast-clear flags (stale memory behind a raised flag) returns
    what the plain lookup returns in the fully written buffer -- for every kind of index the reference's wrapping
    u32 arithmetic (shader.rs:774-775) can produce: inside the frame, a column beyond the row (lands in a later
    row), a row that wraps around 2^32 back into the buffer (y = k * 2^32 / W: every large f32 is such a
    multiple), out of range (flagged), negative and NaN coordinates (cast to 0)."""
    L = E.lib()
    rng = np.random.default_rng(5)
    for W, Hh in ((512, 512), (640, 100), (130, 70), (4096, 64)):
        ntx, nty = (W + 127) // 128, (Hh + 15) // 16
        plain = rng.standard_normal(W * Hh).astype(np.float32)
        flags = (rng.random(ntx * nty) < 0.5).astype(np.uint32) * np.uint32(0xFFFFFFFF)
        tile_of = (np.arange(Hh)[:, None] // 16) * ntx + (np.arange(W)[None, :] // 128)
        clean = flags[tile_of].astype(bool).reshape(-1)
        plain[clean] = np.float32(np.finfo(np.float32).min)          # what a clean tile logically holds
        stale = plain.copy()
        stale[clean] = np.float32(123.0)                              # ... and garbage where the flag is up
        xs = [rng.uniform(-3, W + 3, 4000), rng.uniform(W, 40 * W, 2000), rng.uniform(0, W, 3000), rng.uniform(0, W, 500)]
        ys = [rng.uniform(-3, Hh + 3, 4000), rng.uniform(0, Hh / 2, 2000),
              (rng.integers(1, 64, 3000) * (2.0 ** 32 / W)) + rng.integers(0, Hh, 3000),   # wraps around 2^32
              rng.uniform(Hh, 1e9, 500)]
        x = np.concatenate(xs + [[np.nan, 0.0, -1e30, 1e30]]).astype(np.float32)
        y = np.concatenate(ys + [[0.0, np.nan, 5.0, 1e30]]).astype(np.float32)
        bad = L.tr_emul_shadow_fetch_mismatches(plain.ctypes.data, stale.ctypes.data, flags.ctypes.data, W, Hh,
                                                x.ctypes.data, y.ctypes.data, len(x))
        assert bad == 0, (W, Hh, bad)
        # the wrapped rows really land inside the buffer for some lookups (otherwise the case proves nothing)
        iy = np.minimum(np.round(ys[2].astype(np.float32)).astype(np.float64), 2.0 ** 32 - 1).astype(np.uint64)
        ix = np.round(xs[2].astype(np.float32)).astype(np.uint64)
        idx = (ix + iy * np.uint64(W)) % np.uint64(2 ** 32)
        assert (idx < W * Hh).sum() > 100


def test_darboux_third_column_only_matters_for_zero_sums():
    """shader.rs:632-643 multiply the inverse of the local basis by vectors whose third component is 0.0, so each
    component of local_x / local_y is (i0 * d1 + i1 * d2) + q * 0.0 with q = cofactor / det of the inverse's third column.
    The two-pixel darboux closure leaves that column out (tr_shaders.h, fragment_color_pair).  Why that is exact:
      * q is finite there (cofactors of unit rows over a determinant inside the guard [2^-40, 2^40]), so q * 0.0 is a
        signed zero, and adding a zero of either sign to a NON-zero sum returns the sum, bit for bit;
      * the only sums it can change are zeros: (-0) + (+0) = +0 -- and a zero component of local_x / local_y fails the
        guard of the normalisation that consumes it (zero < 2^-40), which sends the pixel to the plain closure, where all
        nine quotients are formed.
    Enumerated here; the closures themselves are compared on whole models by tests/test_emulation.py and on the GPU."""
    rng = np.random.default_rng(11)
    # a non-zero sum plus a signed zero: unchanged, whatever the magnitudes and signs
    s = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 200000).astype(np.float32),
                        np.array([np.finfo(np.float32).tiny, -np.finfo(np.float32).tiny, 1e-45, -1e-45, 3.4e38, -3.4e38], np.float32)])
    s = s[s != 0]
    q = (rng.standard_normal(len(s)).astype(np.float32) * np.float32(2.0) ** rng.integers(-41, 41, len(s)).astype(np.float32))
    term = q * np.float32(0.0)                       # +0 or -0 by the sign of q
    assert set(np.unique(term.view(np.uint32))) <= {0, 0x80000000} and len(np.unique(term.view(np.uint32))) == 2
    assert np.array_equal((s + term).view(np.uint32), s.view(np.uint32))
    # the four zero cases: the sum's sign survives only with a term of the same sign -- these are the cases the guard
    # hands to the plain closure (a zero operand of normalize3p: PAIR_GUARD_LO = 2^-40 > 0)
    pz, nz = np.float32(0.0), np.float32(-0.0)
    cases = {(a.tobytes(), b.tobytes()): (a + b) for a in (pz, nz) for b in (pz, nz)}
    assert np.signbit(cases[(nz.tobytes(), nz.tobytes())]) and not np.signbit(cases[(nz.tobytes(), pz.tobytes())])
    assert not np.signbit(cases[(pz.tobytes(), nz.tobytes())]) and not np.signbit(cases[(pz.tobytes(), pz.tobytes())])
    assert np.float32(0.0) < np.float32(2.0) ** np.float32(-40)
    # q finite: |cofactor| <= 2 (differences of products of unit-vector components), |det| >= 2^-40
    assert np.isfinite(np.float32(2.0) / np.float32(2.0) ** np.float32(-40))


def test_blend_without_the_zero_term():
    """util.rs:7-13 with color_2 = black: (t * c + (1 - t) * 0.0) as u8.  The product leaves the second term out and
    replaces t = +inf by NaN instead (tr_math.h blend_black): identical bytes for every channel value 0..255 over the
    special weights (zeros, infinities, NaNs, subnormals, the ends of the range, the neighbourhood of 1 and of k/255)
    and a million random bit patterns -- and the literal form is what the oracle's tro_color_blend computes."""
    Le = E.lib()
    special = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754944e-38, 3.4028235e38, -3.4028235e38,
               255.0, 256.0, 1.0 / 255.0, 0.5, 0.99999994, 1.0000001, 2.0, 1e30, -1e30, 1.3e36, 1.4e36]
    t = np.array(special, np.float32)
    nan_payloads = np.array([0x7FC00001, 0xFFC00000, 0x7F800001, 0xFF800001], np.uint32).view(np.float32)
    near = np.nextafter(np.float32(1.0) * np.arange(0, 256, dtype=np.float32) / np.float32(255.0), np.float32(2.0), dtype=np.float32)
    rng = np.random.default_rng(5)
    rnd = rng.integers(0, 2 ** 32, 1_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    t = np.ascontiguousarray(np.concatenate([t, nan_payloads, near, rnd]))
    assert Le.tr_emul_blend_mismatches(t.ctypes.data, t.size) == 0
    # the only weight for which the second term matters, and what it does there
    assert Le.tr_emul_blend(200, np.inf, 1) == 0 and Le.tr_emul_blend(200, np.inf, 0) == 0
    assert Le.tr_emul_blend(200, -np.inf, 1) == 0 and Le.tr_emul_blend(200, 3.0, 0) == 255
    # the literal form against numpy's f32 arithmetic with Rust's saturating cast
    c = np.arange(256, dtype=np.float32)
    for tv in (0.3, 0.99999994, 1.0, 1.5, -0.2, 1e30):
        tv = np.float32(tv)
        with np.errstate(all="ignore"):
            v = tv * c + (np.float32(1.0) - tv) * np.float32(0.0)
        want = np.clip(np.trunc(np.nan_to_num(v, nan=0.0, posinf=255.0, neginf=0.0)), 0, 255).astype(np.uint32)
        got = np.array([Le.tr_emul_blend(int(k), float(tv), 1) for k in range(256)], np.uint32)
        assert np.array_equal(got, want)
