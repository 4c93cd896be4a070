// tr_pk.h -- two-component f32 vectors for the kernels' per-pixel arithmetic.
//
// CDNA4's packed instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) apply one IEEE
// binary32 operation to each half of a register pair, with the same single rounding as the scalar
// forms, so a lane can carry two pixels through the reference's arithmetic bit for bit at half
// the instruction count -- k_tile is bound by vector-ALU issue, not by memory.  On the device `f2`
// is a clang ext_vector (hipcc emits the packed forms for its operators; -ffp-contract=off keeps
// a*b+c unfused); on the host it is a plain struct with the same per-component operations.
#pragma once

#include "tr_math.h"

namespace tr {

// f2s: the same two components as a plain struct -- every operation one scalar instruction per component.  A packed
// f32 instruction occupies the vector ALU for two issue slots on this machine (measured: v_pk_fma_f32 1.9 ns per
// wave-instruction and SIMD, v_fma_f32 1.0), so packing saves instructions, not cycles, and costs register moves
// wherever the two pixels' operands do not already sit in an aligned register pair (the survivors' records of the
// shading step: 32 v_mov per step) plus the even alignment of every pair.  The tile kernels of the light closures run
// their per-pixel arithmetic on f2s (k_tile, tile_scalar_pairs): 59 vector registers and no spills where the packed
// form needs 64 and spills three, 24.4 -> 22.4 us per frame at 4096^2 phong; the two-pixel closures (normal map,
// specular, darboux), whose long dependent chains of normalisations gain from the halved instruction count, stay
// packed (darboux 46.4 -> 49.7 us per frame as scalars).  The library is built -fno-slp-vectorize, or the compiler
// packs the scalar form again.  (profiles/r04_notes.md)
struct f2s {
    float x, y;
};
template <class V>
TR_HD V mk2v(float a, float b)
{
    V r;
    r.x = a;
    r.y = b;
    return r;
}
TR_HD f2s operator+(f2s a, f2s b) { return mk2v<f2s>(a.x + b.x, a.y + b.y); }
TR_HD f2s operator-(f2s a, f2s b) { return mk2v<f2s>(a.x - b.x, a.y - b.y); }
TR_HD f2s operator*(f2s a, f2s b) { return mk2v<f2s>(a.x * b.x, a.y * b.y); }
TR_HD f2s operator-(f2s a) { return mk2v<f2s>(-a.x, -a.y); }
TR_HD f2s fma2(f2s a, f2s b, f2s c) { return mk2v<f2s>(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)); }

#if defined(__HIP_DEVICE_COMPILE__) && !defined(TR_NO_PK)
#define TR_F2_PACKED 1
typedef float f2 __attribute__((ext_vector_type(2)));
TR_HD f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
#define TR_F2_PACKED 0
typedef f2s f2;
#endif
TR_HD f2 mk2(float a, float b) { return mk2v<f2>(a, b); }

template <class V>
TR_HD V splat2v(float a)
{
    return mk2v<V>(a, a);
}
TR_HD f2 splat2(float a) { return mk2(a, a); }
template <class V>
TR_HD f2 to_f2(V v)
{
    return mk2(v.x, v.y);
}

// The signed-zero repair of div_by as one bit operation: q0 = x * y always carries the sign of the
// quotient (no underflow here: |x| >= 1 or x = 0, |y| >= 2^-62), and the corrected q differs from it
// in sign only when x is a zero.
template <class V>
TR_HD V quotient_sign_from(V q, V q0)
{
    return mk2v<V>(copysignf(q.x, q0.x), copysignf(q.y, q0.y));
}

// ---------------------------------------------------------------------------------------------
// Correctly rounded 1/d and sqrt(s) for two values at once.  The device forms start from the
// hardware's 1-ulp estimates (v_rcp_f32, v_rsq_f32) and finish with fused residual corrections in
// packed arithmetic -- 2 + 4 and 2 + 7 instructions per PAIR instead of the 2 x 11 and 2 x 12 of the
// compiler's general-purpose expansions (which also scale for denormal and huge operands).  Valid
// for normal operands well inside the exponent range: |d| in [2^-42, 2^42], s in [2^-82, 2^82] --
// callers guard the range (tr_shaders.h, PairGuard) and fall back to '/' and sqrtf outside it.
// Proof: exhaustive on the device -- tr_selftest_device_unary compares every f32 of those ranges
// (1.4e9 + 2.8e9 values) with the compiler's correctly rounded '/' and sqrtf
// (tests/test_gpu_parity.py::test_pair_rcp_sqrt_exhaustive).  On the host they ARE '/' and sqrtf.
#if defined(__HIP_DEVICE_COMPILE__)
TR_HD f2 rcp2(f2 d)
{
    f2 y = mk2(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y));
    f2 e = fma2(-d, y, mk2(1.0f, 1.0f));
    y = fma2(e, y, y);
    e = fma2(-d, y, mk2(1.0f, 1.0f));
    return fma2(e, y, y);
}
TR_HD f2 sqrt2(f2 s)
{
    const f2 r = mk2(__builtin_amdgcn_rsqf(s.x), __builtin_amdgcn_rsqf(s.y));
    f2 g = s * r, h = r * mk2(0.5f, 0.5f);
    const f2 e = fma2(-h, g, mk2(0.5f, 0.5f));
    g = fma2(g, e, g);
    h = fma2(h, e, h);
    const f2 d = fma2(-g, g, s);
    return fma2(d, h, g);
}
#else
TR_HD f2 rcp2(f2 d) { return mk2(1.0f / d.x, 1.0f / d.y); }
TR_HD f2 sqrt2(f2 s) { return mk2(sqrtf(s.x), sqrtf(s.y)); }
#endif

// x / d for operands inside the guarded range and x != 0, y = rcp2(d): div_by2 without the signed-zero
// repair (Markstein: y = RN(1/d), q faithful after the first correction, exact residuals because
// |x| >= 2^-40 keeps them far above the subnormal grid).
TR_HD f2 div_by2_nonzero(f2 x, f2 d, f2 y)
{
    const f2 q0 = x * y;
    f2 e = fma2(-q0, d, x);
    f2 q = fma2(e, y, q0);
    e = fma2(-q, d, x);
    return fma2(e, y, q);
}

// (a.x*b.x + a.y*b.y) + a.z*b.z per component, the dot3 order
template <class V>
TR_HD V dot3_2(V ax, V ay, V az, V bx, V by, V bz)
{
    return (ax * bx + ay * by) + az * bz;
}

}  // namespace tr
