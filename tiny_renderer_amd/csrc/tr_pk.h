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

#if defined(__HIP_DEVICE_COMPILE__)
typedef float f2 __attribute__((ext_vector_type(2)));
TR_HD f2 mk2(float a, float b)
{
    f2 r;
    r.x = a;
    r.y = b;
    return r;
}
TR_HD f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct f2 {
    float x, y;
};
TR_HD f2 mk2(float a, float b)
{
    f2 r;
    r.x = a;
    r.y = b;
    return r;
}
TR_HD f2 operator+(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
TR_HD f2 operator-(f2 a, f2 b) { return mk2(a.x - b.x, a.y - b.y); }
TR_HD f2 operator*(f2 a, f2 b) { return mk2(a.x * b.x, a.y * b.y); }
TR_HD f2 operator-(f2 a) { return mk2(-a.x, -a.y); }
TR_HD f2 fma2(f2 a, f2 b, f2 c) { return mk2(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)); }
#endif

TR_HD f2 splat2(float a) { return mk2(a, a); }

// div_by (tr_math.h) for two numerators; d and y = RN(1/d) may differ per component.
TR_HD f2 div_by2(f2 x, f2 d, f2 y)
{
    const f2 q0 = x * y;
    f2 e = fma2(-q0, d, x);
    f2 q = fma2(e, y, q0);
    e = fma2(-q, d, x);
    q = fma2(e, y, q);
    return mk2(x.x == 0.0f ? q0.x : q.x, x.y == 0.0f ? q0.y : q.y);
}

// The signed-zero repair of div_by as one bit operation: q0 = x * y always carries the sign of the
// quotient (no underflow here: |x| >= 1 or x = 0, |y| >= 2^-62), and the corrected q differs from it
// in sign only when x is a zero.
TR_HD f2 quotient_sign_from(f2 q, f2 q0) { return mk2(copysignf(q.x, q0.x), copysignf(q.y, q0.y)); }

// The same without the signed-zero repair: a quotient of a -0 numerator may come out as +0.
// For consumers that only compare the result (or sums built from it) with IEEE ordering, where
// +0 and -0 are the same value -- the coverage loop's depth test.
TR_HD f2 div_by2_unsigned_zero(f2 x, f2 d, f2 y)
{
    const f2 q0 = x * y;
    f2 e = fma2(-q0, d, x);
    f2 q = fma2(e, y, q0);
    e = fma2(-q, d, x);
    return fma2(e, y, q);
}

// (a.x*b.x + a.y*b.y) + a.z*b.z per component, the dot3 order
TR_HD f2 dot3_2(f2 ax, f2 ay, f2 az, f2 bx, f2 by, f2 bz) { return (ax * bx + ay * by) + az * bz; }

}  // namespace tr
