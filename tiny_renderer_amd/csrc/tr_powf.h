// tr_powf.h -- powf with the host C library's result, bit for bit, on the device.
//
// The specular closure raises to a power (shader.rs:525); the reference's f32::powf is the
// platform libm's powf, and the CPU oracle calls the same function.  This is that function's
// algorithm -- glibc 2.28+ sysdeps/ieee754/flt-32/e_powf.c (S. Nagy): log2 through a 16-entry
// table and a degree-5 polynomial, exp2 through a 32-entry table and a cubic, in double precision,
// multiply-adds fused as in the FMA build that glibc selects on any x86-64 CPU of the last decade --
// restated with the constants gen_powf_tables.py read out of the installed libm.  An independent
// model of it agreed with the host's powf on 3.3e9 random arguments of the closure's domain
// (0 < x <= 1, y = 1..255); tests/test_host_side.py repeats a shorter run against the function below.
//
// Without the tables (TR_POWF_EXACT == 0) tr_powf is the device library's powf: within 1 ulp.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "tr_math.h"

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define TR_POWF_CONST static __device__ const
#else
#define TR_POWF_CONST static const
#endif
#if defined(__has_include)
#if __has_include("tr_powf_tables.inc")
#include "tr_powf_tables.inc"
#endif
#endif
#ifndef TR_POWF_EXACT
#define TR_POWF_EXACT 0
#endif

namespace tr {

#if TR_POWF_EXACT

TR_HD uint64_t f64_bits(double d)
{
    union {
        double d;
        uint64_t u;
    } c;
    c.d = d;
    return c.u;
}
TR_HD double bits_f64(uint64_t u)
{
    union {
        double d;
        uint64_t u;
    } c;
    c.u = u;
    return c.d;
}

// 0: y is not an integer, 1: odd, 2: even (e_powf.c checkint)
TR_HD int powf_checkint(uint32_t iy)
{
    const int e = (int)(iy >> 23 & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}

TR_HD bool powf_zeroinfnan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }

TR_HD float tr_powf(float x, float y)
{
    uint32_t sign_bias = 0;
    uint32_t ix = f32_bits(x);
    const uint32_t iy = f32_bits(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
        // x < 0x1p-126, inf or nan; or y is 0, inf or nan
        if (powf_zeroinfnan(iy)) {
            if (2u * iy == 0u) return 1.0f;           // (signalling NaNs do not occur: quiet arithmetic)
            if (ix == 0x3f800000u) return 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (powf_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && powf_checkint(iy) == 1) x2 = -x2;
            if (2u * ix == 0u && (iy & 0x80000000u)) return (ix & 0x80000000u) && powf_checkint(iy) == 1 ? -INFINITY : INFINITY;
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {
            const int yint = powf_checkint(iy);
            if (yint == 0) return bits_f32(0x7FC00000u);  // invalid: NaN
            if (yint == 1) sign_bias = 1u << (5 + 11);
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {
            ix = f32_bits(x * 8388608.0f) & 0x7fffffffu;  // normalise a subnormal
            ix -= 23u << 23;
        }
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> (23 - 4)) % 16u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int32_t k = (int32_t)top >> 23;
    const double z = (double)bits_f32(iz);
    const double r = fma(z, tr_powf_invc[i], -1.0);
    const double y0 = tr_powf_logc[i] + (double)k;
    const double r2 = r * r;
    double yl = fma(tr_powf_log2_poly[0], r, tr_powf_log2_poly[1]);
    const double p = fma(tr_powf_log2_poly[2], r, tr_powf_log2_poly[3]);
    const double r4 = r2 * r2;
    double q = fma(tr_powf_log2_poly[4], r, y0);
    q = fma(p, r2, q);
    yl = fma(yl, r4, q);
    const double ylogx = (double)y * yl;
    if ((f64_bits(ylogx) >> 47 & 0xffffu) >= (0x405f800000000000ull /* 126.0 */ >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -INFINITY : INFINITY;   // __math_oflowf
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                         // __math_uflowf
    }
    // exp2_inline
    double kd = ylogx + tr_powf_exp2_shift;
    const uint64_t ki = f64_bits(kd);
    kd -= tr_powf_exp2_shift;
    const double rr = ylogx - kd;
    uint64_t t = tr_powf_exp2_tab[ki % 32u];
    t += (ki + sign_bias) << (52 - 5);
    const double s = bits_f64(t);
    const double zz = fma(tr_powf_exp2_poly[0], rr, tr_powf_exp2_poly[1]);
    const double rr2 = rr * rr;
    double ye = fma(tr_powf_exp2_poly[2], rr, 1.0);
    ye = fma(zz, rr2, ye);
    return (float)(ye * s);
}

#else

TR_HD float tr_powf(float x, float y) { return powf(x, y); }

#endif

}  // namespace tr
