// tr_math.h -- f32 vector/matrix primitives in the reference's operation order.
//
// The reference does all arithmetic through nalgebra 0.31.4 (SURVEY.md Appendix A lists the
// call sites).  Every function here performs the same IEEE binary32 operations in the same
// order; the translation unit must be built with -ffp-contract=off and without fast-math so
// that no a*b+c is fused and '/' and sqrt stay correctly rounded.
//
// Usable from HIP device code and from plain host C++ (the tests build a host-side emulation
// of the kernels from these headers).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TR_HD __host__ __device__ __forceinline__
#else
#define TR_HD inline
#endif

namespace tr {

struct vec3 {
    float x, y, z;
};

TR_HD vec3 make3(float x, float y, float z)
{
    vec3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
TR_HD vec3 sub3(vec3 a, vec3 b) { return make3(a.x - b.x, a.y - b.y, a.z - b.z); }
TR_HD vec3 add3(vec3 a, vec3 b) { return make3(a.x + b.x, a.y + b.y, a.z + b.z); }
TR_HD vec3 scale3(vec3 a, float s) { return make3(a.x * s, a.y * s, a.z * s); }
// Vector3::dot, unrolled for dimension 3: (a0*b0 + a1*b1) + a2*b2
TR_HD float dot3(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
TR_HD vec3 cross3(vec3 a, vec3 b)
{
    return make3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// normalize(): n = sqrt(dot(v,v)), then a true division per component.
TR_HD vec3 normalize3(vec3 a)
{
    float n = sqrtf(dot3(a, a));
    return make3(a.x / n, a.y / n, a.z / n);
}

// Column-major 4x4 (element (r,c) = m[4*c + r]) times (x,y,z,w): nalgebra's gemv accumulates
// column by column: y = col0*x; y = col1*y' + y; ...
struct vec4 {
    float x, y, z, w;
};
TR_HD vec4 mul_m4_v4(const float *m, float x, float y, float z, float w)
{
    vec4 r;
    r.x = m[0] * x;
    r.y = m[1] * x;
    r.z = m[2] * x;
    r.w = m[3] * x;
    r.x = m[4] * y + r.x;
    r.y = m[5] * y + r.y;
    r.z = m[6] * y + r.z;
    r.w = m[7] * y + r.w;
    r.x = m[8] * z + r.x;
    r.y = m[9] * z + r.y;
    r.z = m[10] * z + r.z;
    r.w = m[11] * z + r.w;
    r.x = m[12] * w + r.x;
    r.y = m[13] * w + r.y;
    r.z = m[14] * w + r.z;
    r.w = m[15] * w + r.w;
    return r;
}

// Column-major 3x3 given as three columns.
TR_HD vec3 mul_m3_v3(vec3 c0, vec3 c1, vec3 c2, vec3 v)
{
    vec3 r = make3(c0.x * v.x, c0.y * v.x, c0.z * v.x);
    r = make3(c1.x * v.y + r.x, c1.y * v.y + r.y, c1.z * v.y + r.z);
    r = make3(c2.x * v.z + r.x, c2.y * v.z + r.y, c2.z * v.z + r.z);
    return r;
}

// Rust `as` casts: truncate toward zero, saturate, NaN -> 0.  On gfx950 that is exactly what
// v_cvt_i32_f32 / v_cvt_u32_f32 do (out-of-range saturates, NaN gives 0); the instructions are
// named explicitly because a plain C cast of an out-of-range value is undefined.  Checked on the
// device against the branchy host form by tr_selftest_device_math.
#if defined(__HIP_DEVICE_COMPILE__)
TR_HD int32_t f32_to_i32(float v)
{
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
TR_HD uint32_t f32_to_u32(float v)
{
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
TR_HD uint32_t f32_to_u8(float v)
{
    uint32_t r = f32_to_u32(v);
    return r < 255u ? r : 255u;
}
// `v as u8` into byte `byte` (0..3) of `into`, the other bytes kept: v_trunc_f32 + v_cvt_pk_u8_f32.  The conversion
// saturates to 0..255 and takes NaN to 0 like the Rust cast, but ROUNDS to nearest (0.5023 -> 1: found by the exhaustive
// check below), hence the truncation in front of it, after which it only sees integers.  Two instructions where
// f32_to_u8 and the shift-or take three; compared with them for EVERY f32 on the device (tr_selftest_device_unary,
// which = 2; tests/test_gpu_parity.py::test_pair_rcp_sqrt_exhaustive)
TR_HD uint32_t pack_u8(float v, uint32_t byte, uint32_t into)
{
    uint32_t r;
    float t;
    asm("v_trunc_f32 %0, %1" : "=v"(t) : "v"(v));
    asm("v_cvt_pk_u8_f32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "v"(byte), "v"(into));
    return r;
}
#else
TR_HD int32_t f32_to_i32(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)v;
}
TR_HD uint32_t f32_to_u32(float v)
{
    if (!(v == v)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    if (v <= 0.0f) return 0u;
    return (uint32_t)v;
}
TR_HD uint32_t f32_to_u8(float v)
{
    if (!(v == v)) return 0u;
    if (v >= 255.0f) return 255u;
    if (v <= 0.0f) return 0u;
    return (uint32_t)v;
}
TR_HD uint32_t pack_u8(float v, uint32_t byte, uint32_t into)
{
    return (into & ~(0xFFu << (8u * byte))) | (f32_to_u8(v) << (8u * byte));
}
#endif

// Loads and stores through the GLOBAL address space.  Pointers that come out of an argument table in memory are
// generic to the compiler, which then emits flat_load / flat_store: a 64-bit vector address per access (an extra
// two-slot v_lshl_add_u64), both the vector-memory and the LDS counter to wait for, and an aperture check per access.
// Everything these helpers touch lives in device (or mapped host) memory, never in LDS or scratch.
#ifndef TR_GLOBAL_AS
#define TR_GLOBAL_AS 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && TR_GLOBAL_AS
template <typename T>
TR_HD T gload(const T *p)
{
    return *(const __attribute__((address_space(1))) T *)p;
}
template <typename T>
TR_HD void gstore(T *p, T v)
{
    *(__attribute__((address_space(1))) T *)p = v;
}
#else
template <typename T>
TR_HD T gload(const T *p)
{
    return *p;
}
template <typename T>
TR_HD void gstore(T *p, T v)
{
    *p = v;
}
#endif

// a * b for operands below 2^24 (texture and frame dimensions, record slots): one full-rate
// instruction on the device, where a 32 x 32 bit multiply takes four.
TR_HD uint32_t mul24(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}

// Correctly rounded x / d for many numerators and one divisor.  y = RN(1/d) comes from one IEEE
// division; q0 = RN(x*y) is corrected twice with exact FMA residuals (Markstein: with
// y = RN(1/d) and q faithful, RN(q + (x - d*q)*y) = RN(x/d); the first correction makes q
// faithful).  5 instructions per quotient instead of the 11 of a full division.  Requires finite
// operands without overflow/underflow: the coverage loop's are integer valued with |d| >= 1.
// tests/test_coverage_math.py checks bit equality with '/' on 10^8 adversarial pairs.
struct Recip {
    float d, y;
};
TR_HD Recip recip_of(float d)
{
    Recip r;
    r.d = d;
    r.y = 1.0f / d;
    return r;
}
TR_HD float div_by(float x, Recip r)
{
    const float q0 = x * r.y;
    float e = fmaf(-q0, r.d, x);
    float q = fmaf(e, r.y, q0);
    e = fmaf(-q, r.d, x);
    q = fmaf(e, r.y, q);
    return x == 0.0f ? q0 : q;  // the residual form turns -0/d into +0; the product keeps the sign
}

// util.rs:7-13 with color_2 = (0,0,0), one channel: (t*c + (1-t)*0.0) as u8.  The second term is a zero for every
// finite t (either sign: added to a non-zero product it changes nothing, added to a zero product the sum is a zero,
// and every zero casts to 0) and NaN for t = +-inf and NaN, which makes the channel 0.  Without it NaN and -inf give
// 0 anyway (NaN -> 0; -inf * c is -inf or NaN -> 0), and only t = +inf differs (+inf * c = +inf -> 255 for c > 0).
// So: the weight with +inf replaced by NaN, times the channel -- a compare and a select per PIXEL where the second
// term cost two operations per pixel and an addition per channel (tests/test_coverage_math.py::
// test_blend_without_the_zero_term compares both forms for every channel value over the special and 10^6 random weights).
#ifndef TR_BLEND_FAST
#define TR_BLEND_FAST 1
#endif
TR_HD float blend_weight(float t)
{
    return t == __builtin_inff() ? __builtin_nanf("") : t;
}
TR_HD uint32_t blend_black(uint32_t c, float t)
{
#if TR_BLEND_FAST
    return f32_to_u8(blend_weight(t) * (float)c);
#else
    return f32_to_u8(t * (float)c + (1.0f - t) * 0.0f);
#endif
}
// (the literal form, for the test that compares the two)
TR_HD uint32_t blend_black_literal(uint32_t c, float t)
{
    return f32_to_u8(t * (float)c + (1.0f - t) * 0.0f);
}

TR_HD uint32_t f32_bits(float f)
{
    union {
        float f;
        uint32_t u;
    } c;
    c.f = f;
    return c.u;
}
TR_HD float bits_f32(uint32_t u)
{
    union {
        float f;
        uint32_t u;
    } c;
    c.u = u;
    return c.f;
}

#define TR_F32_MIN_BITS 0xFF7FFFFFu /* f32::MIN */

}  // namespace tr
