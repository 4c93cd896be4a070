// tr_shaders.h -- the vertex and fragment stages of the seven pipelines, plus the coverage
// arithmetic, as host/device inline functions.
//
// Semantics follow the reference's closures (src/scene/shader.rs) and samplers
// (src/scene/util.rs); the structure does not: the reference runs them from a serial
// polygon loop through a shared mutable `Buffer`, here they are pure functions of a
// triangle record so that thousands of wavefronts can run them independently.
#pragma once

#include "tr_math.h"
#include "tr_pk.h"
#include "tr_powf.h"
#include "tr_types.h"

namespace tr {

// ---------------------------------------------------------------------------------------------
// Vertex stage
// ---------------------------------------------------------------------------------------------

// should_cull_face, shader.rs:116-124 (object space, orthographic approximation)
TR_HD bool cull_face(vec3 p0, vec3 p1, vec3 p2, const float *cam)
{
    vec3 n = cross3(sub3(p1, p0), sub3(p2, p0));
    return dot3(make3(cam[0], cam[1], cam[2]), n) <= 0.0f;
}

// Vector3::from_homogeneous(it_m * n.to_homogeneous()).unwrap().normalize()
// (w stays 0 because it_m's last row is (0,0,0,1) times a w = 0 vector; shader.rs:196-199)
TR_HD vec3 transform_normal(const float *it_m, vec3 n)
{
    vec4 q = mul_m4_v4(it_m, n.x, n.y, n.z, 0.0f);
    return normalize3(make3(q.x, q.y, q.z));
}

// store_vertex_transformation_results, shader.rs:150-165.  Returns false when w == 0
// (the reference's unwrap panics).
TR_HD bool project_vertex(const float *mat, vec3 p, int32_t &rx, int32_t &ry, float &rz)
{
    vec4 q = mul_m4_v4(mat, p.x, p.y, p.z, 1.0f);
    if (q.w == 0.0f) return false;
    float sx = q.x / q.w, sy = q.y / q.w, sz = q.z / q.w;
    rx = f32_to_i32(sx);
    ry = f32_to_i32(sy);
    rz = sz;
    return true;
}

// Runs vertex closure `VS` for polygon `t`.  Returns true when the polygon is kept; fills the
// raster coordinates / z of `r` and the varyings.  `err` collects DevErr bits.
template <int VS>
TR_HD bool vertex_stage(const DevMesh &mesh, const DevUniforms &u, uint32_t t, RasterRec &r,
                        float *vary, uint32_t &err)
{
    const float *row = mesh.tri + (size_t)TRI_FLOATS * t;
    float m[TRI_FLOATS];
#if defined(__HIP_DEVICE_COMPILE__)
    {
        const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll
        for (int i = 0; i < TRI_FLOATS / 4; i++) {
            const float4 q = r4[i];
            m[4 * i] = q.x;
            m[4 * i + 1] = q.y;
            m[4 * i + 2] = q.z;
            m[4 * i + 3] = q.w;
        }
    }
#else
    for (int i = 0; i < TRI_FLOATS; i++) m[i] = row[i];
#endif
    vec3 p0 = make3(m[0], m[1], m[2]);
    vec3 p1 = make3(m[3], m[4], m[5]);
    vec3 p2 = make3(m[6], m[7], m[8]);

    if (VS != VS_DEPTH) {
        if (cull_face(p0, p1, p2, u.camera_direction)) return false;
    }

    vec3 tl = make3(u.t_light[0], u.t_light[1], u.t_light[2]);
    if (VS == VS_DEFAULT) {
        // shader.rs:298-305: face normal, transformed, one diffuse coefficient for the polygon
        vec3 fn = cross3(sub3(p1, p0), sub3(p2, p0));
        float diff = dot3(tl, transform_normal(u.it_m, fn));
        vary[6] = diff;
        vary[7] = diff;
        vary[8] = diff;
    } else if (VS == VS_PHONG) {
        // shader.rs:362-373
        for (int i = 0; i < 3; i++) {
            vec3 n = make3(m[9 + 3 * i], m[10 + 3 * i], m[11 + 3 * i]);
            vary[6 + i] = dot3(tl, transform_normal(u.it_m, n));
        }
    } else if (VS == VS_DARBOUX) {
        // shader.rs:562-584; rows 0 and 1 of the local basis (shader.rs:612-623) depend on the
        // polygon only, so they are formed here once instead of once per fragment.
        vec3 tp[3];
        const vec3 pp[3] = { p0, p1, p2 };
        for (int i = 0; i < 3; i++) {
            vec4 q = mul_m4_v4(u.m, pp[i].x, pp[i].y, pp[i].z, 1.0f);
            if (q.w == 0.0f) {
                err |= DE_W_ZERO;
                return false;
            }
            tp[i] = make3(q.x / q.w, q.y / q.w, q.z / q.w);
        }
        vec3 r0 = normalize3(mul_m3_v3(tp[0], tp[1], tp[2], make3(-1.0f, 1.0f, 0.0f)));
        vec3 r1 = normalize3(mul_m3_v3(tp[0], tp[1], tp[2], make3(-1.0f, 0.0f, 1.0f)));
        vary[6] = r0.x;
        vary[7] = r0.y;
        vary[8] = r0.z;
        vary[9] = r1.x;
        vary[10] = r1.y;
        vary[11] = r1.z;
        for (int i = 0; i < 3; i++) {
            vec3 tn = transform_normal(u.it_m, make3(m[9 + 3 * i], m[10 + 3 * i], m[11 + 3 * i]));
            vary[12 + 3 * i + 0] = tn.x;
            vary[12 + 3 * i + 1] = tn.y;
            vary[12 + 3 * i + 2] = tn.z;
        }
    }

    const float *mat = (VS == VS_DEPTH) ? u.shadow_matrix : u.vpmv;
    bool ok = project_vertex(mat, p0, r.x0, r.y0, r.z0);
    ok = project_vertex(mat, p1, r.x1, r.y1, r.z1) && ok;
    ok = project_vertex(mat, p2, r.x2, r.y2, r.z2) && ok;
    if (!ok) {
        err |= DE_W_ZERO;
        return false;
    }

    // store_vertex_uvs, shader.rs:136-147: (u, 1 - v)
    for (int i = 0; i < 3; i++) {
        vary[2 * i + 0] = m[18 + 2 * i];
        vary[2 * i + 1] = 1.0f - m[19 + 2 * i];
    }
    return true;
}

TR_HD int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
TR_HD int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
// i32 subtraction as a release build of the reference performs it (wrapping)
TR_HD int32_t isub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

// Per-triangle constants of to_barycentric_coord (scene.rs:174-197): the first two components
// of both cross-product operands and cross.z do not depend on the pixel.
struct Edge {
    float a0, a1, b0, b1, cz;
    int32_t x0, y0;
};

TR_HD Edge edge_setup(const RasterRec &r)
{
    Edge e;
    e.a0 = (float)isub(r.x1, r.x0);
    e.a1 = (float)isub(r.x2, r.x0);
    e.b0 = (float)isub(r.y1, r.y0);
    e.b1 = (float)isub(r.y2, r.y0);
    e.cz = e.a0 * e.b1 - e.a1 * e.b0;
    e.x0 = r.x0;
    e.y0 = r.y0;
    return e;
}

// get_triangle_bounding_box + clamp (scene.rs:160-171, 233-239) to the rows this scene owns;
// degenerate polygons (|cross.z| < 1: every pixel gets (-1,1,1), scene.rs:188-191) get an
// empty box because they can never produce a fragment.
TR_HD void finish_raster_rec(RasterRec &r, const DevFrame &f)
{
    int32_t llx = imin(imin(r.x0, r.x1), r.x2), lly = imin(imin(r.y0, r.y1), r.y2);
    int32_t urx = imax(imax(r.x0, r.x1), r.x2), ury = imax(imax(r.y0, r.y1), r.y2);
    r.bx0 = imax(0, llx);
    r.bx1 = imin(urx, (int32_t)(f.width - 1u));
    r.by0 = imax(f.band_y0, lly);
    r.by1 = imin(ury, f.band_y1 - 1);
    Edge e = edge_setup(r);
    if (fabsf(e.cz) < 1.0f || r.by0 > r.by1) {
        r.bx0 = 1;
        r.bx1 = 0;
    }
    r.pad[0] = r.pad[1] = 0u;
}

TR_HD void mark_rejected(RasterRec &r)
{
    r.x0 = r.y0 = r.x1 = r.y1 = r.x2 = r.y2 = 0;
    r.z0 = r.z1 = r.z2 = 0.0f;
    r.bx0 = 1;
    r.bx1 = 0;
    r.by0 = 1;
    r.by1 = 0;
    r.pad[0] = r.pad[1] = 0u;
}

// ---------------------------------------------------------------------------------------------
// Coverage
// ---------------------------------------------------------------------------------------------

// x and y of the raw cross product at pixel (px, py); z is Edge::cz.
TR_HD void edge_cross(const Edge &e, int32_t px, int32_t py, float &cx, float &cy)
{
    float a2 = (float)isub(e.x0, px);
    float b2 = (float)isub(e.y0, py);
    cx = e.a1 * b2 - a2 * e.b1;
    cy = a2 * e.b0 - e.a0 * b2;
}

// The reference's inside test is `!(bar.x < 0 || bar.y < 0 || bar.z < 0)` on
// bar = (1 - (cx+cy)/cz, cx/cz, cy/cz) (scene.rs:192-196, 245).  With |cz| >= 1 and cx, cy
// integer valued (products and differences of i32-valued floats) the three sign tests can be
// decided without dividing:
//   fl(cx/cz) < 0   <=>  cx != 0 and sign(cx) != sign(cz)   (no underflow is possible);
//   fl(1 - fl(s/cz)) < 0  <=>  fl(s/cz) > 1  <=>  |s| > |cz| with sign(s) = sign(cz), because for
//   floats s > cz > 0 the quotient exceeds 1 + 2^-24 and so rounds above 1.
// tests/test_coverage_math.py checks the equivalence against the dividing form exhaustively on
// random and adversarial inputs.  Only covered fragments pay for the three IEEE divisions.
TR_HD bool covers(float cx, float cy, float cz)
{
    float s = cx + cy;
    if (cz > 0.0f) return cx >= 0.0f && cy >= 0.0f && s <= cz;
    return cx <= 0.0f && cy <= 0.0f && s >= cz;
}

// The form the tile kernel evaluates (two pixels at a time, packed): the polygon is first
// orientation-normalised -- a0, a1, b0, b1 negated when cross.z < 0, which negates cross.x and
// cross.y exactly -- so that cz > 0, and `s <= cz` is taken as cz - s >= 0 (a difference of two
// floats has the sign of the exact difference; f32 denormals are on), which lets one three-way
// minimum and one compare decide a pixel.  Checked against `covers` and the dividing form by
// tests/test_coverage_math.py.
TR_HD bool covers_oriented(float cx, float cy, float cz_positive)
{
    return fminf(fminf(cx, cy), cz_positive - (cx + cy)) >= 0.0f;
}

TR_HD vec3 barycentric(float cx, float cy, float cz)
{
    return make3(1.0f - (cx + cy) / cz, cx / cz, cy / cz);
}

// Same values through the shared-reciprocal division (one polygon, many pixels).
TR_HD vec3 barycentric_by(float cx, float cy, Recip rz)
{
    return make3(1.0f - div_by(cx + cy, rz), div_by(cx, rz), div_by(cy, rz));
}

// RN(1 / cross.z) of a polygon record: computed once per polygon by the setup kernel, carried
// in the record's spare word.
TR_HD float record_recip(const RasterRec &r) { return 1.0f / edge_setup(r).cz; }

// Two pixels at once (same polygon in coverage, possibly different ones in shading): the raw
// cross products of scene.rs:178-187 per component.
struct Edge2 {
    f2 a0, a1, b0, b1, cz, y;
};

TR_HD void edge_cross2(const Edge2 &e, f2 a2, f2 b2, f2 &cx, f2 &cy)
{
    cx = e.a1 * b2 - a2 * e.b1;
    cy = a2 * e.b0 - e.a0 * b2;
}

struct Bary2 {
    f2 x, y, z;
};

// Barycentrics whose zeros may carry the wrong sign (see div_by2_unsigned_zero): good for the
// depth comparison of the coverage loop, not for values that are stored.
TR_HD Bary2 barycentric2_for_compare(f2 cx, f2 cy, const Edge2 &e)
{
    // the three quotients of div_by2_unsigned_zero advanced in lockstep: each step's three
    // operations are independent, so no dependent packed operation issues back to back
    const f2 s = cx + cy;
    const f2 q0s = s * e.y, q0x = cx * e.y, q0y = cy * e.y;
    f2 es = fma2(-q0s, e.cz, s), ex = fma2(-q0x, e.cz, cx), ey = fma2(-q0y, e.cz, cy);
    f2 qs = fma2(es, e.y, q0s), qx = fma2(ex, e.y, q0x), qy = fma2(ey, e.y, q0y);
    es = fma2(-qs, e.cz, s);
    ex = fma2(-qx, e.cz, cx);
    ey = fma2(-qy, e.cz, cy);
    Bary2 b;
    b.x = splat2(1.0f) - fma2(es, e.y, qs);
    b.y = fma2(ex, e.y, qx);
    b.z = fma2(ey, e.y, qy);
    return b;
}

TR_HD Bary2 barycentric2(f2 cx, f2 cy, const Edge2 &e)
{
    // div_by2 for the three numerators in lockstep (see barycentric2_for_compare), here with the
    // sign of a zero quotient kept: these values are stored and shaded
    const f2 s = cx + cy;
    const f2 q0s = s * e.y, q0x = cx * e.y, q0y = cy * e.y;
    f2 es = fma2(-q0s, e.cz, s), ex = fma2(-q0x, e.cz, cx), ey = fma2(-q0y, e.cz, cy);
    f2 qs = fma2(es, e.y, q0s), qx = fma2(ex, e.y, q0x), qy = fma2(ey, e.y, q0y);
    es = fma2(-qs, e.cz, s);
    ex = fma2(-qx, e.cz, cx);
    ey = fma2(-qy, e.cz, cy);
    qs = fma2(es, e.y, qs);
    qx = fma2(ex, e.y, qx);
    qy = fma2(ey, e.y, qy);
    Bary2 b;
    b.x = splat2(1.0f) - quotient_sign_from(qs, q0s);
    b.y = quotient_sign_from(qx, q0x);
    b.z = quotient_sign_from(qy, q0y);
    return b;
}

// ---------------------------------------------------------------------------------------------
// Fragment stage
// ---------------------------------------------------------------------------------------------

// util.rs:34-83: coord = ((uv.x * w) as u32, (uv.y * h) as u32); `dims` is the image whose
// width/height scale the uv (get_normal_tangent_at_uv uses normal_map's, util.rs:62-63).
TR_HD uint32_t fetch_texel(const DevTextures &tex, int which, int dims, float u, float v,
                           uint32_t &err)
{
    uint32_t cx = f32_to_u32(u * (float)tex.w[dims]);
    uint32_t cy = f32_to_u32(v * (float)tex.h[dims]);
    if (cx >= tex.w[which] || cy >= tex.h[which]) {
        err |= DE_TEX_OOB;  // the reference's get_pixel panics here
        cx = cx >= tex.w[which] ? tex.w[which] - 1u : cx;
        cy = cy >= tex.h[which] ? tex.h[which] - 1u : cy;
    }
    return tex.texel[which][mul24(cy, tex.w[which]) + cx];  // sides are below 2^16 (checked at create)
}

// util.rs:51-56
TR_HD vec3 decode_normal(uint32_t px)
{
    vec3 n = make3((float)(px & 0xFFu) / 255.0f - 0.5f, (float)((px >> 8) & 0xFFu) / 255.0f - 0.5f,
                   (float)((px >> 16) & 0xFFu) / 255.0f - 0.5f);
    return normalize3(n);
}

TR_HD uint32_t pack_rgb(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16); }

TR_HD uint32_t shade_blend(uint32_t c, float t)
{
    return pack_rgb(blend_black(c & 0xFFu, t), blend_black((c >> 8) & 0xFFu, t),
                    blend_black((c >> 16) & 0xFFu, t));
}

// Point3::from_homogeneous(mat * (x,y,z,1)).unwrap()
TR_HD bool project_point(const float *mat, vec3 p, vec3 &out)
{
    vec4 q = mul_m4_v4(mat, p.x, p.y, p.z, 1.0f);
    if (q.w == 0.0f) return false;
    out = make3(q.x / q.w, q.y / q.w, q.z / q.w);
    return true;
}

// (c.x.round() as u32 + (c.y.round() as u32) * width) as usize with wrapping u32 arithmetic
// (shader.rs:774-775); out-of-range indices panic in the reference and are flagged here.
TR_HD float shadow_fetch(const float *shadow, uint32_t W, uint32_t H, vec3 c, uint32_t &err)
{
    uint32_t ix = f32_to_u32(roundf(c.x));
    uint32_t iy = f32_to_u32(roundf(c.y));
    uint32_t idx = ix + iy * W;
    if (idx >= W * H) {
        err |= DE_SHADOW_OOB;
        return bits_f32(TR_F32_MIN_BITS);
    }
    return shadow[idx];
}

// Runs fragment closure `FS` for a fragment whose depth test has already passed.
// Returns packed rgb (r | g<<8 | b<<16).
// The part of fragment closure `FS` after `uv = vertex_uvs * bar`: texel fetches, lighting, blend.
template <int FS>
TR_HD uint32_t fragment_color(const DevUniforms &u, const DevTextures &tex, const float *vary,
                              vec3 bar, float uu, float vv, uint32_t x, uint32_t y, float z,
                              const float *shadow, uint32_t W, uint32_t H, uint32_t &err)
{
    vec3 tl = make3(u.t_light[0], u.t_light[1], u.t_light[2]);

    if (FS == FS_DEFAULT) {
        uint32_t c = fetch_texel(tex, 0, 0, uu, vv, err);
        return shade_blend(c, vary[6]);
    }
    if (FS == FS_PHONG) {
        uint32_t c = fetch_texel(tex, 0, 0, uu, vv, err);
        float diff = dot3(bar, make3(vary[6], vary[7], vary[8]));
        return shade_blend(c, diff);
    }
    if (FS == FS_NORMAL_MAP) {
        uint32_t c = fetch_texel(tex, 0, 0, uu, vv, err);
        vec3 tn = transform_normal(u.it_m, decode_normal(fetch_texel(tex, 1, 1, uu, vv, err)));
        return shade_blend(c, dot3(tl, tn));
    }
    if (FS == FS_SPECULAR) {
        uint32_t c = fetch_texel(tex, 0, 0, uu, vv, err);
        vec3 tn = transform_normal(u.it_m, decode_normal(fetch_texel(tex, 1, 1, uu, vv, err)));
        // (2.0 * (t_n * t_light.dot(t_n)) - t_light).normalize(), shader.rs:515-518
        vec3 a = scale3(tn, dot3(tl, tn));
        vec3 refl = normalize3(sub3(make3(2.0f * a.x, 2.0f * a.y, 2.0f * a.z), tl));
        float diff = dot3(tl, tn);
        float e = (float)(fetch_texel(tex, 3, 3, uu, vv, err) & 0xFFu);
        float spec = 0.6f * tr_powf(fmaxf(refl.z, 0.0f), e);  // the host libm's powf, bit for bit (tr_powf.h)
        float k = diff + spec;
        return pack_rgb(f32_to_u8(fminf(k * (float)(c & 0xFFu), 255.0f)),
                        f32_to_u8(fminf(k * (float)((c >> 8) & 0xFFu), 255.0f)),
                        f32_to_u8(fminf(k * (float)((c >> 16) & 0xFFu), 255.0f)));
    }
    if (FS == FS_DARBOUX) {
        uint32_t c = fetch_texel(tex, 0, 0, uu, vv, err);
        vec3 nt = decode_normal(fetch_texel(tex, 2, 1, uu, vv, err));
        vec3 n0 = make3(vary[12], vary[13], vary[14]);
        vec3 n1 = make3(vary[15], vary[16], vary[17]);
        vec3 n2 = make3(vary[18], vary[19], vary[20]);
        vec3 local_z = mul_m3_v3(n0, n1, n2, bar);
        vec3 r0 = make3(vary[6], vary[7], vary[8]);
        vec3 r1 = make3(vary[9], vary[10], vary[11]);
        vec3 r2 = normalize3(local_z);
        // 3x3 try_inverse of the matrix with rows r0, r1, r2 (nalgebra's cofactor form)
        float m11 = r0.x, m12 = r0.y, m13 = r0.z;
        float m21 = r1.x, m22 = r1.y, m23 = r1.z;
        float m31 = r2.x, m32 = r2.y, m33 = r2.z;
        float minor_m12_m23 = m22 * m33 - m32 * m23;
        float minor_m11_m23 = m21 * m33 - m31 * m23;
        float minor_m11_m22 = m21 * m32 - m31 * m22;
        float det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
        if (det == 0.0f) {
            err |= DE_SINGULAR;  // shader.rs:631 unwrap
            return 0u;
        }
        vec3 i0 = make3(minor_m12_m23 / det, -minor_m11_m23 / det, minor_m11_m22 / det);  // column 0
        vec3 i1 = make3((m13 * m32 - m33 * m12) / det, (m11 * m33 - m31 * m13) / det,
                        (m12 * m31 - m32 * m11) / det);                                    // column 1
        vec3 i2 = make3((m12 * m23 - m22 * m13) / det, (m13 * m21 - m23 * m11) / det,
                        (m11 * m22 - m21 * m12) / det);                                    // column 2
        vec3 local_x = mul_m3_v3(i0, i1, i2, make3(vary[2] - vary[0], vary[4] - vary[0], 0.0f));
        vec3 local_y = mul_m3_v3(i0, i1, i2, make3(vary[3] - vary[1], vary[5] - vary[1], 0.0f));
        vec3 n = normalize3(mul_m3_v3(normalize3(local_x), normalize3(local_y), normalize3(local_z), nt));
        return shade_blend(c, dot3(tl, n));
    }
    if (FS == FS_SHADOW2) {
        vec3 sc;
        if (!project_point(u.sm_ivpmv, make3((float)x, (float)y, z), sc)) {
            err |= DE_W_ZERO;
            return 0u;
        }
        float sv = shadow_fetch(shadow, W, H, sc, err);
        float coef = 1.0f;
        if (sc.z + 1.0f < sv) coef = 0.3f;
        uint32_t c = fetch_texel(tex, 0, 0, uu, vv, err);
        float diff = dot3(bar, make3(vary[6], vary[7], vary[8]));
        return shade_blend(c, diff * coef);
    }
    if (FS == FS_OCCLUSION2) {
        vec3 fp = make3((float)x, (float)y, z);
        vec3 world, fsc;
        if (!project_point(u.i_vpmv, fp, world) || !project_point(u.sm_ivpmv, fp, fsc)) {
            err |= DE_W_ZERO;
            return 0u;
        }
        float fsv = shadow_fetch(shadow, W, H, fsc, err);
        float occ = 1.0f;
        for (int i = 0; i < 16; i++) {
            vec3 sample = add3(world, make3(u.occl_steps[3 * i], u.occl_steps[3 * i + 1],
                                            u.occl_steps[3 * i + 2]));
            vec3 ssc;
            if (!project_point(u.shadow_matrix, sample, ssc)) {
                err |= DE_W_ZERO;
                return 0u;
            }
            float sv = shadow_fetch(shadow, W, H, ssc, err);
            if (sv - 1.0f > fsv) {
                float strength = fminf((sv - fsv) / 20.0f, 1.0f);
                occ -= (1.0f / 16.0f) * strength;
            }
        }
        // color_blend((255,255,255), (0,0,0), occ)
        uint32_t g = blend_black(255u, occ);
        return pack_rgb(g, g, g);
    }
    return 0u;
}

template <int FS>
TR_HD uint32_t fragment_stage(const DevUniforms &u, const DevTextures &tex, const float *vary,
                              vec3 bar, uint32_t x, uint32_t y, float z, const float *shadow,
                              uint32_t W, uint32_t H, uint32_t &err)
{
    // uv = vertex_uvs * bar (2x3 gemv)
    float uu = vary[0] * bar.x, vv = vary[1] * bar.x;
    uu = vary[2] * bar.y + uu;
    vv = vary[3] * bar.y + vv;
    uu = vary[4] * bar.z + uu;
    vv = vary[5] * bar.z + vv;
    return fragment_color<FS>(u, tex, vary, bar, uu, vv, x, y, z, shadow, W, H, err);
}

}  // namespace tr
