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
#include "tr_texels.h"
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

// The reference's inside test is `!(bar.x < 0 || bar.y < 0 || bar.z < 0)` on
// bar = (1 - (cx+cy)/cz, cx/cz, cy/cz) (scene.rs:192-196, 245), cx / cy / cz the components of the raw
// cross product of scene.rs:178-187.  With |cz| >= 1 and cx, cy integer valued (products and
// differences of i32-valued floats) the three sign tests can be decided without dividing:
//   fl(cx/cz) < 0   <=>  cx != 0 and sign(cx) != sign(cz)   (no underflow is possible);
//   fl(1 - fl(s/cz)) < 0  <=>  fl(s/cz) > 1  <=>  |s| > |cz| with sign(s) = sign(cz), because for
//   floats s > cz > 0 the quotient exceeds 1 + 2^-24 and so rounds above 1.
// The form the tile kernel evaluates (two pixels at a time, packed): the polygon is first
// orientation-normalised -- a0, a1, b0, b1 negated when cross.z < 0, which negates cross.x and
// cross.y exactly -- so that cz > 0, and `s <= cz` is taken as cz - s >= 0 (a difference of two
// floats has the sign of the exact difference; f32 denormals are on), which lets one three-way
// minimum and one compare decide a pixel.  Checked against the dividing form (the oracle's
// to_barycentric_coord + sign tests) by tests/test_coverage_math.py.
TR_HD bool covers_oriented(float cx, float cy, float cz_positive)
{
    return fminf(fminf(cx, cy), cz_positive - (cx + cy)) >= 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Conservative coverage masks of a (polygon, tile) pair
// ---------------------------------------------------------------------------------------------
// The setup kernel stores, with every copy of a record it appends to a tile's bin, which parts of the
// polygon's clamped box INSIDE THAT TILE can hold a fragment at all, so that the tile kernel never
// evaluates the edge functions to find out (it used to classify 32 blocks per polygon and wave).
// cross.x, cross.y and cross.z - (cross.x + cross.y) (scene.rs:178-187, 245-247) are linear in the
// pixel, so their largest value over a cell is the value at the cell's origin plus the extent times the
// positive part of the slope; a cell is dropped only when that maximum misses zero by more than a bound
// (2^-21 of the operand magnitudes) on the f32 rounding of both this estimate and the per-pixel
// evaluation -- a superset of the exact test's pixels is always kept (tests/test_emulation.py counts
// violations on random and far-vertex soups; every GPU parity test runs through it).
//
//   SMALL pair (box inside the tile at most SCAN_MAX_CHUNKS 8-pixel chunks wide, counted from the even
//   pixel at or left of its first column): 64 cells, bit 4 * row + chunk, row 0 = the box's first row
//   inside the tile, chunk c = pixels xs + 8c .. xs + 8c + 7 of that row.
//   LARGE pair -- and every pair of a pass whose tile kernel owns columns (cells = false) -- : bits 0..15 of
//   `lo` = the 8-pixel wide block columns of the tile (both block rows) with a live 8x8 block.
constexpr int SCAN_MAX_CHUNKS = 4;

struct PairBox {
    int32_t ax0, ax1, ay0, ay1;  // the polygon's clamped box cut to the tile (absolute pixels)
    int32_t xs;                  // even pixel at or left of ax0
    int32_t nch;                 // 8-pixel chunks from xs that reach ax1
};

TR_HD PairBox pair_box(int32_t bx0, int32_t bx1, int32_t by0, int32_t by1, int32_t tile_x0, int32_t tile_y0)
{
    PairBox b;
    b.ax0 = imax(bx0, tile_x0);
    b.ax1 = imin(bx1, tile_x0 + TILE_W - 1);
    b.ay0 = imax(by0, tile_y0);
    b.ay1 = imin(by1, tile_y0 + TILE_H - 1);
    b.xs = b.ax0 & ~1;
    b.nch = (isub(b.ax1, b.xs) >> 3) + 1;
    return b;
}

TR_HD float fma_est(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return fmaf(a, b, c);
#endif
}

// (bx0 .. by1: the polygon's clamped box; (x0, y0): its vertex 0; a0, a1, b0, b1: the edge vectors from it as the
// record holds them -- Edge::a0 .. b1, edge_setup)
TR_HD void pair_masks(int32_t bx0, int32_t bx1, int32_t by0, int32_t by1, int32_t x0, int32_t y0, float a0, float a1, float b0,
                      float b1, int32_t tile_x0, int32_t tile_y0, bool cells, uint32_t &lo, uint32_t &hi)
{
    const PairBox pb = pair_box(bx0, bx1, by0, by1, tile_x0, tile_y0);
    lo = hi = 0u;
    if (pb.ax0 > pb.ax1 || pb.ay0 > pb.ay1) return;
    // orientation-normalised edge constants (cross.z > 0), as the tile kernel uses them
    float cz = a0 * b1 - a1 * b0;
    if (cz < 0.0f) {
        a0 = -a0; a1 = -a1; b0 = -b0; b1 = -b1;
        cz = -cz;
    }
    const bool small = cells && pb.nch <= SCAN_MAX_CHUNKS;
    const int32_t org_x = small ? pb.xs : tile_x0, org_y = small ? pb.ay0 : tile_y0;
    const float ox = (float)isub(x0, org_x), oy = (float)isub(y0, org_y);
    const float e0x = b1, e0y = -a1, e1x = -b0, e1y = a0, e2x = b0 - b1, e2y = a1 - a0;
    const float e0 = a1 * oy - ox * b1, e1 = ox * b0 - a0 * oy;
    const float e2 = cz - (e0 + e1);
    const float margin = 4.76837158e-7f /* 2^-21 */ *
                         (((fabsf(a1) + fabsf(a0)) * (fabsf(oy) + (float)TILE_H) +
                           (fabsf(b1) + fabsf(b0)) * (fabsf(ox) + (small ? 8.0f * SCAN_MAX_CHUNKS : (float)TILE_W))) + cz);
    const float px0 = fmaxf(e0x, 0.0f), px1 = fmaxf(e1x, 0.0f), px2 = fmaxf(e2x, 0.0f);
    // (the loops run over the box's own rows / block columns only: on the device they stay rolled -- eight copies of
    // a fully unrolled form in k_setup's pair loop were 45 KB of code and most of that kernel's time)
    if (small) {
        const float m0 = e0 + (7.0f * px0 + margin), m1 = e1 + (7.0f * px1 + margin), m2 = e2 + (7.0f * px2 + 2.0f * margin);
        const int32_t rows = pb.ay1 - pb.ay0 + 1;
        const uint32_t in_box = (1u << pb.nch) - 1u;
        unsigned long long cells_mask = 0ull;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int j = 0; j < rows; j++) {
            const float n0 = fma_est((float)j, e0y, m0), n1 = fma_est((float)j, e1y, m1), n2 = fma_est((float)j, e2y, m2);
            uint32_t bits = 0u;
            for (int c = 0; c < SCAN_MAX_CHUNKS; c++) {
                const float worst = fminf(fminf(fma_est(8.0f * c, e0x, n0), fma_est(8.0f * c, e1x, n1)), fma_est(8.0f * c, e2x, n2));
                bits |= worst >= 0.0f ? 1u << c : 0u;
            }
            cells_mask |= (unsigned long long)(bits & in_box) << (4 * j);
        }
        lo = (uint32_t)cells_mask;
        hi = (uint32_t)(cells_mask >> 32);
    } else {
        const float py0 = fmaxf(e0y, 0.0f), py1 = fmaxf(e1y, 0.0f), py2 = fmaxf(e2y, 0.0f);
        const float m0 = e0 + (7.0f * (px0 + py0) + margin), m1 = e1 + (7.0f * (px1 + py1) + margin);
        const float m2 = e2 + (7.0f * (px2 + py2) + 2.0f * margin);
        // block columns / rows the box meets
        const int32_t ia = isub(pb.ax0, tile_x0) >> 3, ib = isub(pb.ax1, tile_x0) >> 3;
        const int32_t ja = isub(pb.ay0, tile_y0) >> 3, jb = isub(pb.ay1, tile_y0) >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int j = ja; j <= jb; j++) {
            const float n0 = fma_est(8.0f * (float)j, e0y, m0), n1 = fma_est(8.0f * (float)j, e1y, m1), n2 = fma_est(8.0f * (float)j, e2y, m2);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 2
#endif
            for (int i = ia; i <= ib; i++) {
                const float x = 8.0f * (float)i;
                const float worst = fminf(fminf(fma_est(x, e0x, n0), fma_est(x, e1x, n1)), fma_est(x, e2x, n2));
                lo |= worst >= 0.0f ? 1u << i : 0u;
            }
        }
    }
}

TR_HD void pair_masks(const RasterRec &r, int32_t tile_x0, int32_t tile_y0, bool cells, uint32_t &lo, uint32_t &hi)
{
    const Edge e = edge_setup(r);
    pair_masks(r.bx0, r.bx1, r.by0, r.by1, r.x0, r.y0, e.a0, e.a1, e.b0, e.b1, tile_x0, tile_y0, cells, lo, hi);
}

// Block columns of the tile (8 pixels wide, bit i) that can hold a fragment of a pair, from its masks:
// a large pair's `lo` as it is; a small pair's from the chunk columns of its cells (a chunk starts at an
// even pixel and may straddle two block columns).
TR_HD uint32_t pair_block_columns(uint32_t lo, uint32_t hi, bool cells, const PairBox &pb, int32_t tile_x0)
{
    if (!cells || pb.nch > SCAN_MAX_CHUNKS) return lo & 0xFFFFu;
    uint32_t c = lo | hi;
    c |= c >> 16;
    c |= c >> 8;
    c = (c | (c >> 4)) & 0xFu;  // chunk columns with a live cell
    const uint32_t first = (uint32_t)isub(pb.xs, tile_x0);
    const uint32_t cols = c << (first >> 3);
    return (cols | ((first & 7u) ? cols << 1 : 0u)) & 0xFFFFu;
}

// RN(1 / cross.z) of a polygon record: computed once per polygon by the setup kernel, carried
// in the record's spare word.
TR_HD float record_recip(const RasterRec &r) { return 1.0f / edge_setup(r).cz; }

// Two pixels at once (same polygon in coverage, possibly different ones in shading): the raw
// cross products of scene.rs:178-187 per component.
template <class V>
struct Edge2T {
    V a0, a1, b0, b1, cz, y;
};
using Edge2 = Edge2T<f2>;

template <class V>
TR_HD void edge_cross2(const Edge2T<V> &e, V a2, V b2, V &cx, V &cy)
{
    cx = e.a1 * b2 - a2 * e.b1;
    cy = a2 * e.b0 - e.a0 * b2;
}

template <class V>
struct Bary2T {
    V x, y, z;
};
using Bary2 = Bary2T<f2>;

// Barycentrics whose zeros may carry the wrong sign (the residual corrections turn a -0 / d into +0
// and the repair is skipped): good for the depth comparison of the coverage loop, where +0 and -0
// are the same value, not for values that are stored.
template <class V>
TR_HD Bary2T<V> barycentric2_for_compare(V cx, V cy, const Edge2T<V> &e)
{
    // the three shared-reciprocal quotients (div_by, tr_math.h) advanced in lockstep: each step's three
    // operations are independent, so no dependent packed operation issues back to back
    const V s = cx + cy;
    const V q0s = s * e.y, q0x = cx * e.y, q0y = cy * e.y;
    V es = fma2(-q0s, e.cz, s), ex = fma2(-q0x, e.cz, cx), ey = fma2(-q0y, e.cz, cy);
    V qs = fma2(es, e.y, q0s), qx = fma2(ex, e.y, q0x), qy = fma2(ey, e.y, q0y);
    es = fma2(-qs, e.cz, s);
    ex = fma2(-qx, e.cz, cx);
    ey = fma2(-qy, e.cz, cy);
    Bary2T<V> b;
    b.x = splat2v<V>(1.0f) - fma2(es, e.y, qs);
    b.y = fma2(ex, e.y, qx);
    b.z = fma2(ey, e.y, qy);
    return b;
}

template <class V>
TR_HD Bary2T<V> barycentric2(V cx, V cy, const Edge2T<V> &e)
{
    // div_by2 for the three numerators in lockstep (see barycentric2_for_compare), here with the
    // sign of a zero quotient kept: these values are stored and shaded
    const V s = cx + cy;
    const V q0s = s * e.y, q0x = cx * e.y, q0y = cy * e.y;
    V es = fma2(-q0s, e.cz, s), ex = fma2(-q0x, e.cz, cx), ey = fma2(-q0y, e.cz, cy);
    V qs = fma2(es, e.y, q0s), qx = fma2(ex, e.y, q0x), qy = fma2(ey, e.y, q0y);
    es = fma2(-qs, e.cz, s);
    ex = fma2(-qx, e.cz, cx);
    ey = fma2(-qy, e.cz, cy);
    qs = fma2(es, e.y, qs);
    qx = fma2(ex, e.y, qx);
    qy = fma2(ey, e.y, qy);
    Bary2T<V> b;
    b.x = splat2v<V>(1.0f) - quotient_sign_from(qs, q0s);
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
    return gload(tex.texel[which] + (mul24(cy, tex.w[which]) + cx));  // sides are below 2^16 (checked at create)
}

// util.rs:51-56
TR_HD vec3 decode_normal(uint32_t px)
{
    vec3 n = make3((float)(px & 0xFFu) / 255.0f - 0.5f, (float)((px >> 8) & 0xFFu) / 255.0f - 0.5f,
                   (float)((px >> 16) & 0xFFu) / 255.0f - 0.5f);
    return normalize3(n);
}

// The texels one fragment needs, in ONE fetch -- and the normal already decoded.  A closure reads up to three images at
// the same (u, v); when the images have the same size -- the reference's assets are all 1024 x 1024 -- the coordinate
// arithmetic is the same three times over and the three loads go to three arrays.  The scene therefore keeps, beside the
// plain images, a texel set for ITS closure (tr_texels.h): per texel the colour image's texel and, for the closures with a
// normal map, the decoded normal as three floats (with the specular exponent riding in the colour word's spare byte),
// tiled into blocks of 128 bytes (8x4 or 4x2 texels): one coordinate computation and one 4- or 16-byte load per fragment,
// a wave's fragments -- a patch of the screen, hence a patch of the image whatever the orientation of the model's uv
// chart -- touch a third to a half of the cache lines that rows of an image would, and decode_normal's six divisions and
// its square root are paid once per texel instead of once per fragment.  Values are the images' own and the decode is
// the closures' own function (IEEE division and square root on the host, the device forms are proven equal to them, §3
// of DESIGN.md): nothing about the result changes.
// t0: `texture`; t1: the raw texel of normal_map (normal-map, specular closures) or normal_map_tangent (darboux:
// util.rs:62-63 scales its coordinates by normal_map's size) when the set is absent; t2: specular_map's exponent byte.
// Returns true when `n` holds the decoded normal (the set), false when the caller decodes t1 itself.
template <int FS>
TR_HD bool fetch_texels(const DevTextures &tex, float u, float v, uint32_t &err, uint32_t &t0, uint32_t &t1, uint32_t &t2, vec3 &n)
{
    constexpr int K = packed_words(FS);
    t1 = t2 = 0u;
    if (tex.packed) {
        // (all four images have the size of image 0: the coordinates and the range check are those of every fetch)
        uint32_t cx = f32_to_u32(u * (float)tex.w[0]);
        uint32_t cy = f32_to_u32(v * (float)tex.h[0]);
        if (cx >= tex.w[0] || cy >= tex.h[0]) {
            err |= DE_TEX_OOB;  // the reference's get_pixel panics here
            cx = cx >= tex.w[0] ? tex.w[0] - 1u : cx;
            cy = cy >= tex.h[0] ? tex.h[0] - 1u : cy;
        }
        const uint32_t at = packed_index(K, tex.packed_bpr, cx, cy);
        if (K == 1) {
            t0 = gload(tex.packed + at);
            return false;
        }
        const Texel4 q = gload(reinterpret_cast<const Texel4 *>(tex.packed) + at);
        t0 = q.x & 0xFFFFFFu;
        t2 = q.x >> 24;
        n = make3(bits_f32(q.y), bits_f32(q.z), bits_f32(q.w));
        return true;
    }
    t0 = fetch_texel(tex, 0, 0, u, v, err);
    if (FS == FS_NORMAL_MAP || FS == FS_SPECULAR) t1 = fetch_texel(tex, 1, 1, u, v, err);
    if (FS == FS_DARBOUX) t1 = fetch_texel(tex, 2, 1, u, v, err);
    if (FS == FS_SPECULAR) t2 = fetch_texel(tex, 3, 3, u, v, err) & 0xFFu;
    return false;
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
// `sclean` (may be null): the shadow buffer's fast-clear flags, one per 128x16 tile of the whole frame --
// non-zero = every value of the tile is f32::MIN and its memory is stale (the depth pass did not write the
// tiles it had no polygons for).  The flag of the tile the flat index falls into is consulted first.
TR_HD float shadow_fetch(const float *shadow, const uint32_t *sclean, uint32_t W, uint32_t H, vec3 c, uint32_t &err)
{
    uint32_t ix = f32_to_u32(roundf(c.x));
    uint32_t iy = f32_to_u32(roundf(c.y));
    uint32_t idx = ix + iy * W;
    if (idx >= W * H) {
        err |= DE_SHADOW_OOB;
        return bits_f32(TR_F32_MIN_BITS);
    }
    if (sclean) {
        // the flat index is what counts upstream (wrapping u32 arithmetic): a column beyond the row lands in
        // a later row, and a huge row can wrap around 2^32 into the buffer (y = 2^23 at W = 512) -- then the
        // tile is that of the pixel the index names
        if (ix >= W || iy >= H) {
            iy = idx / W;
            ix = idx - iy * W;
        }
        if (gload(sclean + ((iy / (uint32_t)TILE_H) * ((W + (uint32_t)TILE_W - 1u) / (uint32_t)TILE_W) + ix / (uint32_t)TILE_W)) != 0u)
            return bits_f32(TR_F32_MIN_BITS);
    }
    return gload(shadow + idx);
}

// Runs fragment closure `FS` for a fragment whose depth test has already passed.
// The normal-map and specular closures after their texel fetches (shader.rs:439-459, 498-534): colour texel `c`, the
// normal decoded from the normal map's texel, the specular map's exponent byte.  Nothing else enters them but the
// frame's constants: the colour is a function of the TEXEL -- which is what lets k_lit run them once per texel and
// frame instead of once per fragment.
template <int FS>
TR_HD uint32_t shade_texel(const DevUniforms &u, uint32_t c, vec3 n, uint32_t exponent)
{
    const vec3 tl = make3(u.t_light[0], u.t_light[1], u.t_light[2]);
    const vec3 tn = transform_normal(u.it_m, n);
    if (FS == FS_NORMAL_MAP) return shade_blend(c, dot3(tl, tn));
    // (2.0 * (t_n * t_light.dot(t_n)) - t_light).normalize(), shader.rs:515-518
    vec3 a = scale3(tn, dot3(tl, tn));
    vec3 refl = normalize3(sub3(make3(2.0f * a.x, 2.0f * a.y, 2.0f * a.z), tl));
    float diff = dot3(tl, tn);
    float e = (float)(exponent & 0xFFu);
    float spec = 0.6f * tr_powf(fmaxf(refl.z, 0.0f), e);  // the host libm's powf, bit for bit (tr_powf.h)
    float k = diff + spec;
    return pack_rgb(f32_to_u8(fminf(k * (float)(c & 0xFFu), 255.0f)),
                    f32_to_u8(fminf(k * (float)((c >> 8) & 0xFFu), 255.0f)),
                    f32_to_u8(fminf(k * (float)((c >> 16) & 0xFFu), 255.0f)));
}

// Returns packed rgb (r | g<<8 | b<<16).
// The part of fragment closure `FS` after `uv = vertex_uvs * bar`: texel fetches, lighting, blend.
template <int FS>
TR_HD uint32_t fragment_color(const DevUniforms &u, const DevTextures &tex, const float *vary,
                              vec3 bar, float uu, float vv, uint32_t x, uint32_t y, float z,
                              const float *shadow, uint32_t W, uint32_t H, uint32_t &err,
                              const uint32_t *sclean = nullptr)
{
    vec3 tl = make3(u.t_light[0], u.t_light[1], u.t_light[2]);

    uint32_t c, t1, t2;  // the closure's texels (fetch_texels)
    vec3 nd = make3(0.0f, 0.0f, 0.0f);
    if (FS == FS_DEFAULT) {
        fetch_texels<FS>(tex, uu, vv, err, c, t1, t2, nd);
        return shade_blend(c, vary[6]);
    }
    if (FS == FS_PHONG) {
        fetch_texels<FS>(tex, uu, vv, err, c, t1, t2, nd);
        float diff = dot3(bar, make3(vary[6], vary[7], vary[8]));
        return shade_blend(c, diff);
    }
    if (FS == FS_NORMAL_MAP) {
        const bool decoded = fetch_texels<FS>(tex, uu, vv, err, c, t1, t2, nd);
        return shade_texel<FS>(u, c, decoded ? nd : decode_normal(t1), t2);
    }
    if (FS == FS_LIT) {  // the closure has run for this texel already (k_lit)
        fetch_texels<FS>(tex, uu, vv, err, c, t1, t2, nd);
        return c;
    }
    if (FS == FS_SPECULAR) {
        const bool decoded = fetch_texels<FS>(tex, uu, vv, err, c, t1, t2, nd);
        return shade_texel<FS>(u, c, decoded ? nd : decode_normal(t1), t2);
    }
    if (FS == FS_DARBOUX) {
        const bool decoded = fetch_texels<FS>(tex, uu, vv, err, c, t1, t2, nd);
        vec3 nt = decoded ? nd : decode_normal(t1);
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
        float sv = shadow_fetch(shadow, sclean, W, H, sc, err);
        float coef = 1.0f;
        if (sc.z + 1.0f < sv) coef = 0.3f;
        fetch_texels<FS>(tex, uu, vv, err, c, t1, t2, nd);
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
        float fsv = shadow_fetch(shadow, sclean, W, H, fsc, err);
        float occ = 1.0f;
        for (int i = 0; i < 16; i++) {
            vec3 sample = add3(world, make3(u.occl_steps[3 * i], u.occl_steps[3 * i + 1],
                                            u.occl_steps[3 * i + 2]));
            vec3 ssc;
            if (!project_point(u.shadow_matrix, sample, ssc)) {
                err |= DE_W_ZERO;
                return 0u;
            }
            float sv = shadow_fetch(shadow, sclean, W, H, ssc, err);
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

// ---------------------------------------------------------------------------------------------
// Fragment stage, two pixels at a time
// ---------------------------------------------------------------------------------------------
// The closures that normalise vectors and invert a basis per fragment (normal_map, specular,
// darboux: shader.rs:439-459, 498-534, 597-655) spend most of their instructions in IEEE divisions
// and square roots: darboux has 27 of the former and 5 of the latter per fragment, each an 11-12
// instruction expansion.  The tile kernel shades two pixels per lane; here both go through the
// closure together in packed arithmetic (one v_pk_* instruction = the same IEEE operation on both
// pixels, tr_pk.h), the divisions of one normalisation or one inverse share ONE reciprocal
// (div_by2_nonzero: Markstein's correction, bit-equal to '/'), and reciprocal and square root come
// from the hardware estimates plus fused corrections (rcp2 / sqrt2, exhaustively equal to '/' and
// sqrtf on the guarded range).  Every operation keeps the reference's order and rounding; the
// results are bit-identical BY CONSTRUCTION inside the guarded operand range, and outside it --
// exact zeros among the numerators, values below 2^-40 or above 2^40, NaN -- the caller runs the
// plain closure (fragment_color) instead, so the fast path never has to be right there.
struct vec3p {
    f2 x, y, z;
};
TR_HD vec3p make3p(f2 x, f2 y, f2 z)
{
    vec3p r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
TR_HD vec3p splat3p(vec3 v) { return make3p(splat2(v.x), splat2(v.y), splat2(v.z)); }
TR_HD f2 dot3p(vec3p a, vec3p b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// mul_m3_v3 (tr_math.h) for two pixels: column-accumulating gemv
TR_HD vec3p mul_m3_v3p(vec3p c0, vec3p c1, vec3p c2, vec3p v)
{
    vec3p r = make3p(c0.x * v.x, c0.y * v.x, c0.z * v.x);
    r = make3p(c1.x * v.y + r.x, c1.y * v.y + r.y, c1.z * v.y + r.z);
    r = make3p(c2.x * v.z + r.x, c2.y * v.z + r.y, c2.z * v.z + r.z);
    return r;
}

// Smallest and largest magnitude among the operands of the fast divisions and square roots of one
// closure invocation, per pixel.  fminf / fmaxf skip NaN: the closures test their result for it.
constexpr float PAIR_GUARD_LO = 9.094947017729282e-13f;  // 2^-40
constexpr float PAIR_GUARD_HI = 1099511627776.0f;        // 2^40
struct PairGuard {
    f2 lo, hi;
};
TR_HD PairGuard guard_init()
{
    PairGuard g;
    g.lo = splat2(1.0f);
    g.hi = splat2(1.0f);
    return g;
}
TR_HD void guard_add(PairGuard &g, f2 a)
{
    g.lo = mk2(fminf(g.lo.x, fabsf(a.x)), fminf(g.lo.y, fabsf(a.y)));
    g.hi = mk2(fmaxf(g.hi.x, fabsf(a.x)), fmaxf(g.hi.y, fabsf(a.y)));
}
TR_HD void guard_add3(PairGuard &g, f2 a, f2 b, f2 c)
{
    g.lo = mk2(fminf(fminf(fminf(g.lo.x, fabsf(a.x)), fabsf(b.x)), fabsf(c.x)),
               fminf(fminf(fminf(g.lo.y, fabsf(a.y)), fabsf(b.y)), fabsf(c.y)));
    g.hi = mk2(fmaxf(fmaxf(fmaxf(g.hi.x, fabsf(a.x)), fabsf(b.x)), fabsf(c.x)),
               fmaxf(fmaxf(fmaxf(g.hi.y, fabsf(a.y)), fabsf(b.y)), fabsf(c.y)));
}
// true = some operand of this pixel was outside the range the fast forms are proven on
TR_HD bool guard_bad(const PairGuard &g, int half)
{
    const float lo = half ? g.lo.y : g.lo.x, hi = half ? g.hi.y : g.hi.x;
    return !(lo >= PAIR_GUARD_LO && hi <= PAIR_GUARD_HI);
}

// normalize3 for two pixels, operands already known to lie in the guarded range and to be non-zero
TR_HD vec3p normalize3p_in_range(vec3p a)
{
    const f2 n = sqrt2(dot3p(a, a));
    const f2 y = rcp2(n);
    return make3p(div_by2_nonzero(a.x, n, y), div_by2_nonzero(a.y, n, y), div_by2_nonzero(a.z, n, y));
}
// normalize3 for two pixels; the components enter the guard (components in [2^-40, 2^40] put the sum
// of squares in [2^-80, 2^82] and the norm in [2^-40, 2^41]: inside sqrt2's and rcp2's ranges)
TR_HD vec3p normalize3p(vec3p a, PairGuard &g)
{
    guard_add3(g, a.x, a.y, a.z);
    return normalize3p_in_range(a);
}

// decode_normal (util.rs:51-56) for two texels: channel / 255 - 0.5, normalised.  No guard: the
// channels are integers 0..255 (c / 255 through the constant's reciprocal is checked for all 256,
// tests/test_coverage_math.py), so every component's magnitude is one of (k + 0.5) / 255, k =
// 0..127: never zero, within [0.00196, 0.5].
TR_HD vec3p decode_normal_p(uint32_t pa, uint32_t pb)
{
    const f2 d = splat2(255.0f), y = splat2(1.0f / 255.0f), half = splat2(0.5f);
    const f2 cx = mk2((float)(pa & 0xFFu), (float)(pb & 0xFFu));
    const f2 cy = mk2((float)((pa >> 8) & 0xFFu), (float)((pb >> 8) & 0xFFu));
    const f2 cz = mk2((float)((pa >> 16) & 0xFFu), (float)((pb >> 16) & 0xFFu));
    return normalize3p_in_range(make3p(div_by2_nonzero(cx, d, y) - half, div_by2_nonzero(cy, d, y) - half,
                                       div_by2_nonzero(cz, d, y) - half));
}

// transform_normal for two pixels: xyz of it_m * (n, 0) in mul_m4_v4's accumulation order (the w = 0
// column's products are kept: they can turn a -0 sum into +0), then normalised
TR_HD vec3p transform_normal_p(const float *m, vec3p n, PairGuard &g)
{
    const f2 zero = splat2(0.0f);
    f2 rx = splat2(m[0]) * n.x, ry = splat2(m[1]) * n.x, rz = splat2(m[2]) * n.x;
    rx = splat2(m[4]) * n.y + rx;
    ry = splat2(m[5]) * n.y + ry;
    rz = splat2(m[6]) * n.y + rz;
    rx = splat2(m[8]) * n.z + rx;
    ry = splat2(m[9]) * n.z + ry;
    rz = splat2(m[10]) * n.z + rz;
    rx = splat2(m[12]) * zero + rx;
    ry = splat2(m[13]) * zero + ry;
    rz = splat2(m[14]) * zero + rz;
    return normalize3p(make3p(rx, ry, rz), g);
}

// color_blend(c, 0, t) for one pixel of a pair, channels from a packed texel (shade_blend)
TR_HD void shade_blend_p(uint32_t ta, uint32_t tb, f2 t, uint32_t &ca, uint32_t &cb)
{
#if TR_BLEND_FAST
    const f2 w = mk2(blend_weight(t.x), blend_weight(t.y));  // (blend_black, tr_math.h: the (1 - t) * 0.0 term as a select)
#else
    const f2 w = t, k = (splat2(1.0f) - t) * splat2(0.0f);
#endif
    ca = 0u;
    cb = 0u;
    for (int ch = 0; ch < 3; ch++) {
#if TR_BLEND_FAST
        const f2 v = w * mk2((float)((ta >> (8 * ch)) & 0xFFu), (float)((tb >> (8 * ch)) & 0xFFu));
#else
        const f2 v = w * mk2((float)((ta >> (8 * ch)) & 0xFFu), (float)((tb >> (8 * ch)) & 0xFFu)) + k;
#endif
        ca = pack_u8(v.x, (uint32_t)ch, ca);
        cb = pack_u8(v.y, (uint32_t)ch, cb);
    }
}

// Which closures have a two-pixel form
#ifndef TR_PAIR_CLOSURES
#define TR_PAIR_CLOSURES ((1 << FS_NORMAL_MAP) | (1 << FS_SPECULAR) | (1 << FS_DARBOUX))
#endif
constexpr bool has_pair_closure(int fs)
{
    return (fs == FS_NORMAL_MAP || fs == FS_SPECULAR || fs == FS_DARBOUX) && ((TR_PAIR_CLOSURES >> fs) & 1);
}

// shade_texel for two texels at once in the packed forms (normal-map and specular closures): colour texels ta / tb, the
// decoded normals, the exponent bytes.  Returns the value whose NaN-ness -- with the guard -- tells whether a texel must
// be redone by the plain closure; used by the two-pixel fragment closure and by k_lit.
template <int FS>
TR_HD f2 shade_texel_pair(const DevUniforms &u, uint32_t ta, uint32_t tb, vec3p nrm, uint32_t ea_byte, uint32_t eb_byte,
                          PairGuard &g, uint32_t &ca, uint32_t &cb)
{
    const vec3p tl = splat3p(make3(u.t_light[0], u.t_light[1], u.t_light[2]));
    const vec3p tn = transform_normal_p(u.it_m, nrm, g);
    f2 result;
    if (FS == FS_NORMAL_MAP) {
        result = dot3p(tl, tn);
        shade_blend_p(ta, tb, result, ca, cb);
    } else {
        // (2.0 * (t_n * t_light.dot(t_n)) - t_light).normalize(), shader.rs:515-518
        const f2 d0 = dot3p(tl, tn);
        const f2 two = splat2(2.0f);
        const vec3p refl = normalize3p(make3p(two * (tn.x * d0) - tl.x, two * (tn.y * d0) - tl.y, two * (tn.z * d0) - tl.z), g);
        const f2 diff = dot3p(tl, tn);
        const float e0 = (float)(ea_byte & 0xFFu), e1 = (float)(eb_byte & 0xFFu);
        const f2 spec = splat2(0.6f) * mk2(tr_powf(fmaxf(refl.z.x, 0.0f), e0), tr_powf(fmaxf(refl.z.y, 0.0f), e1));
        result = diff + spec;
        ca = 0u;
        cb = 0u;
        for (int ch = 0; ch < 3; ch++) {
            const f2 v = result * mk2((float)((ta >> (8 * ch)) & 0xFFu), (float)((tb >> (8 * ch)) & 0xFFu));
            ca |= f32_to_u8(fminf(v.x, 255.0f)) << (8 * ch);
            cb |= f32_to_u8(fminf(v.y, 255.0f)) << (8 * ch);
        }
    }
    return result;
}

// The part of fragment closure `FS` after `uv = vertex_uvs * bar` for two pixels (possibly of
// different polygons): `vary(k)` returns varying k of both.  bad_a / bad_b: that pixel left the
// guarded range -- its colour must come from fragment_color<FS> instead (error bits too: the fast
// path reports texture range errors only, the plain closure also a singular basis).
template <int FS, typename VARY>
TR_HD void fragment_color_pair(const DevUniforms &u, const DevTextures &tex, const VARY &vary, vec3p bar, f2 uu,
                               f2 vv, uint32_t &ca, uint32_t &cb, uint32_t &ea, uint32_t &eb, bool &bad_a, bool &bad_b)
{
    const vec3p tl = splat3p(make3(u.t_light[0], u.t_light[1], u.t_light[2]));
    PairGuard g = guard_init();
    uint32_t ta, ta1, ta2, tb, tb1, tb2;  // the closure's texels at both pixels (fetch_texels)
    vec3 na = make3(0.0f, 0.0f, 0.0f), nb = na;
    const bool decoded = fetch_texels<FS>(tex, uu.x, vv.x, ea, ta, ta1, ta2, na);
    fetch_texels<FS>(tex, uu.y, vv.y, eb, tb, tb1, tb2, nb);
    // the normal the closure decodes from its normal map: from the texel set, or decoded here (plain images)
    const vec3p nrm = decoded ? make3p(mk2(na.x, nb.x), mk2(na.y, nb.y), mk2(na.z, nb.z)) : decode_normal_p(ta1, tb1);
    f2 result;  // the value whose NaN-ness decides (every fast operation feeds it)
    if (FS == FS_NORMAL_MAP || FS == FS_SPECULAR) {
        result = shade_texel_pair<FS>(u, ta, tb, nrm, ta2, tb2, g, ca, cb);
    } else {  // FS_DARBOUX
        const vec3p nt = nrm;
        const vec3p local_z = mul_m3_v3p(make3p(vary(12), vary(13), vary(14)), make3p(vary(15), vary(16), vary(17)),
                                         make3p(vary(18), vary(19), vary(20)), bar);
        const vec3p r2 = normalize3p(local_z, g);
        // 3x3 try_inverse of the matrix with rows r0, r1, r2 (nalgebra's cofactor form): SIX quotients by one
        // determinant.  The inverse is only ever multiplied by (du1, du2, 0.0) and (dv1, dv2, 0.0)
        // (shader.rs:632-643), so its third column enters as q * 0.0 added last: a signed zero (q = cofactor / det is
        // finite here: the cofactors of that column are products of the unit rows r0, r1, the determinant is inside
        // the guard), which changes the sum (a + b) only when that sum is itself a zero -- and a zero component of
        // local_x / local_y puts the pixel outside the guard of the normalisation below, i.e. on the plain closure,
        // which divides all nine.  The column's three cofactors, divisions and products are left out
        // (tests/test_coverage_math.py::test_darboux_third_column_only_matters_for_zero_sums).
        const f2 m11 = vary(6), m12 = vary(7), m13 = vary(8);
        const f2 m21 = vary(9), m22 = vary(10), m23 = vary(11);
        const f2 m31 = r2.x, m32 = r2.y, m33 = r2.z;
        const f2 minor_m12_m23 = m22 * m33 - m32 * m23;
        const f2 minor_m11_m23 = m21 * m33 - m31 * m23;
        const f2 minor_m11_m22 = m21 * m32 - m31 * m22;
        const f2 det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
        const f2 c10 = -minor_m11_m23;
        const f2 c01 = m13 * m32 - m33 * m12, c11 = m11 * m33 - m31 * m13, c21 = m12 * m31 - m32 * m11;
        guard_add(g, det);
        guard_add3(g, minor_m12_m23, c10, minor_m11_m22);
        guard_add3(g, c01, c11, c21);
        const f2 yd = rcp2(det);
        const vec3p i0 = make3p(div_by2_nonzero(minor_m12_m23, det, yd), div_by2_nonzero(c10, det, yd),
                                div_by2_nonzero(minor_m11_m22, det, yd));
        const vec3p i1 = make3p(div_by2_nonzero(c01, det, yd), div_by2_nonzero(c11, det, yd), div_by2_nonzero(c21, det, yd));
        const f2 u0 = vary(0), v0 = vary(1);
        const f2 du1 = vary(2) - u0, du2 = vary(4) - u0, dv1 = vary(3) - v0, dv2 = vary(5) - v0;
        const vec3p local_x = make3p(i1.x * du2 + i0.x * du1, i1.y * du2 + i0.y * du1, i1.z * du2 + i0.z * du1);
        const vec3p local_y = make3p(i1.x * dv2 + i0.x * dv1, i1.y * dv2 + i0.y * dv1, i1.z * dv2 + i0.z * dv1);
        // normalize(local_z) is r2 again (the reference recomputes it, shader.rs:644-648)
        const vec3p n = normalize3p(mul_m3_v3p(normalize3p(local_x, g), normalize3p(local_y, g), r2, nt), g);
        result = dot3p(tl, n);
        shade_blend_p(ta, tb, result, ca, cb);
    }
    bad_a = guard_bad(g, 0) || !(result.x == result.x);
    bad_b = guard_bad(g, 1) || !(result.y == result.y);
}


#if !defined(__HIP_DEVICE_COMPILE__)
}  // namespace tr
#include <vector>
namespace tr {
// Host: the texel set of closure `fs` (tr_texels.h) from the four rgba8 images (all w x h); `bpr` receives the blocks
// per row.  The normals are decode_normal's -- the function the closures call when there is no set.
inline std::vector<uint32_t> pack_texels(int fs, const uint32_t *const image[4], uint32_t w, uint32_t h, uint32_t &bpr)
{
    const int K = packed_words(fs), lbw = packed_lbw(K), lbh = packed_lbh(K), nsrc = packed_normal_source(fs);
    bpr = (w + (1u << lbw) - 1u) >> lbw;
    const uint32_t rows = (h + (1u << lbh) - 1u) >> lbh;
    std::vector<uint32_t> packed(((size_t)bpr * rows << (lbw + lbh)) * K, 0u);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const size_t at = (size_t)packed_index(K, bpr, x, y) * K, i = (size_t)y * w + x;
            packed[at] = (image[0][i] & 0xFFFFFFu) | (fs == FS_SPECULAR ? (image[3][i] & 0xFFu) << 24 : 0u);
            if (K == 4) {
                const vec3 n = decode_normal(image[nsrc][i]);
                packed[at + 1] = f32_bits(n.x);
                packed[at + 2] = f32_bits(n.y);
                packed[at + 3] = f32_bits(n.z);
            }
        }
    return packed;
}
#endif
}  // namespace tr
