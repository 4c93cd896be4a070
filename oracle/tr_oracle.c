/*
 * tr_oracle.c -- CPU oracle for the tiny_renderer triangle-fill path.
 *
 * TEST INFRASTRUCTURE ONLY (see tr_oracle.h).  PARITY UNPINNED (see tr_oracle.h).
 *
 * Restates, single-threaded and in the reference's loop order:
 *   src/scene.rs:128-137 (clear), :151-268 (render), :92-125 (readbacks)
 *   src/scene/shader.rs:116-180 (helpers, z test), :183-279 (prepares), :282-963 (7 pipelines)
 *   src/scene/util.rs:7-83 (colour blend, samplers)
 * with nalgebra 0.31.4's f32 operation order (the crate source is not
 * available here, the order is restated from its published implementation).
 *
 * Build with -O2 -ffp-contract=off -fno-fast-math (x86-64 SSE scalar f32 = IEEE binary32,
 * the same arithmetic rustc emits).  See oracle/Makefile.
 */
#include "tr_oracle.h"

/* TRO_VARIANT (default 0 = the normative restatement).  Non-zero values build ALTERNATIVE readings
 * of the nalgebra operations whose order SURVEY.md Appendix A could not verify against the crate's
 * source; scripts/oracle_variants.py renders the BASELINE configs with each and counts the pixels
 * that change, which is the size of the risk "parity unpinned" leaves:
 *   1  dot of 3-vectors associates to the right: a0*b0 + (a1*b1 + a2*b2)
 *   2  4x4 gemv sums pairwise: (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
 *   3  gemv accumulates into a zeroed vector: y = 0 + col0*v0 + ... (differs in signed zeros only)
 *   4  normalize multiplies by the reciprocal norm: v * (1 / n)
 *   5  3x3 try_inverse multiplies the cofactors by 1 / det
 *   6  4x4 try_inverse divides the cofactors by det
 * Never used by tests/, smoke() or bench.py. */
#ifndef TRO_VARIANT
#define TRO_VARIANT 0
#endif

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Rust `as` casts (saturating, NaN -> 0): shader.rs:161, util.rs:9-11,36-37, scene.rs:103
 * ---------------------------------------------------------------------------------------- */
int32_t tro_f32_to_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

uint32_t tro_f32_to_u32(float v)
{
    if (v != v) return 0;
    if (v >= 4294967296.0f) return UINT32_MAX;
    if (v <= 0.0f) return 0;
    return (uint32_t)v;
}

uint8_t tro_f32_to_u8(float v)
{
    if (v != v) return 0;
    if (v >= 255.0f) return 255;
    if (v <= 0.0f) return 0;
    return (uint8_t)v;
}

/* ------------------------------------------------------------------------------------------
 * nalgebra primitives, 3-vectors (Appendix A)
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
/* a.dot(b) = (a0*b0 + a1*b1) + a2*b2 */
#if TRO_VARIANT == 1
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
#else
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
#endif
/* a.cross(b) */
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* v.normalize(): n = sqrt(dot(v,v)); each component v_k / n (true division) */
static inline v3 v3_normalize(v3 a)
{
    float n = sqrtf(v3_dot(a, a));
#if TRO_VARIANT == 4
    float r = 1.0f / n;
    return v3_make(a.x * r, a.y * r, a.z * r);
#else
    return v3_make(a.x / n, a.y / n, a.z / n);
#endif
}

/* 4x4, column-major: element (r,c) = m[c*4 + r].  M*v is nalgebra's column-accumulating
 * gemv: y = col0*v0; y = col1*v1 + y; y = col2*v2 + y; y = col3*v3 + y. */
void tro_mat4_mul_vec4(const float a[16], const float v[4], float out[4])
{
    float y[4];
#if TRO_VARIANT == 2
    for (int i = 0; i < 4; i++) y[i] = (a[i] * v[0] + a[4 + i] * v[1]) + (a[8 + i] * v[2] + a[12 + i] * v[3]);
#elif TRO_VARIANT == 3
    for (int i = 0; i < 4; i++) y[i] = 0.0f;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) y[i] = a[j * 4 + i] * v[j] + y[i];
#else
    for (int i = 0; i < 4; i++) y[i] = a[0 * 4 + i] * v[0];
    for (int j = 1; j < 4; j++)
        for (int i = 0; i < 4; i++) y[i] = a[j * 4 + i] * v[j] + y[i];
#endif
    for (int i = 0; i < 4; i++) out[i] = y[i];
}

/* A*B: column j of the result = A * (column j of B). */
void tro_mat4_mul(const float a[16], const float b[16], float out[16])
{
    float r[16];
    for (int j = 0; j < 4; j++) tro_mat4_mul_vec4(a, &b[j * 4], &r[j * 4]);
    memcpy(out, r, sizeof r);
}

static void mat4_transpose(const float a[16], float out[16])
{
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) r[c * 4 + rr] = a[rr * 4 + c];
    memcpy(out, r, sizeof r);
}

/* 4x4 try_inverse: cofactor expansion on the column-major slice (the MESA gluInvertMatrix
 * form nalgebra uses), det = m0*c00 + m1*c01 + m2*c02 + m3*c03, every entry * (1/det).
 * Returns 0 when det == 0 (reference: unwrap panic). */
int tro_mat4_inverse(const float m[16], float out[16])
{
    float inv[16];

    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15]
           + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15]
           - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15]
           + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14]
            - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];

    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0f) return 0;

    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15]
           - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15]
           + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11]
           - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];

    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15]
           + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15]
           - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11]
           + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];

    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15]
           - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15]
            + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11]
            - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];

    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14]
            + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14]
            - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10]
            + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];

#if TRO_VARIANT == 6
    for (int i = 0; i < 16; i++) out[i] = inv[i] / det;
#else
    float inv_det = 1.0f / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * inv_det;
#endif
    return 1;
}

/* 3x3, column-major: element (r,c) = a[c*3 + r]. */
static inline v3 mat3_mul_v3(const float a[9], v3 v)
{
#if TRO_VARIANT == 3
    float y0 = a[0] * v.x + 0.0f, y1 = a[1] * v.x + 0.0f, y2 = a[2] * v.x + 0.0f;
#else
    float y0 = a[0] * v.x, y1 = a[1] * v.x, y2 = a[2] * v.x;
#endif
    y0 = a[3] * v.y + y0; y1 = a[4] * v.y + y1; y2 = a[5] * v.y + y2;
    y0 = a[6] * v.z + y0; y1 = a[7] * v.z + y1; y2 = a[8] * v.z + y2;
    return v3_make(y0, y1, y2);
}

/* 3x3 try_inverse, nalgebra's cofactor form (Appendix A). mIJ = row I, col J, 1-based. */
int tro_mat3_inverse(const float a[9], float out[9])
{
    float m11 = a[0], m21 = a[1], m31 = a[2];
    float m12 = a[3], m22 = a[4], m32 = a[5];
    float m13 = a[6], m23 = a[7], m33 = a[8];

    float minor_m12_m23 = m22 * m33 - m32 * m23;
    float minor_m11_m23 = m21 * m33 - m31 * m23;
    float minor_m11_m22 = m21 * m32 - m31 * m22;

    float det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
    if (det == 0.0f) return 0;

    float r[9];
    r[0] = minor_m12_m23 / det;                /* (0,0) */
    r[3] = (m13 * m32 - m33 * m12) / det;      /* (0,1) */
    r[6] = (m12 * m23 - m22 * m13) / det;      /* (0,2) */
    r[1] = -minor_m11_m23 / det;               /* (1,0) */
    r[4] = (m11 * m33 - m31 * m13) / det;      /* (1,1) */
    r[7] = (m13 * m21 - m23 * m11) / det;      /* (1,2) */
    r[2] = minor_m11_m22 / det;                /* (2,0) */
    r[5] = (m12 * m31 - m32 * m11) / det;      /* (2,1) */
    r[8] = (m11 * m22 - m21 * m12) / det;      /* (2,2) */
#if TRO_VARIANT == 5
    {
        const float rd = 1.0f / det;
        r[0] = minor_m12_m23 * rd;
        r[3] = (m13 * m32 - m33 * m12) * rd;
        r[6] = (m12 * m23 - m22 * m13) * rd;
        r[1] = -minor_m11_m23 * rd;
        r[4] = (m11 * m33 - m31 * m13) * rd;
        r[7] = (m13 * m21 - m23 * m11) * rd;
        r[2] = minor_m11_m22 * rd;
        r[5] = (m12 * m31 - m32 * m11) * rd;
        r[8] = (m11 * m22 - m21 * m12) * rd;
    }
#endif
    memcpy(out, r, sizeof r);
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * shader.rs:14-38 Buffer
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t width, height;
    float *z_buffer;
    float *shadow_buffer;
    tro_uniforms u;
    float vertex_intensities[3];
    float vertex_t_positions[9]; /* columns = vertices */
    float vertex_t_normals[9];   /* columns = vertices */
    float vertex_uvs[6];         /* 2x3, column i = (u_i, v_i) */
    int32_t vertex_t_raster[6];  /* 2x3, column i = (x_i, y_i) */
    float vertex_z_values[3];
    uint8_t fragment_color[3];
} buffer_t;

typedef struct {
    float *pos, *tex, *nrm;
    uint32_t *idx;
    uint32_t n_pos, n_tex, n_nrm, n_tri;
    uint8_t *img[4];
    uint32_t img_w[4], img_h[4];
} model_t;

struct tro_scene;
typedef int (*prepare_fn)(struct tro_scene *);
typedef int (*vertex_fn)(struct tro_scene *, const uint32_t *tri); /* 1 keep, 0 skip */
typedef int (*fragment_fn)(struct tro_scene *, uint32_t x, uint32_t y, v3 bar);

typedef struct {
    prepare_fn prepare;
    vertex_fn vertex;
    fragment_fn fragment;
} pass_t;

struct tro_scene {
    uint32_t width, height;
    model_t model;
    buffer_t buf;
    pass_t passes[2];
    int n_passes;
    float light_direction[3], look_from[3], look_at[3], up[3];
    uint8_t *frame_buffer;
    uint32_t *winner;
    uint32_t cur_tri;
    int err;
    tro_stats stats[2];
    int cur_pass;
    float occl_steps[48];
    /* screen-band shard: the clamp rectangle of scene.rs:236-239 with its y range cut
       to the rows [band_y0, band_y1] (internal rows, row 0 = bottom); whole frame by default */
    int32_t band_y0, band_y1;
};

/* ------------------------------------------------------------------------------------------
 * util.rs:7-13 color_blend
 * ---------------------------------------------------------------------------------------- */
void tro_color_blend(const uint8_t c1[3], const uint8_t c2[3], float t, uint8_t out[3])
{
    for (int k = 0; k < 3; k++)
        out[k] = tro_f32_to_u8(t * (float)c1[k] + (1.0f - t) * (float)c2[k]);
}

/* util.rs:34-83 samplers: coord = ((uv.x*w) as u32, (uv.y*h) as u32); get_pixel panics OOB */
static int texel(struct tro_scene *s, int which, uint32_t dim_from, float u, float v,
                 const uint8_t **px)
{
    uint32_t cx = tro_f32_to_u32(u * (float)s->model.img_w[dim_from]);
    uint32_t cy = tro_f32_to_u32(v * (float)s->model.img_h[dim_from]);
    if (cx >= s->model.img_w[which] || cy >= s->model.img_h[which]) {
        s->err |= TRO_E_TEX_OOB;
        return 0;
    }
    *px = s->model.img[which] + 3 * ((size_t)cy * s->model.img_w[which] + cx);
    return 1;
}

static inline v3 decode_normal(const uint8_t *px)
{
    /* util.rs:51-56: px as f32 / 255.0 - 0.5 per channel, then normalize */
    v3 n = v3_make((float)px[0] / 255.0f - 0.5f, (float)px[1] / 255.0f - 0.5f,
                   (float)px[2] / 255.0f - 0.5f);
    return v3_normalize(n);
}

/* ------------------------------------------------------------------------------------------
 * shader.rs:183-279 prepares
 * ---------------------------------------------------------------------------------------- */
static int default_prepare_u(tro_uniforms *u, uint32_t width, uint32_t height, v3 light,
                             v3 from, v3 at, v3 up)
{
    int err = 0;
    v3 new_z = v3_normalize(v3_sub(from, at));
    /* up - new_z.dot(&up) * new_z */
    float d = v3_dot(new_z, up);
    v3 new_y = v3_normalize(v3_sub(up, v3_make(d * new_z.x, d * new_z.y, d * new_z.z)));
    v3 new_x = v3_normalize(v3_cross(new_y, new_z));

    float model[16] = { new_x.x, new_y.x, new_z.x, 0.0f,   /* column 0 */
                        new_x.y, new_y.y, new_z.y, 0.0f,   /* column 1 */
                        new_x.z, new_y.z, new_z.z, 0.0f,   /* column 2 */
                        0.0f,    0.0f,    0.0f,    1.0f }; /* column 3 */
    float view[16] = { 1.0f, 0.0f, 0.0f, 0.0f,
                       0.0f, 1.0f, 0.0f, 0.0f,
                       0.0f, 0.0f, 1.0f, 0.0f,
                       -from.x, -from.y, -from.z, 1.0f };
    float coef = -1.0f / 5.0f;
    float proj[16] = { 1.0f, 0.0f, 0.0f, 0.0f,
                       0.0f, 1.0f, 0.0f, 0.0f,
                       0.0f, 0.0f, 1.0f, coef,
                       0.0f, 0.0f, 0.0f, 1.0f };
    float w = (float)(width - 1u);
    float h = (float)(height - 1u);
    float dd = 255.0f;
    float viewport[16] = { w / 2.0f, 0.0f,     0.0f,      0.0f,
                           0.0f,     h / 2.0f, 0.0f,      0.0f,
                           0.0f,     0.0f,     dd / 2.0f, 0.0f,
                           w / 2.0f, h / 2.0f, dd / 2.0f, 1.0f };

    /* viewport * projection * model * view, left to right (shader.rs:221) */
    float t1[16], t2[16];
    tro_mat4_mul(viewport, proj, t1);
    tro_mat4_mul(t1, model, t2);
    tro_mat4_mul(t2, view, u->vpmv);

    memcpy(u->m, model, sizeof model);
    float mt[16];
    mat4_transpose(model, mt);
    if (!tro_mat4_inverse(mt, u->it_m)) err |= TRO_E_SINGULAR;
    u->camera_direction[0] = new_z.x;
    u->camera_direction[1] = new_z.y;
    u->camera_direction[2] = new_z.z;

    /* Vector3::from_homogeneous(m * light.to_homogeneous()).unwrap().normalize() */
    float lh[4] = { light.x, light.y, light.z, 0.0f }, tl[4];
    tro_mat4_mul_vec4(u->m, lh, tl);
    if (tl[3] != 0.0f) err |= TRO_E_VEC_W_NONZERO;
    v3 tln = v3_normalize(v3_make(tl[0], tl[1], tl[2]));
    u->t_light_direction[0] = tln.x;
    u->t_light_direction[1] = tln.y;
    u->t_light_direction[2] = tln.z;
    return err;
}

int tro_prepare(int kind, tro_uniforms *u, uint32_t width, uint32_t height, const float light[3],
                const float from[3], const float at[3], const float up[3])
{
    v3 l = v3_make(light[0], light[1], light[2]);
    v3 f = v3_make(from[0], from[1], from[2]);
    v3 a = v3_make(at[0], at[1], at[2]);
    v3 p = v3_make(up[0], up[1], up[2]);
    int err;
    switch (kind) {
    case 0:
        return default_prepare_u(u, width, height, l, f, a, p);
    case 1: /* shader.rs:234-255: camera placed at the light; shadow_matrix := vpmv */
        err = default_prepare_u(u, width, height, l, l, a, p);
        memcpy(u->shadow_matrix, u->vpmv, sizeof u->vpmv);
        return err;
    case 2: /* shader.rs:259-279 */
        err = default_prepare_u(u, width, height, l, f, a, p);
        if (!tro_mat4_inverse(u->vpmv, u->i_vpmv)) err |= TRO_E_SINGULAR;
        if (!tro_mat4_inverse(u->m, u->i_m)) err |= TRO_E_SINGULAR;
        return err;
    default:
        return TRO_E_UNKNOWN_PIPELINE;
    }
}

static int prepare_kind(struct tro_scene *s, int kind)
{
    return tro_prepare(kind, &s->buf.u, s->width, s->height, s->light_direction, s->look_from,
                       s->look_at, s->up);
}
static int prep_default(struct tro_scene *s) { return prepare_kind(s, 0); }
static int prep_shadow_1(struct tro_scene *s) { return prepare_kind(s, 1); }
static int prep_shadow_2(struct tro_scene *s) { return prepare_kind(s, 2); }

/* shader.rs:916-929: the 16 step vectors are frame constants; computed once per pass-2 prepare
 * exactly as the fragment shader would for every fragment. */
int tro_occlusion_steps(const float light_direction[3], float out[48])
{
    v3 a = v3_make(0.0f, 0.0f, 1.0f);
    v3 b = v3_make(light_direction[0], light_direction[1], light_direction[2]);
    float rot[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }; /* identity, column-major */

    /* Rotation3::rotation_between = scaled_rotation_between(a, b, 1.0) */
    float na_n = sqrtf(v3_dot(a, a));
    float nb_n = sqrtf(v3_dot(b, b));
    if (na_n > 0.0f && nb_n > 0.0f) { /* try_normalize(0): None when norm <= 0 */
        v3 na = v3_make(a.x / na_n, a.y / na_n, a.z / na_n);
        v3 nb = v3_make(b.x / nb_n, b.y / nb_n, b.z / nb_n);
        v3 c = v3_cross(na, nb);
        float sq = v3_dot(c, c);
        float eps = 1.1920929e-7f; /* f32::EPSILON */
        if (sq > eps * eps) { /* Unit::try_new(c, eps) */
            float cn = sqrtf(sq);
            v3 ax = v3_make(c.x / cn, c.y / cn, c.z / cn);
            float angle = acosf(v3_dot(na, nb)) * 1.0f;
            /* Rotation3::from_axis_angle (identity when angle == 0) */
            if (angle == 0.0f) goto steps;
            float ux = ax.x, uy = ax.y, uz = ax.z;
            float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
            float sn = sinf(angle), cs = cosf(angle);
            float one_m_cos = 1.0f - cs;
            /* row-major constructor arguments -> column-major storage */
            float r00 = sqx + (1.0f - sqx) * cs;
            float r01 = ux * uy * one_m_cos - uz * sn;
            float r02 = ux * uz * one_m_cos + uy * sn;
            float r10 = ux * uy * one_m_cos + uz * sn;
            float r11 = sqy + (1.0f - sqy) * cs;
            float r12 = uy * uz * one_m_cos - ux * sn;
            float r20 = ux * uz * one_m_cos - uy * sn;
            float r21 = uy * uz * one_m_cos + ux * sn;
            float r22 = sqz + (1.0f - sqz) * cs;
            rot[0] = r00; rot[1] = r10; rot[2] = r20;
            rot[3] = r01; rot[4] = r11; rot[5] = r21;
            rot[6] = r02; rot[7] = r12; rot[8] = r22;
        } else if (v3_dot(na, nb) < 0.0f) {
            return TRO_E_ROTATION; /* antiparallel: None -> unwrap panics (shader.rs:921) */
        }
    }
steps:;
    float step_size = 0.02f;
    float angle_coef = (2.0f * 3.14159265358979323846f) / 16.0f;
    for (int i = 0; i < 16; i++) {
        v3 g = v3_make(sinf(angle_coef * (float)i), 0.0f, cosf(angle_coef * (float)i));
        v3 sd = mat3_mul_v3(rot, g);
        v3 st = v3_scale(sd, step_size);
        out[3 * i + 0] = st.x;
        out[3 * i + 1] = st.y;
        out[3 * i + 2] = st.z;
    }
    return 0;
}

static int prep_occlusion_2(struct tro_scene *s)
{
    int err = prepare_kind(s, 2);
    /* shader.rs:882-885: light_direction = Vector3::from_homogeneous(i_m * (t_light, 0)) */
    const tro_uniforms *u = &s->buf.u;
    float tl[4] = { u->t_light_direction[0], u->t_light_direction[1], u->t_light_direction[2], 0.0f };
    float ld[4];
    tro_mat4_mul_vec4(u->i_m, tl, ld);
    if (ld[3] != 0.0f) err |= TRO_E_VEC_W_NONZERO;
    err |= tro_occlusion_steps(ld, s->occl_steps);
    return err;
}

/* ------------------------------------------------------------------------------------------
 * shader.rs:116-180 vertex-stage helpers and the depth test
 * ---------------------------------------------------------------------------------------- */
static int get_vertex_positions(struct tro_scene *s, const uint32_t *tri, v3 p[3])
{
    for (int i = 0; i < 3; i++) {
        uint32_t k = tri[3 * i + 0];
        if (k >= s->model.n_pos) { s->err |= TRO_E_INDEX_OOB; return 0; }
        p[i] = v3_make(s->model.pos[3 * k], s->model.pos[3 * k + 1], s->model.pos[3 * k + 2]);
    }
    return 1;
}

static int vertex_normal(struct tro_scene *s, const uint32_t *tri, int i, v3 *n)
{
    uint32_t k = tri[3 * i + 2];
    if (k >= s->model.n_nrm) { s->err |= TRO_E_INDEX_OOB; return 0; }
    *n = v3_make(s->model.nrm[3 * k], s->model.nrm[3 * k + 1], s->model.nrm[3 * k + 2]);
    return 1;
}

static int should_cull_face(const v3 p[3], const float cam[3])
{
    v3 face_normal = v3_cross(v3_sub(p[1], p[0]), v3_sub(p[2], p[0]));
    return v3_dot(v3_make(cam[0], cam[1], cam[2]), face_normal) <= 0.0f;
}

static int store_vertex_uvs(struct tro_scene *s, const uint32_t *tri)
{
    for (int i = 0; i < 3; i++) {
        uint32_t k = tri[3 * i + 1];
        if (k >= s->model.n_tex) { s->err |= TRO_E_INDEX_OOB; return 0; }
        s->buf.vertex_uvs[2 * i + 0] = s->model.tex[3 * k];
        s->buf.vertex_uvs[2 * i + 1] = 1.0f - s->model.tex[3 * k + 1];
    }
    return 1;
}

static int store_vertex_transformation_results(struct tro_scene *s, const v3 p[3],
                                               const float mat[16])
{
    for (int i = 0; i < 3; i++) {
        float ph[4] = { p[i].x, p[i].y, p[i].z, 1.0f }, q[4];
        tro_mat4_mul_vec4(mat, ph, q);
        if (q[3] == 0.0f) { s->err |= TRO_E_W_ZERO; return 0; }
        float sx = q[0] / q[3], sy = q[1] / q[3], sz = q[2] / q[3];
        s->buf.vertex_t_raster[2 * i + 0] = tro_f32_to_i32(sx);
        s->buf.vertex_t_raster[2 * i + 1] = tro_f32_to_i32(sy);
        s->buf.vertex_z_values[i] = sz;
    }
    return 1;
}

/* xyz(it_m * (n, 0)).normalize() */
static int transform_normal(struct tro_scene *s, v3 n, v3 *out)
{
    float nh[4] = { n.x, n.y, n.z, 0.0f }, q[4];
    tro_mat4_mul_vec4(s->buf.u.it_m, nh, q);
    if (q[3] != 0.0f) { s->err |= TRO_E_VEC_W_NONZERO; return 0; }
    *out = v3_normalize(v3_make(q[0], q[1], q[2]));
    return 1;
}

static inline v3 t_light(const struct tro_scene *s)
{
    return v3_make(s->buf.u.t_light_direction[0], s->buf.u.t_light_direction[1],
                   s->buf.u.t_light_direction[2]);
}

static inline float bar_dot_z(const struct tro_scene *s, v3 bar)
{
    return v3_dot(bar, v3_make(s->buf.vertex_z_values[0], s->buf.vertex_z_values[1],
                               s->buf.vertex_z_values[2]));
}

static int process_z_value(struct tro_scene *s, v3 bar, uint32_t x, uint32_t y)
{
    size_t index = (size_t)x + (size_t)(y * s->buf.width);
    float z = bar_dot_z(s, bar);
    if (z <= s->buf.z_buffer[index]) return 0;
    s->buf.z_buffer[index] = z;
    s->winner[index] = s->cur_tri;
    return 1;
}

/* vertex_uvs * bar (2x3 gemv) */
static inline void interp_uv(const struct tro_scene *s, v3 bar, float *u, float *v)
{
    const float *m = s->buf.vertex_uvs;
    float y0 = m[0] * bar.x, y1 = m[1] * bar.x;
    y0 = m[2] * bar.y + y0; y1 = m[3] * bar.y + y1;
    y0 = m[4] * bar.z + y0; y1 = m[5] * bar.z + y1;
    *u = y0; *v = y1;
}

/* ------------------------------------------------------------------------------------------
 * Pipelines (shader.rs:282-963)
 * ---------------------------------------------------------------------------------------- */

/* default: shader.rs:285-333 */
static int vs_default(struct tro_scene *s, const uint32_t *tri)
{
    v3 p[3];
    if (!get_vertex_positions(s, tri, p)) return 0;
    if (should_cull_face(p, s->buf.u.camera_direction)) return 0;
    v3 face_normal = v3_cross(v3_sub(p[1], p[0]), v3_sub(p[2], p[0]));
    v3 tn;
    if (!transform_normal(s, face_normal, &tn)) return 0;
    float diff = v3_dot(t_light(s), tn);
    s->buf.vertex_intensities[0] = diff;
    s->buf.vertex_intensities[1] = diff;
    s->buf.vertex_intensities[2] = diff;
    if (!store_vertex_transformation_results(s, p, s->buf.u.vpmv)) return 0;
    if (!store_vertex_uvs(s, tri)) return 0;
    return 1;
}

static int fs_default(struct tro_scene *s, uint32_t x, uint32_t y, v3 bar)
{
    if (!process_z_value(s, bar, x, y)) return 0;
    float u, v;
    interp_uv(s, bar, &u, &v);
    const uint8_t *c;
    if (!texel(s, 0, 0, u, v, &c)) return 0;
    const uint8_t zero[3] = { 0, 0, 0 };
    tro_color_blend(c, zero, s->buf.vertex_intensities[0], s->buf.fragment_color);
    return 1;
}

/* phong: shader.rs:349-401 */
static int vs_phong(struct tro_scene *s, const uint32_t *tri)
{
    v3 p[3];
    if (!get_vertex_positions(s, tri, p)) return 0;
    if (should_cull_face(p, s->buf.u.camera_direction)) return 0;
    for (int i = 0; i < 3; i++) {
        v3 n, tn;
        if (!vertex_normal(s, tri, i, &n)) return 0;
        if (!transform_normal(s, n, &tn)) return 0;
        s->buf.vertex_intensities[i] = v3_dot(t_light(s), tn);
    }
    if (!store_vertex_transformation_results(s, p, s->buf.u.vpmv)) return 0;
    if (!store_vertex_uvs(s, tri)) return 0;
    return 1;
}

static int fs_phong(struct tro_scene *s, uint32_t x, uint32_t y, v3 bar)
{
    if (!process_z_value(s, bar, x, y)) return 0;
    float u, v;
    interp_uv(s, bar, &u, &v);
    const uint8_t *c;
    if (!texel(s, 0, 0, u, v, &c)) return 0;
    float diff = v3_dot(bar, v3_make(s->buf.vertex_intensities[0], s->buf.vertex_intensities[1],
                                     s->buf.vertex_intensities[2]));
    const uint8_t zero[3] = { 0, 0, 0 };
    tro_color_blend(c, zero, diff, s->buf.fragment_color);
    return 1;
}

/* normal_map / specular / occlusion pass 2 vertex: cull + transform + uvs
 * (shader.rs:416-437, 475-496, 849-870) */
static int vs_plain(struct tro_scene *s, const uint32_t *tri)
{
    v3 p[3];
    if (!get_vertex_positions(s, tri, p)) return 0;
    if (should_cull_face(p, s->buf.u.camera_direction)) return 0;
    if (!store_vertex_transformation_results(s, p, s->buf.u.vpmv)) return 0;
    if (!store_vertex_uvs(s, tri)) return 0;
    return 1;
}

/* normal_map: shader.rs:439-459 */
static int fs_normal_map(struct tro_scene *s, uint32_t x, uint32_t y, v3 bar)
{
    if (!process_z_value(s, bar, x, y)) return 0;
    float u, v;
    interp_uv(s, bar, &u, &v);
    const uint8_t *c, *npx;
    if (!texel(s, 0, 0, u, v, &c)) return 0;
    if (!texel(s, 1, 1, u, v, &npx)) return 0;
    v3 tn;
    if (!transform_normal(s, decode_normal(npx), &tn)) return 0;
    float diff = v3_dot(t_light(s), tn);
    const uint8_t zero[3] = { 0, 0, 0 };
    tro_color_blend(c, zero, diff, s->buf.fragment_color);
    return 1;
}

/* specular: shader.rs:498-534 */
static int fs_specular(struct tro_scene *s, uint32_t x, uint32_t y, v3 bar)
{
    if (!process_z_value(s, bar, x, y)) return 0;
    float u, v;
    interp_uv(s, bar, &u, &v);
    const uint8_t *c, *npx, *spx;
    if (!texel(s, 0, 0, u, v, &c)) return 0;
    if (!texel(s, 1, 1, u, v, &npx)) return 0;
    v3 tn;
    if (!transform_normal(s, decode_normal(npx), &tn)) return 0;
    v3 tl = t_light(s);
    /* (2.0 * (t_n * t_light.dot(t_n)) - t_light).normalize() */
    float d = v3_dot(tl, tn);
    v3 a = v3_scale(tn, d);
    v3 r = v3_normalize(v3_sub(v3_make(2.0f * a.x, 2.0f * a.y, 2.0f * a.z), tl));
    float diff = v3_dot(tl, tn);
    if (!texel(s, 3, 3, u, v, &spx)) return 0;
    float spec = 0.6f * powf(fmaxf(r.z, 0.0f), (float)spx[0]);
    for (int k = 0; k < 3; k++)
        s->buf.fragment_color[k] = tro_f32_to_u8(fminf((diff + spec) * (float)c[k], 255.0f));
    return 1;
}

/* darboux: shader.rs:549-655 */
static int vs_darboux(struct tro_scene *s, const uint32_t *tri)
{
    v3 p[3];
    if (!get_vertex_positions(s, tri, p)) return 0;
    if (should_cull_face(p, s->buf.u.camera_direction)) return 0;
    for (int i = 0; i < 3; i++) {
        float ph[4] = { p[i].x, p[i].y, p[i].z, 1.0f }, q[4];
        tro_mat4_mul_vec4(s->buf.u.m, ph, q);
        if (q[3] == 0.0f) { s->err |= TRO_E_W_ZERO; return 0; }
        s->buf.vertex_t_positions[3 * i + 0] = q[0] / q[3];
        s->buf.vertex_t_positions[3 * i + 1] = q[1] / q[3];
        s->buf.vertex_t_positions[3 * i + 2] = q[2] / q[3];
    }
    for (int i = 0; i < 3; i++) {
        v3 n, tn;
        if (!vertex_normal(s, tri, i, &n)) return 0;
        if (!transform_normal(s, n, &tn)) return 0;
        s->buf.vertex_t_normals[3 * i + 0] = tn.x;
        s->buf.vertex_t_normals[3 * i + 1] = tn.y;
        s->buf.vertex_t_normals[3 * i + 2] = tn.z;
    }
    if (!store_vertex_transformation_results(s, p, s->buf.u.vpmv)) return 0;
    if (!store_vertex_uvs(s, tri)) return 0;
    return 1;
}

static int fs_darboux(struct tro_scene *s, uint32_t x, uint32_t y, v3 bar)
{
    if (!process_z_value(s, bar, x, y)) return 0;
    float u, v;
    interp_uv(s, bar, &u, &v);
    const uint8_t *c, *npx;
    if (!texel(s, 0, 0, u, v, &c)) return 0;
    /* util.rs:60-73: coordinates from normal_map's dims, texel from normal_map_tangent */
    if (!texel(s, 2, 1, u, v, &npx)) return 0;
    v3 nt = decode_normal(npx);

    const float *tp = s->buf.vertex_t_positions;
    const float *tn = s->buf.vertex_t_normals;
    v3 local_z = mat3_mul_v3(tn, bar);
    v3 r0 = v3_normalize(mat3_mul_v3(tp, v3_make(-1.0f, 1.0f, 0.0f)));
    v3 r1 = v3_normalize(mat3_mul_v3(tp, v3_make(-1.0f, 0.0f, 1.0f)));
    v3 r2 = v3_normalize(mat3_mul_v3(tn, bar));
    float A[9] = { r0.x, r1.x, r2.x, r0.y, r1.y, r2.y, r0.z, r1.z, r2.z }; /* rows r0,r1,r2 */
    float Ai[9];
    if (!tro_mat3_inverse(A, Ai)) { s->err |= TRO_E_SINGULAR; return 0; }
    const float *uv = s->buf.vertex_uvs;
    v3 local_x = mat3_mul_v3(Ai, v3_make(uv[2] - uv[0], uv[4] - uv[0], 0.0f));
    v3 local_y = mat3_mul_v3(Ai, v3_make(uv[3] - uv[1], uv[5] - uv[1], 0.0f));
    v3 c0 = v3_normalize(local_x), c1 = v3_normalize(local_y), c2 = v3_normalize(local_z);
    float B[9] = { c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, c2.x, c2.y, c2.z };
    v3 n = v3_normalize(mat3_mul_v3(B, nt));
    float diff = v3_dot(t_light(s), n);
    const uint8_t zero[3] = { 0, 0, 0 };
    tro_color_blend(c, zero, diff, s->buf.fragment_color);
    return 1;
}

/* shadow / occlusion pass 1: shader.rs:671-709, 809-847 */
static int vs_depth(struct tro_scene *s, const uint32_t *tri)
{
    v3 p[3];
    if (!get_vertex_positions(s, tri, p)) return 0;
    if (!store_vertex_transformation_results(s, p, s->buf.u.shadow_matrix)) return 0;
    if (!store_vertex_uvs(s, tri)) return 0;
    return 1;
}

static int fs_depth(struct tro_scene *s, uint32_t x, uint32_t y, v3 bar)
{
    size_t index = (size_t)(x + y * s->buf.width);
    float z = bar_dot_z(s, bar);
    if (z >= s->buf.shadow_buffer[index]) {
        s->buf.shadow_buffer[index] = z;
        s->stats[s->cur_pass].shadow_upd++;
    }
    return 0;
}

/* Point3::from_homogeneous(mat * (x, y, z, 1)).unwrap() */
static int project_point(struct tro_scene *s, const float mat[16], v3 p, v3 *out)
{
    float ph[4] = { p.x, p.y, p.z, 1.0f }, q[4];
    tro_mat4_mul_vec4(mat, ph, q);
    if (q[3] == 0.0f) { s->err |= TRO_E_W_ZERO; return 0; }
    *out = v3_make(q[0] / q[3], q[1] / q[3], q[2] / q[3]);
    return 1;
}

/* (coord.x.round() as u32 + (coord.y.round() as u32) * width) as usize; u32 arithmetic wraps
 * in a release build; index >= len panics. */
static int shadow_lookup(struct tro_scene *s, v3 sc, float *val)
{
    uint32_t ix = tro_f32_to_u32(roundf(sc.x));
    uint32_t iy = tro_f32_to_u32(roundf(sc.y));
    uint32_t idx = ix + iy * s->buf.width;
    if ((size_t)idx >= (size_t)s->buf.width * s->buf.height) { s->err |= TRO_E_SHADOW_OOB; return 0; }
    *val = s->buf.shadow_buffer[idx];
    return 1;
}

/* shadow pass 2: shader.rs:749-788 */
static int fs_shadow_2(struct tro_scene *s, uint32_t x, uint32_t y, v3 bar)
{
    if (!process_z_value(s, bar, x, y)) return 0;
    float sm_ivpmv[16];
    tro_mat4_mul(s->buf.u.shadow_matrix, s->buf.u.i_vpmv, sm_ivpmv);
    v3 sc;
    if (!project_point(s, sm_ivpmv, v3_make((float)x, (float)y, bar_dot_z(s, bar)), &sc)) return 0;
    float sv;
    if (!shadow_lookup(s, sc, &sv)) return 0;
    float shadow_coef = 1.0f;
    if (sc.z + 1.0f < sv) shadow_coef = 0.3f;
    float u, v;
    interp_uv(s, bar, &u, &v);
    const uint8_t *c;
    if (!texel(s, 0, 0, u, v, &c)) return 0;
    float diff = v3_dot(bar, v3_make(s->buf.vertex_intensities[0], s->buf.vertex_intensities[1],
                                     s->buf.vertex_intensities[2]));
    const uint8_t zero[3] = { 0, 0, 0 };
    tro_color_blend(c, zero, diff * shadow_coef, s->buf.fragment_color);
    return 1;
}

/* occlusion pass 2: shader.rs:872-947 */
static int fs_occlusion_2(struct tro_scene *s, uint32_t x, uint32_t y, v3 bar)
{
    if (!process_z_value(s, bar, x, y)) return 0;
    v3 fp = v3_make((float)x, (float)y, bar_dot_z(s, bar));
    v3 world;
    if (!project_point(s, s->buf.u.i_vpmv, fp, &world)) return 0;
    float sm_ivpmv[16];
    tro_mat4_mul(s->buf.u.shadow_matrix, s->buf.u.i_vpmv, sm_ivpmv);
    v3 fsc;
    if (!project_point(s, sm_ivpmv, fp, &fsc)) return 0;
    float fsv;
    if (!shadow_lookup(s, fsc, &fsv)) return 0;

    float threshold = 1.0f;
    float occlusion_coef = 1.0f;
    for (int i = 0; i < 16; i++) {
        v3 sample = v3_add(world, v3_make(s->occl_steps[3 * i], s->occl_steps[3 * i + 1],
                                          s->occl_steps[3 * i + 2]));
        v3 ssc;
        if (!project_point(s, s->buf.u.shadow_matrix, sample, &ssc)) return 0;
        float sv;
        if (!shadow_lookup(s, ssc, &sv)) return 0;
        if (sv - threshold > fsv) {
            float strength = (sv - fsv) / 20.0f;
            strength = fminf(strength, 1.0f);
            occlusion_coef -= (1.0f / 16.0f) * strength;
        }
    }
    const uint8_t white[3] = { 255, 255, 255 }, zero[3] = { 0, 0, 0 };
    tro_color_blend(white, zero, occlusion_coef, s->buf.fragment_color);
    return 1;
}

/* shader.rs:97-112 */
static int select_pipeline(struct tro_scene *s, const char *name)
{
    pass_t *p = s->passes;
    s->n_passes = 1;
    if (!strcmp(name, "default")) {
        p[0] = (pass_t){ prep_default, vs_default, fs_default };
    } else if (!strcmp(name, "phong")) {
        p[0] = (pass_t){ prep_default, vs_phong, fs_phong };
    } else if (!strcmp(name, "normal_map")) {
        p[0] = (pass_t){ prep_default, vs_plain, fs_normal_map };
    } else if (!strcmp(name, "specular")) {
        p[0] = (pass_t){ prep_default, vs_plain, fs_specular };
    } else if (!strcmp(name, "darboux")) {
        p[0] = (pass_t){ prep_default, vs_darboux, fs_darboux };
    } else if (!strcmp(name, "shadow")) {
        p[0] = (pass_t){ prep_shadow_1, vs_depth, fs_depth };
        p[1] = (pass_t){ prep_shadow_2, vs_phong, fs_shadow_2 };
        s->n_passes = 2;
    } else if (!strcmp(name, "occlusion")) {
        p[0] = (pass_t){ prep_shadow_1, vs_depth, fs_depth };
        p[1] = (pass_t){ prep_occlusion_2, vs_plain, fs_occlusion_2 };
        s->n_passes = 2;
    } else {
        return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * scene.rs
 * ---------------------------------------------------------------------------------------- */
static void *dup_mem(const void *src, size_t n)
{
    void *p = malloc(n ? n : 1);
    if (p && n) memcpy(p, src, n);
    return p;
}

tro_scene *tro_scene_new(uint32_t width, uint32_t height, const tro_mesh *mesh,
                         const tro_image tex[4], const char *pipeline_name)
{
    tro_scene *s = (tro_scene *)calloc(1, sizeof *s);
    if (!s) return NULL;
    if (!select_pipeline(s, pipeline_name)) { free(s); return NULL; }
    s->width = width;
    s->height = height;
    s->model.n_pos = mesh->n_pos; s->model.n_tex = mesh->n_tex;
    s->model.n_nrm = mesh->n_nrm; s->model.n_tri = mesh->n_tri;
    s->model.pos = (float *)dup_mem(mesh->pos, sizeof(float) * 3 * mesh->n_pos);
    s->model.tex = (float *)dup_mem(mesh->tex, sizeof(float) * 3 * mesh->n_tex);
    s->model.nrm = (float *)dup_mem(mesh->nrm, sizeof(float) * 3 * mesh->n_nrm);
    s->model.idx = (uint32_t *)dup_mem(mesh->idx, sizeof(uint32_t) * 9 * mesh->n_tri);
    for (int k = 0; k < 4; k++) {
        s->model.img_w[k] = tex[k].w;
        s->model.img_h[k] = tex[k].h;
        s->model.img[k] = (uint8_t *)dup_mem(tex[k].rgb, (size_t)3 * tex[k].w * tex[k].h);
    }
    size_t n = (size_t)width * height;
    s->buf.width = width;
    s->buf.height = height;
    s->buf.z_buffer = (float *)calloc(n ? n : 1, sizeof(float));      /* shader.rs:46 */
    s->buf.shadow_buffer = (float *)calloc(n ? n : 1, sizeof(float)); /* shader.rs:47 */
    s->frame_buffer = (uint8_t *)calloc(n ? 3 * n : 1, 1);            /* scene.rs:71 */
    s->winner = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    memset(s->winner, 0xFF, (n ? n : 1) * sizeof(uint32_t));
    /* scene.rs:66-69 */
    s->light_direction[0] = 0.0f; s->light_direction[1] = 0.0f; s->light_direction[2] = -1.0f;
    s->look_from[0] = 0.0f; s->look_from[1] = 0.0f; s->look_from[2] = 1.0f;
    s->look_at[0] = s->look_at[1] = s->look_at[2] = 0.0f;
    s->up[0] = 0.0f; s->up[1] = 1.0f; s->up[2] = 0.0f;
    s->band_y0 = 0;
    s->band_y1 = (int32_t)height - 1;
    return s;
}

void tro_scene_free(tro_scene *s)
{
    if (!s) return;
    free(s->model.pos); free(s->model.tex); free(s->model.nrm); free(s->model.idx);
    for (int k = 0; k < 4; k++) free(s->model.img[k]);
    free(s->buf.z_buffer); free(s->buf.shadow_buffer); free(s->frame_buffer); free(s->winner);
    free(s);
}

void tro_scene_clear(tro_scene *s)
{
    size_t n = (size_t)s->width * s->height;
    for (size_t i = 0; i < n; i++) {
        s->buf.z_buffer[i] = -3.40282347e+38f;      /* f32::MIN */
        s->buf.shadow_buffer[i] = -3.40282347e+38f;
        s->frame_buffer[3 * i + 0] = 0;
        s->frame_buffer[3 * i + 1] = 0;
        s->frame_buffer[3 * i + 2] = 0;
        s->winner[i] = 0xFFFFFFFFu;
    }
}

/* Not in the reference: restrict the colour pass to output rows [row0, row1) (row 0 = top, as
 * get_frame_buffer returns them) by cutting the clamp rectangle of scene.rs:236-239 -- the shard
 * hook SURVEY 8e names.  Pixels outside the band keep what they held. */
int tro_scene_set_output_band(tro_scene *s, uint32_t row0, uint32_t row1)
{
    if (row0 >= row1 || row1 > s->height) return -1;
    s->band_y0 = (int32_t)(s->height - row1);
    s->band_y1 = (int32_t)(s->height - row0) - 1;
    return 0;
}

void tro_scene_set_light_direction(tro_scene *s, const float v[3])
{
    memcpy(s->light_direction, v, 3 * sizeof(float));
}

void tro_scene_set_camera(tro_scene *s, const float from[3], const float at[3], const float up[3])
{
    memcpy(s->look_from, from, 3 * sizeof(float));
    memcpy(s->look_at, at, 3 * sizeof(float));
    memcpy(s->up, up, 3 * sizeof(float));
}

/* scene.rs:174-197 */
static inline v3 to_barycentric(const int32_t *c, int32_t px, int32_t py)
{
    /* i32 subtraction wraps in a release build; then `as f32` */
    float a0 = (float)(int32_t)((uint32_t)c[2] - (uint32_t)c[0]); /* m12 - m11 */
    float a1 = (float)(int32_t)((uint32_t)c[4] - (uint32_t)c[0]); /* m13 - m11 */
    float a2 = (float)(int32_t)((uint32_t)c[0] - (uint32_t)px);   /* m11 - p.x */
    float b0 = (float)(int32_t)((uint32_t)c[3] - (uint32_t)c[1]); /* m22 - m21 */
    float b1 = (float)(int32_t)((uint32_t)c[5] - (uint32_t)c[1]); /* m23 - m21 */
    float b2 = (float)(int32_t)((uint32_t)c[1] - (uint32_t)py);   /* m21 - p.y */
    v3 rc = v3_cross(v3_make(a0, a1, a2), v3_make(b0, b1, b2));
    if (fabsf(rc.z) < 1.0f) return v3_make(-1.0f, 1.0f, 1.0f);
    return v3_make(1.0f - (rc.x + rc.y) / rc.z, rc.x / rc.z, rc.y / rc.z);
}

void tro_barycentric(const int32_t raster[6], int32_t px, int32_t py, float out[3])
{
    v3 b = to_barycentric(raster, px, py);
    out[0] = b.x; out[1] = b.y; out[2] = b.z;
}

static inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

int tro_scene_render(tro_scene *s)
{
    s->err = 0;
    memset(s->stats, 0, sizeof s->stats);
    for (int pi = 0; pi < s->n_passes; pi++) {
        const pass_t *pass = &s->passes[pi];
        tro_stats *st = &s->stats[pi];
        s->cur_pass = pi;
        s->err |= pass->prepare(s);
        if (s->err) return s->err;
        for (uint32_t t = 0; t < s->model.n_tri; t++) {
            const uint32_t *tri = &s->model.idx[9 * (size_t)t];
            s->cur_tri = t;
            st->tri_total++;
            int keep = pass->vertex(s, tri);
            if (s->err) return s->err;
            if (!keep) continue;
            st->tri_kept++;

            const int32_t *r = s->buf.vertex_t_raster;
            int32_t llx = imin(imin(r[0], r[2]), r[4]);
            int32_t lly = imin(imin(r[1], r[3]), r[5]);
            int32_t urx = imax(imax(r[0], r[2]), r[4]);
            int32_t ury = imax(imax(r[1], r[3]), r[5]);
            int32_t x_min = imax(0, llx);
            int32_t x_max = imin(urx, (int32_t)(s->width - 1u));
            /* depth passes fill the whole shadow buffer on every shard (lookups are in light space,
               shader.rs:774-778); only the colour pass is cut to the band */
            const int colour = (pi == s->n_passes - 1);
            int32_t y_min = imax(colour ? s->band_y0 : 0, lly);
            int32_t y_max = imin(ury, colour ? s->band_y1 : (int32_t)(s->height - 1u));
            for (int32_t i = x_min; i <= x_max; i++) {
                for (int32_t j = y_min; j <= y_max; j++) {
                    st->bbox_px++;
                    v3 bar = to_barycentric(r, i, j);
                    if (bar.x < 0.0f || bar.y < 0.0f || bar.z < 0.0f) continue;
                    st->frag_covered++;
                    int drawn = pass->fragment(s, (uint32_t)i, (uint32_t)j, bar);
                    if (s->err) return s->err;
                    if (!drawn) continue;
                    st->frag_accept++;
                    size_t pixel_index = (size_t)(i + j * (int32_t)s->width);
                    s->frame_buffer[3 * pixel_index + 0] = s->buf.fragment_color[0];
                    s->frame_buffer[3 * pixel_index + 1] = s->buf.fragment_color[1];
                    s->frame_buffer[3 * pixel_index + 2] = s->buf.fragment_color[2];
                }
            }
        }
    }
    return s->err;
}

static void flip_copy(const uint8_t *src, uint8_t *dst, uint32_t w, uint32_t h)
{
    size_t row = (size_t)3 * w;
    for (uint32_t r = 0; r < h; r++) memcpy(dst + (size_t)r * row, src + (size_t)(h - 1 - r) * row, row);
}

void tro_scene_get_frame_buffer(const tro_scene *s, uint8_t *rgb)
{
    flip_copy(s->frame_buffer, rgb, s->width, s->height);
}

static void depth_view(const tro_scene *s, const float *src, uint8_t *rgb)
{
    size_t n = (size_t)s->width * s->height;
    uint8_t *tmp = (uint8_t *)malloc(n ? 3 * n : 1);
    for (size_t i = 0; i < n; i++) {
        uint8_t v = tro_f32_to_u8(src[i]);
        tmp[3 * i + 0] = v; tmp[3 * i + 1] = v; tmp[3 * i + 2] = v;
    }
    flip_copy(tmp, rgb, s->width, s->height);
    free(tmp);
}

void tro_scene_get_z_buffer(const tro_scene *s, uint8_t *rgb) { depth_view(s, s->buf.z_buffer, rgb); }
void tro_scene_get_shadow_buffer(const tro_scene *s, uint8_t *rgb) { depth_view(s, s->buf.shadow_buffer, rgb); }

const float *tro_scene_z_f32(const tro_scene *s) { return s->buf.z_buffer; }
const float *tro_scene_shadow_f32(const tro_scene *s) { return s->buf.shadow_buffer; }
const uint32_t *tro_scene_winner_u32(const tro_scene *s) { return s->winner; }
const uint8_t *tro_scene_frame_raw(const tro_scene *s) { return s->frame_buffer; }

void tro_scene_stats(const tro_scene *s, tro_stats out[2]) { memcpy(out, s->stats, sizeof s->stats); }
void tro_scene_uniforms(const tro_scene *s, tro_uniforms *out) { *out = s->buf.u; }
