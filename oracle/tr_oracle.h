/*
 * tr_oracle.h -- CPU oracle for the tiny_renderer triangle-fill path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / timed CPU baseline.
 *
 * It is a plain-C, single-threaded restatement of the reference's algorithm
 * (src/scene.rs, src/scene/shader.rs, src/scene/util.rs) with the f32
 * operation order of nalgebra 0.31.4.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures,
 * cannot be built here (Rust, no toolchain, crates absent) and its arithmetic
 * lives in nalgebra 0.31.4 / image 0.24.5 / obj-rs 0.7.0, none of which is
 * under /root/reference.  The restatement is pinned only by hand-derived known
 * answers, the survey-time workload counts and algebraic
 * invariants -- see tests/test_oracle_*.py and DESIGN.md.
 */
#ifndef TR_ORACLE_H
#define TR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* obj::raw::RawObj as used by the path (util.rs:25-31, shader.rs:136-147, 363-367,
 * scene.rs:216-226): positions xyz (w dropped), tex_coords uvw, normals xyz,
 * polygons = first three (pos,tex,nrm) index triples, zero based. */
typedef struct tro_mesh {
    const float *pos;    /* n_pos * 3 */
    const float *tex;    /* n_tex * 3 */
    const float *nrm;    /* n_nrm * 3 */
    const uint32_t *idx; /* n_tri * 9 : p0,t0,n0, p1,t1,n1, p2,t2,n2 */
    uint32_t n_pos, n_tex, n_nrm, n_tri;
} tro_mesh;

/* image::RgbImage: tightly packed rgb8, row 0 = top. */
typedef struct tro_image {
    const uint8_t *rgb;
    uint32_t w, h;
} tro_image;

/* Error bits: each one is a site where the reference would panic. */
enum {
    TRO_OK = 0,
    TRO_E_UNKNOWN_PIPELINE = 1 << 0, /* shader.rs:108 */
    TRO_E_W_ZERO = 1 << 1,           /* Point3::from_homogeneous(..).unwrap(), shader.rs:158 ... */
    TRO_E_SINGULAR = 1 << 2,         /* try_inverse().unwrap(), shader.rs:224,277,278,631 */
    TRO_E_TEX_OOB = 1 << 3,          /* get_pixel out of bounds, util.rs:40,52,68,82 */
    TRO_E_SHADOW_OOB = 1 << 4,       /* shadow_buffer[idx], shader.rs:778,912,935 */
    TRO_E_ROTATION = 1 << 5,         /* rotation_between(..).unwrap(), shader.rs:921 */
    TRO_E_VEC_W_NONZERO = 1 << 6,    /* Vector3::from_homogeneous(..).unwrap(), shader.rs:227 ... */
    TRO_E_INDEX_OOB = 1 << 7         /* positions/tex_coords/normals index out of range */
};

/* Counters per render pass (workload sizes of BASELINE.md section 4). */
typedef struct tro_stats {
    uint64_t tri_total;    /* polygons visited */
    uint64_t tri_kept;     /* vertex shader returned true */
    uint64_t bbox_px;      /* inner-loop iterations (scene.rs:240-241) */
    uint64_t frag_covered; /* passed the inside test (scene.rs:245) */
    uint64_t frag_accept;  /* fragment shader ran to completion / returned true */
    uint64_t shadow_upd;   /* shadow-buffer stores in a depth pass (shader.rs:703-705) */
} tro_stats;

/* Uniform part of shader.rs:14-38 `Buffer` -- exposed so host-side prepare code can
 * be compared matrix by matrix.  Matrices are column-major like nalgebra. */
typedef struct tro_uniforms {
    float camera_direction[3];
    float t_light_direction[3];
    float vpmv[16];
    float i_vpmv[16];
    float m[16];
    float i_m[16];
    float it_m[16];
    float shadow_matrix[16];
} tro_uniforms;

typedef struct tro_scene tro_scene;

/* Scene::new (scene.rs:47-88).  tex[] order = texture, normal_map, normal_map_tangent,
 * specular_map.  Inputs are copied.  Returns NULL for an unknown pipeline name. */
tro_scene *tro_scene_new(uint32_t width, uint32_t height, const tro_mesh *mesh,
                         const tro_image tex[4], const char *pipeline_name);
void tro_scene_free(tro_scene *s);

void tro_scene_clear(tro_scene *s);                                 /* scene.rs:128-137 */
void tro_scene_set_light_direction(tro_scene *s, const float v[3]); /* scene.rs:140-142 */
void tro_scene_set_camera(tro_scene *s, const float from[3], const float at[3],
                          const float up[3]);                       /* scene.rs:145-149 */
/* Scene::render (scene.rs:151-268).  Returns 0 or the OR of TRO_E_* bits; on an error the
 * render stops at the panic site (buffers hold whatever was written so far). */
int tro_scene_render(tro_scene *s);
/* Not in the reference (SURVEY 8e's shard hook): cut the colour pass's clamp rectangle
 * (scene.rs:236-239) to output rows [row0, row1), row 0 = top.  0 ok, -1 bad rows. */
int tro_scene_set_output_band(tro_scene *s, uint32_t row0, uint32_t row1);

/* scene.rs:92-125: rgb8 images, 3*W*H bytes, row 0 = top (flipped). */
void tro_scene_get_frame_buffer(const tro_scene *s, uint8_t *rgb);
void tro_scene_get_z_buffer(const tro_scene *s, uint8_t *rgb);
void tro_scene_get_shadow_buffer(const tro_scene *s, uint8_t *rgb);

/* Parity taps (not in the reference): raw buffers in the reference's internal layout,
 * index = x + y*W with row 0 = bottom. */
const float *tro_scene_z_f32(const tro_scene *s);
const float *tro_scene_shadow_f32(const tro_scene *s);
const uint32_t *tro_scene_winner_u32(const tro_scene *s); /* polygon index of last accepted
                                                             fragment, 0xFFFFFFFF = none */
const uint8_t *tro_scene_frame_raw(const tro_scene *s);   /* unflipped frame_buffer */

/* Counters of the most recent render(): pass 0 and pass 1. */
void tro_scene_stats(const tro_scene *s, tro_stats out[2]);
/* Uniforms as left by the most recent render() (after the last pass' prepare). */
void tro_scene_uniforms(const tro_scene *s, tro_uniforms *out);

/* The prepares on their own (shader.rs:183-279).  kind: 0 = default_prepare,
 * 1 = shadow_pass_prepare_1, 2 = shadow_pass_prepare_2.  `u` is read-modify-write like
 * the reference's Buffer.  Returns 0 or TRO_E_* bits. */
int tro_prepare(int kind, tro_uniforms *u, uint32_t width, uint32_t height,
                const float light[3], const float from[3], const float at[3],
                const float up[3]);

/* Primitive taps for known-answer tests (Appendix A). */
void tro_mat4_mul(const float a[16], const float b[16], float out[16]);
void tro_mat4_mul_vec4(const float a[16], const float v[4], float out[4]);
int tro_mat4_inverse(const float a[16], float out[16]);
int tro_mat3_inverse(const float a[9], float out[9]);
void tro_barycentric(const int32_t raster[6], int32_t px, int32_t py, float out[3]);
int32_t tro_f32_to_i32(float v);
uint32_t tro_f32_to_u32(float v);
uint8_t tro_f32_to_u8(float v);
void tro_color_blend(const uint8_t c1[3], const uint8_t c2[3], float t, uint8_t out[3]);
/* 16 occlusion step vectors `rot * (sin a_k, 0, cos a_k) * 0.02` (shader.rs:916-929);
 * returns 0 or TRO_E_ROTATION. */
int tro_occlusion_steps(const float light_direction[3], float out[48]);

#ifdef __cplusplus
}
#endif
#endif
