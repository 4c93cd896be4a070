// tr_prepare.cpp -- host-side pass preparation: the reference's `prepare` closures
// (src/scene/shader.rs:183-279) plus the frame constants its fragment closures recompute for
// every fragment (shadow_matrix * i_vpmv, shader.rs:763-764; the occlusion sample offsets,
// shader.rs:916-929).  Pure host C++, no GPU needed; exported through the C ABI as
// tr_prepare_uniforms so it can be checked on its own.
//
// Build with -ffp-contract=off: every product and sum below must round once, in nalgebra's
// order.
#include "tr_prepare.h"

#include <math.h>
#include <string.h>

#include "tr_math.h"

namespace tr {

namespace {

struct Mat4 {
    float v[16];  // column-major: (r,c) = v[4*c + r]
    float &at(int r, int c) { return v[4 * c + r]; }
    float at(int r, int c) const { return v[4 * c + r]; }
};

Mat4 identity4()
{
    Mat4 m;
    for (int i = 0; i < 16; i++) m.v[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    return m;
}

// A * B, column j = A * (column j of B), each by the column-accumulating gemv of tr_math.h.
Mat4 mul(const Mat4 &a, const Mat4 &b)
{
    Mat4 r;
    for (int j = 0; j < 4; j++) {
        const float *c = &b.v[4 * j];
        vec4 y = mul_m4_v4(a.v, c[0], c[1], c[2], c[3]);
        r.v[4 * j + 0] = y.x;
        r.v[4 * j + 1] = y.y;
        r.v[4 * j + 2] = y.z;
        r.v[4 * j + 3] = y.w;
    }
    return r;
}

Mat4 transpose(const Mat4 &a)
{
    Mat4 r;
    for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) r.at(rr, c) = a.at(c, rr);
    return r;
}

// Determinant of the 3x3 left after deleting one row and one column, written as the six signed
// triple products nalgebra's do_inverse4 (the gluInvertMatrix expansion) uses.  `m` is the
// column-major slice; the six index triples select the factors in the published order.
struct Term {
    int sign, a, b, c;
};

float cofactor(const float *m, const Term t[6])
{
    float acc = 0.0f;
    for (int k = 0; k < 6; k++) {
        // (-m[a]) * m[b] * m[c] and m[a] * m[b] * m[c] round identically; subtraction of a
        // product equals addition of its negation, so one accumulate form covers both.
        float p = (t[k].sign < 0 && k == 0) ? (-m[t[k].a]) * m[t[k].b] * m[t[k].c]
                                            : m[t[k].a] * m[t[k].b] * m[t[k].c];
        if (k == 0)
            acc = p;
        else if (t[k].sign < 0)
            acc = acc - p;
        else
            acc = acc + p;
    }
    return acc;
}

// 4x4 try_inverse (shader.rs:224,277,278).  false when the determinant is zero.
bool inverse(const Mat4 &a, Mat4 &out)
{
    static const Term T[16][6] = {
        /* 0*/ {{+1, 5, 10, 15}, {-1, 5, 11, 14}, {-1, 9, 6, 15}, {+1, 9, 7, 14}, {+1, 13, 6, 11}, {-1, 13, 7, 10}},
        /* 1*/ {{-1, 1, 10, 15}, {+1, 1, 11, 14}, {+1, 9, 2, 15}, {-1, 9, 3, 14}, {-1, 13, 2, 11}, {+1, 13, 3, 10}},
        /* 2*/ {{+1, 1, 6, 15}, {-1, 1, 7, 14}, {-1, 5, 2, 15}, {+1, 5, 3, 14}, {+1, 13, 2, 7}, {-1, 13, 3, 6}},
        /* 3*/ {{-1, 1, 6, 11}, {+1, 1, 7, 10}, {+1, 5, 2, 11}, {-1, 5, 3, 10}, {-1, 9, 2, 7}, {+1, 9, 3, 6}},
        /* 4*/ {{-1, 4, 10, 15}, {+1, 4, 11, 14}, {+1, 8, 6, 15}, {-1, 8, 7, 14}, {-1, 12, 6, 11}, {+1, 12, 7, 10}},
        /* 5*/ {{+1, 0, 10, 15}, {-1, 0, 11, 14}, {-1, 8, 2, 15}, {+1, 8, 3, 14}, {+1, 12, 2, 11}, {-1, 12, 3, 10}},
        /* 6*/ {{-1, 0, 6, 15}, {+1, 0, 7, 14}, {+1, 4, 2, 15}, {-1, 4, 3, 14}, {-1, 12, 2, 7}, {+1, 12, 3, 6}},
        /* 7*/ {{+1, 0, 6, 11}, {-1, 0, 7, 10}, {-1, 4, 2, 11}, {+1, 4, 3, 10}, {+1, 8, 2, 7}, {-1, 8, 3, 6}},
        /* 8*/ {{+1, 4, 9, 15}, {-1, 4, 11, 13}, {-1, 8, 5, 15}, {+1, 8, 7, 13}, {+1, 12, 5, 11}, {-1, 12, 7, 9}},
        /* 9*/ {{-1, 0, 9, 15}, {+1, 0, 11, 13}, {+1, 8, 1, 15}, {-1, 8, 3, 13}, {-1, 12, 1, 11}, {+1, 12, 3, 9}},
        /*10*/ {{+1, 0, 5, 15}, {-1, 0, 7, 13}, {-1, 4, 1, 15}, {+1, 4, 3, 13}, {+1, 12, 1, 7}, {-1, 12, 3, 5}},
        /*11*/ {{-1, 0, 5, 11}, {+1, 0, 7, 9}, {+1, 4, 1, 11}, {-1, 4, 3, 9}, {-1, 8, 1, 7}, {+1, 8, 3, 5}},
        /*12*/ {{-1, 4, 9, 14}, {+1, 4, 10, 13}, {+1, 8, 5, 14}, {-1, 8, 6, 13}, {-1, 12, 5, 10}, {+1, 12, 6, 9}},
        /*13*/ {{+1, 0, 9, 14}, {-1, 0, 10, 13}, {-1, 8, 1, 14}, {+1, 8, 2, 13}, {+1, 12, 1, 10}, {-1, 12, 2, 9}},
        /*14*/ {{-1, 0, 5, 14}, {+1, 0, 6, 13}, {+1, 4, 1, 14}, {-1, 4, 2, 13}, {-1, 12, 1, 6}, {+1, 12, 2, 5}},
        /*15*/ {{+1, 0, 5, 10}, {-1, 0, 6, 9}, {-1, 4, 1, 10}, {+1, 4, 2, 9}, {+1, 8, 1, 6}, {-1, 8, 2, 5}},
    };
    const float *m = a.v;
    float inv[16];
    for (int i = 0; i < 16; i++) inv[i] = cofactor(m, T[i]);
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0f) return false;
    float inv_det = 1.0f / det;
    for (int i = 0; i < 16; i++) out.v[i] = inv[i] * inv_det;
    return true;
}

vec3 v3(const float *p) { return make3(p[0], p[1], p[2]); }

// default_prepare, shader.rs:183-230
int default_prepare(tr_uniforms *u, uint32_t width, uint32_t height, vec3 light, vec3 from, vec3 at,
                    vec3 up)
{
    int err = 0;
    vec3 new_z = normalize3(sub3(from, at));
    float d = dot3(new_z, up);
    vec3 new_y = normalize3(sub3(up, make3(d * new_z.x, d * new_z.y, d * new_z.z)));
    vec3 new_x = normalize3(cross3(new_y, new_z));

    Mat4 model = identity4();
    model.at(0, 0) = new_x.x; model.at(0, 1) = new_x.y; model.at(0, 2) = new_x.z;
    model.at(1, 0) = new_y.x; model.at(1, 1) = new_y.y; model.at(1, 2) = new_y.z;
    model.at(2, 0) = new_z.x; model.at(2, 1) = new_z.y; model.at(2, 2) = new_z.z;

    Mat4 view = identity4();
    view.at(0, 3) = -from.x;
    view.at(1, 3) = -from.y;
    view.at(2, 3) = -from.z;

    Mat4 proj = identity4();
    proj.at(3, 2) = -1.0f / 5.0f;

    const float w = (float)(width - 1u), h = (float)(height - 1u), depth = 255.0f;
    Mat4 viewport = identity4();
    viewport.at(0, 0) = w / 2.0f; viewport.at(0, 3) = w / 2.0f;
    viewport.at(1, 1) = h / 2.0f; viewport.at(1, 3) = h / 2.0f;
    viewport.at(2, 2) = depth / 2.0f; viewport.at(2, 3) = depth / 2.0f;

    Mat4 vpmv = mul(mul(mul(viewport, proj), model), view);  // left to right, shader.rs:221
    memcpy(u->vpmv, vpmv.v, sizeof vpmv.v);
    memcpy(u->m, model.v, sizeof model.v);

    Mat4 it_m;
    if (inverse(transpose(model), it_m))
        memcpy(u->it_m, it_m.v, sizeof it_m.v);
    else
        err = TR_E_SINGULAR;
    u->camera_direction[0] = new_z.x;
    u->camera_direction[1] = new_z.y;
    u->camera_direction[2] = new_z.z;

    // Vector3::from_homogeneous(m * light.to_homogeneous()).unwrap().normalize()
    vec4 tl = mul_m4_v4(model.v, light.x, light.y, light.z, 0.0f);
    if (tl.w != 0.0f) err = TR_E_SINGULAR;
    vec3 tln = normalize3(make3(tl.x, tl.y, tl.z));
    u->t_light_direction[0] = tln.x;
    u->t_light_direction[1] = tln.y;
    u->t_light_direction[2] = tln.z;
    return err;
}

}  // namespace

int prepare_uniforms(int kind, tr_uniforms *u, uint32_t width, uint32_t height, const float light[3],
                     const float from[3], const float at[3], const float up[3])
{
    int err;
    switch (kind) {
    case 0:
        return default_prepare(u, width, height, v3(light), v3(from), v3(at), v3(up));
    case 1:  // shadow_pass_prepare_1, shader.rs:234-255: the camera sits at the light
        err = default_prepare(u, width, height, v3(light), v3(light), v3(at), v3(up));
        memcpy(u->shadow_matrix, u->vpmv, sizeof u->vpmv);
        return err;
    case 2: {  // shadow_pass_prepare_2, shader.rs:259-279
        err = default_prepare(u, width, height, v3(light), v3(from), v3(at), v3(up));
        Mat4 a, inv;
        memcpy(a.v, u->vpmv, sizeof a.v);
        if (inverse(a, inv))
            memcpy(u->i_vpmv, inv.v, sizeof inv.v);
        else
            err = TR_E_SINGULAR;
        memcpy(a.v, u->m, sizeof a.v);
        if (inverse(a, inv))
            memcpy(u->i_m, inv.v, sizeof inv.v);
        else
            err = TR_E_SINGULAR;
        return err;
    }
    default:
        return TR_E_INVALID;
    }
}

void shadow_times_inverse(const tr_uniforms *u, float out[16])
{
    Mat4 a, b;
    memcpy(a.v, u->shadow_matrix, sizeof a.v);
    memcpy(b.v, u->i_vpmv, sizeof b.v);
    Mat4 r = mul(a, b);
    memcpy(out, r.v, sizeof r.v);
}

// Rotation3::rotation_between((0,0,1), L).unwrap() applied to the 16 sample directions and
// scaled by the step size (shader.rs:916-929), with L = xyz(i_m * (t_light, 0)) (shader.rs:882-885).
int occlusion_steps(const tr_uniforms *u, float out[48])
{
    vec4 lh = mul_m4_v4(u->i_m, u->t_light_direction[0], u->t_light_direction[1],
                        u->t_light_direction[2], 0.0f);
    if (lh.w != 0.0f) return TR_E_SINGULAR;
    const vec3 a = make3(0.0f, 0.0f, 1.0f), b = make3(lh.x, lh.y, lh.z);

    vec3 c0 = make3(1.0f, 0.0f, 0.0f), c1 = make3(0.0f, 1.0f, 0.0f), c2 = make3(0.0f, 0.0f, 1.0f);
    const float an = sqrtf(dot3(a, a)), bn = sqrtf(dot3(b, b));
    if (an > 0.0f && bn > 0.0f) {
        const vec3 na = make3(a.x / an, a.y / an, a.z / an), nb = make3(b.x / bn, b.y / bn, b.z / bn);
        const vec3 c = cross3(na, nb);
        const float sq = dot3(c, c);
        const float eps = 1.1920929e-7f;
        if (sq > eps * eps) {
            const float cn = sqrtf(sq);
            const float ux = c.x / cn, uy = c.y / cn, uz = c.z / cn;
            const float angle = acosf(dot3(na, nb)) * 1.0f;
            if (angle != 0.0f) {
                const float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
                const float sn = sinf(angle), cs = cosf(angle);
                const float omc = 1.0f - cs;
                // rows of the rotation as from_axis_angle lists them
                const float r00 = sqx + (1.0f - sqx) * cs, r01 = ux * uy * omc - uz * sn, r02 = ux * uz * omc + uy * sn;
                const float r10 = ux * uy * omc + uz * sn, r11 = sqy + (1.0f - sqy) * cs, r12 = uy * uz * omc - ux * sn;
                const float r20 = ux * uz * omc - uy * sn, r21 = uy * uz * omc + ux * sn, r22 = sqz + (1.0f - sqz) * cs;
                c0 = make3(r00, r10, r20);
                c1 = make3(r01, r11, r21);
                c2 = make3(r02, r12, r22);
            }
        } else if (dot3(na, nb) < 0.0f) {
            return TR_E_SINGULAR;  // antiparallel: rotation_between is None, shader.rs:921 panics
        }
    }
    const float step_size = 0.02f;
    const float angle_coef = (2.0f * 3.14159265358979323846f) / 16.0f;
    for (int i = 0; i < 16; i++) {
        const vec3 g = make3(sinf(angle_coef * (float)i), 0.0f, cosf(angle_coef * (float)i));
        const vec3 s = scale3(mul_m3_v3(c0, c1, c2, g), step_size);
        out[3 * i + 0] = s.x;
        out[3 * i + 1] = s.y;
        out[3 * i + 2] = s.z;
    }
    return 0;
}

}  // namespace tr
