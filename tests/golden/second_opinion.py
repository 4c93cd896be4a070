"""A SECOND, independent restatement of the reference's triangle-fill path -- test infrastructure only.

Why it exists: the reference has no tests, fixtures or golden vectors and cannot be built here (Rust, crates not
vendored), so `oracle/tr_oracle.c` -- the restatement every GPU parity test compares with -- is pinned by nothing
upstream.  This file guards against ONE failure mode of that situation: a misreading of the Rust source that the C
oracle and the kernels share (they were written by the same reader).  It was written from the Rust source alone --
/root/reference/src/scene.rs:92-268, src/scene/shader.rs:116-963, src/scene/util.rs:7-83 -- not from the C, in a
different language and with a different structure (numpy, every polygon's bounding box evaluated as one array
expression; the vertex stage vectorised over all polygons), and tests/test_second_opinion.py compares its frame, z
buffer and winning polygon per pixel with the C oracle's bit for bit.

What agreement proves: both readings of the control flow, casts, clamps, loop bounds, comparison directions, pass
structure and closure formulas are the same.  What it does NOT prove: nalgebra's operation ORDER (dot products, matrix
products, inverses: SURVEY.md Appendix A) -- this file applies the same published readings of nalgebra 0.31.4 (the
crate's source is not on this machine), so that part of the parity stays "unpinned" (DESIGN.md section 2).

Everything is np.float32 arithmetic, one IEEE rounding per operation (numpy never fuses a multiply with an add);
powf / sinf / cosf / acosf / roundf are the C library's, called through ctypes (Rust's f32 methods call the same libm).
"""
import ctypes
import ctypes.util

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("powf", "sinf", "cosf", "acosf", "roundf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float] * (2 if _n == "powf" else 1)

F32_MIN = f32(np.finfo(np.float32).min)   # f32::MIN (scene.rs:131)


def _libm1(name, x):
    fn = getattr(_libm, name)
    x = np.asarray(x, f32)
    return np.array([fn(float(v)) for v in x.ravel()], f32).reshape(x.shape)


def _powf(a, b):
    a, b = np.broadcast_arrays(np.asarray(a, f32), np.asarray(b, f32))
    return np.array([_libm.powf(float(p), float(q)) for p, q in zip(a.ravel(), b.ravel())], f32).reshape(a.shape)


# ---- Rust `as` casts (saturating, truncating toward zero, NaN -> 0) -------------------------------------------------
def _as_int(x, lo, hi, dtype):
    x = np.asarray(x, f32)
    t = np.trunc(x.astype(np.float64))
    t = np.where(np.isnan(t), 0.0, t)
    return np.clip(t, lo, hi).astype(dtype)


def as_i32(x):
    return _as_int(x, -2147483648.0, 2147483647.0, np.int64)


def as_u32(x):
    return _as_int(x, 0.0, 4294967295.0, np.int64)


def as_u8(x):
    return _as_int(x, 0.0, 255.0, np.uint8)


# ---- nalgebra 0.31 operations, in the order SURVEY.md Appendix A reads them -------------------------------------------
def dot3(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def normalize(v):
    with np.errstate(all="ignore"):   # (a zero vector gives NaNs, as upstream: no guard)
        n = np.sqrt(dot3(v, v))
        return [v[0] / n, v[1] / n, v[2] / n]


def mat_vec(cols, v):
    """M * v, M given as its columns (lists of components): y = col0*v0; y += col1*v1; ..."""
    out = []
    for i in range(len(cols[0])):
        acc = cols[0][i] * v[0]
        for k in range(1, len(cols)):
            acc = acc + cols[k][i] * v[k]
        out.append(acc)
    return out


def mat4_mul(A, B):
    """4x4 product, matrices as lists of 4 columns of 4 floats; column j of the result = A * (column j of B)."""
    return [mat_vec(A, B[j]) for j in range(4)]


def mat4_from_rows(rows):
    return [[f32(rows[r][c]) for r in range(4)] for c in range(4)]


def mat4_transpose(M):
    return [[M[r][c] for r in range(4)] for c in range(4)]


def mat4_inverse(M):
    """Matrix4::try_inverse: the cofactor expansion on column-major storage (the MESA gluInvertMatrix form nalgebra's
    do_inverse4 uses), det from the first row of cofactors, every cofactor times 1/det.  None when det == 0."""
    m = [M[c][r] for c in range(4) for r in range(4)]   # column-major slice
    inv = [None] * 16
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10]
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10]
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9]
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9]
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10]
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10]
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9]
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9]
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6]
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6]
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5]
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5]
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6]
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6]
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5]
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5]
    det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12]
    if det == f32(0.0):
        return None
    inv_det = f32(1.0) / det
    return [[inv[c * 4 + r] * inv_det for r in range(4)] for c in range(4)]


def mat3_inverse(rows):
    """Matrix3::try_inverse on a matrix given by its ROWS (m11 .. m33 as in nalgebra's source); returns rows of the
    inverse (arrays), and a mask of singular entries (determinant == 0: the reference's unwrap() panics)."""
    (m11, m12, m13), (m21, m22, m23), (m31, m32, m33) = rows
    minor_m12_m23 = m22 * m33 - m32 * m23
    minor_m11_m23 = m21 * m33 - m31 * m23
    minor_m11_m22 = m21 * m32 - m31 * m22
    det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22
    inv = [[minor_m12_m23 / det, (m13 * m32 - m33 * m12) / det, (m12 * m23 - m22 * m13) / det],
           [-minor_m11_m23 / det, (m11 * m33 - m31 * m13) / det, (m13 * m21 - m23 * m11) / det],
           [minor_m11_m22 / det, (m12 * m31 - m32 * m11) / det, (m11 * m22 - m21 * m12) / det]]
    return inv, det == f32(0.0)


class Panic(Exception):
    """The reference would panic here (unwrap on None, get_pixel / index out of range)."""


# ---- shader.rs:183-279: the prepares -------------------------------------------------------------------------------
def default_prepare(buf, width, height, light, look_from, look_at, up):
    light, look_from, look_at, up = ([f32(c) for c in v] for v in (light, look_from, look_at, up))
    new_z = normalize([look_from[k] - look_at[k] for k in range(3)])
    d = dot3(new_z, up)
    new_y = normalize([up[k] - d * new_z[k] for k in range(3)])
    new_x = normalize(cross(new_y, new_z))
    z, o = f32(0.0), f32(1.0)
    model = mat4_from_rows([[new_x[0], new_x[1], new_x[2], z], [new_y[0], new_y[1], new_y[2], z],
                            [new_z[0], new_z[1], new_z[2], z], [z, z, z, o]])
    view = mat4_from_rows([[o, z, z, -look_from[0]], [z, o, z, -look_from[1]], [z, z, o, -look_from[2]], [z, z, z, o]])
    coef = f32(-1.0) / f32(5.0)
    proj = mat4_from_rows([[o, z, z, z], [z, o, z, z], [z, z, o, z], [z, z, coef, o]])
    w, h, dd = f32(width - 1), f32(height - 1), f32(255.0)
    two = f32(2.0)
    viewport = mat4_from_rows([[w / two, z, z, w / two], [z, h / two, z, h / two], [z, z, dd / two, dd / two], [z, z, z, o]])
    buf["vpmv"] = mat4_mul(mat4_mul(mat4_mul(viewport, proj), model), view)
    buf["m"] = model
    inv = mat4_inverse(mat4_transpose(model))
    if inv is None:
        raise Panic("it_m: try_inverse().unwrap()")
    buf["it_m"] = inv
    buf["camera_direction"] = new_z
    tl = mat_vec(model, [light[0], light[1], light[2], z])
    if tl[3] != z:
        raise Panic("Vector3::from_homogeneous: w != 0")
    buf["t_light"] = normalize(tl[:3])


def shadow_pass_prepare_1(buf, width, height, light, look_from, look_at, up):
    default_prepare(buf, width, height, light, light, look_at, up)
    buf["shadow_matrix"] = buf["vpmv"]


def shadow_pass_prepare_2(buf, width, height, light, look_from, look_at, up):
    default_prepare(buf, width, height, light, look_from, look_at, up)
    buf["i_vpmv"] = mat4_inverse(buf["vpmv"])
    buf["i_m"] = mat4_inverse(buf["m"])
    if buf["i_vpmv"] is None or buf["i_m"] is None:
        raise Panic("try_inverse().unwrap()")


# ---- the scene ------------------------------------------------------------------------------------------------------
PIPELINES = {   # shader.rs:100-109 -> passes of (prepare, vertex kind, fragment kind)
    "default": [(default_prepare, "default", "default")],
    "phong": [(default_prepare, "phong", "phong")],
    "normal_map": [(default_prepare, "plain", "normal_map")],
    "specular": [(default_prepare, "plain", "specular")],
    "darboux": [(default_prepare, "darboux", "darboux")],
    "shadow": [(shadow_pass_prepare_1, "depth", "depth"), (shadow_pass_prepare_2, "phong", "shadow2")],
    "occlusion": [(shadow_pass_prepare_1, "depth", "depth"), (shadow_pass_prepare_2, "plain", "occlusion2")],
}


class Scene:
    """scene.rs:25-269 with the per-pixel winning polygon kept beside the z buffer."""

    def __init__(self, width, height, mesh, textures, pipeline):
        self.W, self.H = int(width), int(height)
        self.pos = np.asarray(mesh["pos"], f32).reshape(-1, 3)
        self.tex = np.asarray(mesh["tex"], f32).reshape(-1, 3)
        self.nrm = np.asarray(mesh["nrm"], f32).reshape(-1, 3)
        self.idx = np.asarray(mesh["idx"], np.int64).reshape(-1, 9)
        self.texture, self.normal_map, self.normal_map_tangent, self.specular_map = (np.asarray(t, np.uint8) for t in textures)
        self.passes = PIPELINES[pipeline]
        n = self.W * self.H
        self.buf = {"z": np.zeros(n, f32), "shadow": np.zeros(n, f32)}     # Buffer::new: zeros (shader.rs:46-47)
        self.frame = np.zeros((n, 3), np.uint8)                             # scene.rs:71
        self.winner = np.full(n, 0xFFFFFFFF, np.uint32)
        self.light = [0.0, 0.0, -1.0]                                       # scene.rs:66-69
        self.look_from, self.look_at, self.up = [0.0, 0.0, 1.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0]

    def clear(self):                                                        # scene.rs:128-137
        self.buf["z"][:] = F32_MIN
        self.buf["shadow"][:] = F32_MIN
        self.frame[:] = 0
        self.winner[:] = 0xFFFFFFFF

    def set_light_direction(self, v):
        self.light = [float(c) for c in v]

    def set_camera(self, look_from, look_at, up):
        self.look_from, self.look_at, self.up = ([float(c) for c in v] for v in (look_from, look_at, up))

    def get_frame_buffer(self):                                             # scene.rs:92-97: flip_vertical_in_place
        return self.frame.reshape(self.H, self.W, 3)[::-1].copy()

    # -- util.rs:34-83: nearest texel, truncating coordinates, panics out of range
    def _texel(self, image, dims_of, u, v):
        cx = as_u32(u * f32(dims_of.shape[1]))
        cy = as_u32(v * f32(dims_of.shape[0]))
        if (cx >= image.shape[1]).any() or (cy >= image.shape[0]).any():
            raise Panic("get_pixel out of range")
        return image[cy, cx]

    def _normal_from(self, image, u, v):
        # (get_normal_tangent_at_uv takes the dimensions of normal_map, util.rs:62-63)
        px = self._texel(image, self.normal_map, u, v).astype(f32)
        return normalize([px[:, k] / f32(255.0) - f32(0.5) for k in range(3)])

    @staticmethod
    def _blend_black(color, t):                                             # util.rs:7-13 with color_2 = (0, 0, 0)
        c = color.astype(f32)
        rest = (f32(1.0) - t) * f32(0.0)
        return np.stack([as_u8(t * c[:, k] + rest) for k in range(3)], axis=1)

    # -- vertex closures, for ALL polygons at once (arrays of length n_polygons)
    def _vertex_stage(self, kind):
        b = self.buf
        ix = self.idx
        P = [[self.pos[ix[:, 3 * i], k] for k in range(3)] for i in range(3)]      # vertex_positions[i][k]
        keep = np.ones(len(ix), bool)
        out = {}
        if kind != "depth":
            e1 = [P[1][k] - P[0][k] for k in range(3)]
            e2 = [P[2][k] - P[0][k] for k in range(3)]
            face_normal = cross(e1, e2)
            keep = ~(dot3(b["camera_direction"], face_normal) <= f32(0.0))        # should_cull_face, shader.rs:116-124
            if kind == "default":
                tn = mat_vec(b["it_m"], face_normal + [f32(0.0)])
                tn = normalize(tn[:3])
                d = dot3(b["t_light"], tn)
                out["intensity"] = [d, d, d]
            if kind in ("phong", "darboux"):
                tnorm = []
                for i in range(3):
                    n_i = [self.nrm[ix[:, 3 * i + 2], k] for k in range(3)]
                    tnorm.append(normalize(mat_vec(b["it_m"], n_i + [f32(0.0)])[:3]))
                if kind == "phong":
                    out["intensity"] = [dot3(b["t_light"], tnorm[i]) for i in range(3)]
                else:
                    out["t_normals"] = tnorm                                      # columns
                    tp = []
                    for i in range(3):
                        q = mat_vec(b["m"], P[i] + [f32(1.0)])
                        if np.any(q[3] == f32(0.0)):
                            raise Panic("Point3::from_homogeneous: w == 0")
                        tp.append([q[k] / q[3] for k in range(3)])
                    out["t_positions"] = tp
        M = b["shadow_matrix"] if kind == "depth" else b["vpmv"]
        rx, ry, rz = [], [], []
        with np.errstate(all="ignore"):
            for i in range(3):                                                      # shader.rs:150-165
                q = mat_vec(M, P[i] + [f32(1.0)])
                if np.any((q[3] == f32(0.0)) & keep):
                    raise Panic("Point3::from_homogeneous: w == 0")
                rx.append(as_i32(q[0] / q[3]))
                ry.append(as_i32(q[1] / q[3]))
                rz.append(q[2] / q[3])
        out["rx"], out["ry"], out["z"] = rx, ry, rz
        out["u"] = [self.tex[ix[:, 3 * i + 1], 0] for i in range(3)]              # shader.rs:136-147
        out["v"] = [f32(1.0) - self.tex[ix[:, 3 * i + 1], 1] for i in range(3)]
        out["keep"] = keep
        return out

    # -- fragment closures for the accepted pixels of ONE polygon
    def _fragment_color(self, kind, V, t, bar, xs, ys, zval):
        b = self.buf
        W = self.W
        if kind in ("default", "phong", "normal_map", "specular", "darboux", "shadow2"):
            uu = (V["u"][0][t] * bar[0] + V["u"][1][t] * bar[1]) + V["u"][2][t] * bar[2]
            vv = (V["v"][0][t] * bar[0] + V["v"][1][t] * bar[1]) + V["v"][2][t] * bar[2]
            color = self._texel(self.texture, self.texture, uu, vv)
        if kind == "default":
            return self._blend_black(color, np.full(len(xs), V["intensity"][0][t], f32))
        if kind == "phong":
            return self._blend_black(color, dot3(bar, [V["intensity"][i][t] for i in range(3)]))
        if kind in ("normal_map", "specular"):
            n = self._normal_from(self.normal_map, uu, vv)
            tn = normalize(mat_vec(b["it_m"], n + [f32(0.0)])[:3])
            diff = dot3(b["t_light"], tn)
            if kind == "normal_map":
                return self._blend_black(color, diff)
            s = dot3(b["t_light"], tn)
            r = normalize([f32(2.0) * (tn[k] * s) - b["t_light"][k] for k in range(3)])
            exponent = self._texel(self.specular_map, self.specular_map, uu, vv)[:, 0].astype(f32)
            spec = f32(0.6) * _powf(np.fmax(r[2], f32(0.0)), exponent)
            c = color.astype(f32)
            return np.stack([as_u8(np.fmin((diff + spec) * c[:, k], f32(255.0))) for k in range(3)], axis=1)
        if kind == "darboux":
            nt = self._normal_from(self.normal_map_tangent, uu, vv)
            tn_cols = [[V["t_normals"][i][k][t] for k in range(3)] for i in range(3)]
            tp_cols = [[V["t_positions"][i][k][t] for k in range(3)] for i in range(3)]
            local_z = mat_vec(tn_cols, bar)
            o, z, m = f32(1.0), f32(0.0), f32(-1.0)
            row0 = normalize(mat_vec(tp_cols, [m, o, z]))
            row1 = normalize(mat_vec(tp_cols, [m, z, o]))
            row2 = normalize(mat_vec(tn_cols, bar))
            ones = np.ones(len(xs), f32)
            inv, singular = mat3_inverse([[row0[k] * ones for k in range(3)], [row1[k] * ones for k in range(3)], row2])
            if np.any(singular):
                raise Panic("local_basis_matrix.try_inverse().unwrap()")
            inv_cols = [[inv[r][c] for r in range(3)] for c in range(3)]
            du = [V["u"][1][t] - V["u"][0][t], V["u"][2][t] - V["u"][0][t], z]
            dv = [V["v"][1][t] - V["v"][0][t], V["v"][2][t] - V["v"][0][t], z]
            cols = [normalize(mat_vec(inv_cols, du)), normalize(mat_vec(inv_cols, dv)), normalize(local_z)]
            tn = normalize(mat_vec(cols, nt))
            return self._blend_black(color, dot3(b["t_light"], tn))
        # the two colour passes that look the shadow buffer up
        SI = mat4_mul(b["shadow_matrix"], b["i_vpmv"])
        p = [xs.astype(f32), ys.astype(f32), zval, np.ones(len(xs), f32)]

        def shadow_index(q):
            if np.any(q[3] == f32(0.0)):
                raise Panic("Point3::from_homogeneous: w == 0")
            sx, sy = _libm1("roundf", q[0] / q[3]), _libm1("roundf", q[1] / q[3])
            index = (as_u32(sx) + as_u32(sy) * W) % (1 << 32)      # u32 arithmetic (release build: wrapping)
            if np.any(index >= self.W * self.H):
                raise Panic("shadow_buffer index out of range")
            return index, q[2] / q[3]
        index, sz = shadow_index(mat_vec(SI, p))
        if kind == "shadow2":
            coef = np.where(sz + f32(1.0) < b["shadow"][index], f32(0.3), f32(1.0)).astype(f32)
            diff = dot3(bar, [V["intensity"][i][t] for i in range(3)])
            return self._blend_black(color, diff * coef)
        # occlusion, shader.rs:872-947
        tl = b["t_light"]
        ld = mat_vec(b["i_m"], [tl[0], tl[1], tl[2], f32(0.0)])
        if ld[3] != f32(0.0):
            raise Panic("Vector3::from_homogeneous: w != 0")
        wq = mat_vec(b["i_vpmv"], p)
        if np.any(wq[3] == f32(0.0)):
            raise Panic("Point3::from_homogeneous: w == 0")
        world = [wq[k] / wq[3] for k in range(3)]
        own = b["shadow"][index]
        rot = _rotation_between([f32(0.0), f32(0.0), f32(1.0)], ld[:3])
        occ = np.ones(len(xs), f32)
        angle_coef = (f32(2.0) * f32(np.pi)) / f32(16.0)
        for i in range(16):
            a = angle_coef * f32(i)
            g = [f32(_libm.sinf(float(a))), f32(0.0), f32(_libm.cosf(float(a)))]
            step = mat_vec(rot, g)
            sample = [world[k] + step[k] * f32(0.02) for k in range(3)]
            s_index, _ = shadow_index(mat_vec(b["shadow_matrix"], sample + [np.ones(len(xs), f32)]))
            val = b["shadow"][s_index]
            hit = val - f32(1.0) > own
            strength = np.fmin((val - own) / f32(20.0), f32(1.0))
            occ = np.where(hit, occ - (f32(1.0) / f32(16.0)) * strength, occ).astype(f32)
        white = np.full((len(xs), 3), 255, np.uint8)
        return self._blend_black(white, occ)

    def render(self):                                                       # scene.rs:151-268
        W, H = self.W, self.H
        b = self.buf
        for prepare, vkind, fkind in self.passes:
            prepare(b, W, H, self.light, self.look_from, self.look_at, self.up)
            V = self._vertex_stage(vkind)
            for t in np.nonzero(V["keep"])[0]:
                x = [int(V["rx"][i][t]) for i in range(3)]
                y = [int(V["ry"][i][t]) for i in range(3)]
                x_min, x_max = max(0, min(x)), min(max(x), W - 1)
                y_min, y_max = max(0, min(y)), min(max(y), H - 1)
                if x_min > x_max or y_min > y_max:
                    continue
                ii, jj = np.meshgrid(np.arange(x_min, x_max + 1), np.arange(y_min, y_max + 1), indexing="ij")
                ii, jj = ii.ravel(), jj.ravel()
                # to_barycentric_coord: integer differences, THEN the conversion to f32
                v1 = [f32(x[1] - x[0]), f32(x[2] - x[0]), (x[0] - ii).astype(f32)]
                v2 = [f32(y[1] - y[0]), f32(y[2] - y[0]), (y[0] - jj).astype(f32)]
                raw = cross(v1, v2)
                raw = [np.broadcast_to(np.asarray(c, f32), ii.shape) for c in raw]
                with np.errstate(all="ignore"):
                    degenerate = np.abs(raw[2]) < f32(1.0)
                    bar = [f32(1.0) - (raw[0] + raw[1]) / raw[2], raw[0] / raw[2], raw[1] / raw[2]]
                inside = ~degenerate & ~((bar[0] < 0) | (bar[1] < 0) | (bar[2] < 0))
                if not inside.any():
                    continue
                ii, jj = ii[inside], jj[inside]
                bar = [c[inside] for c in bar]
                index = ii + jj * W
                zval = dot3(bar, [V["z"][i][t] for i in range(3)])
                if fkind == "depth":                                        # shader.rs:694-709: returns false, draws nothing
                    upd = zval >= b["shadow"][index]
                    b["shadow"][index[upd]] = zval[upd]
                    continue
                accept = ~(zval <= b["z"][index])                           # process_z_value, shader.rs:169-180
                if not accept.any():
                    continue
                ii, jj, index, zval = ii[accept], jj[accept], index[accept], zval[accept]
                bar = [c[accept] for c in bar]
                b["z"][index] = zval
                self.winner[index] = t
                self.frame[index] = self._fragment_color(fkind, V, t, bar, ii, jj, zval)


def _rotation_between(a, b):
    """Rotation3::rotation_between(a, b).unwrap() as 3 columns (shader.rs:921): normalise both, axis = a x b, angle =
    acos(a . b), Rodrigues' formula; identity when the vectors are parallel, panic when they are antiparallel."""
    na_n, nb_n = np.sqrt(dot3(a, a)), np.sqrt(dot3(b, b))
    ident = [[f32(1), f32(0), f32(0)], [f32(0), f32(1), f32(0)], [f32(0), f32(0), f32(1)]]
    if not (na_n > 0 and nb_n > 0):
        return ident
    na, nb = [c / na_n for c in a], [c / nb_n for c in b]
    c = cross(na, nb)
    sq = dot3(c, c)
    eps = f32(np.finfo(np.float32).eps)
    if sq > eps * eps:
        cn = np.sqrt(sq)
        ux, uy, uz = (v / cn for v in c)
        angle = f32(_libm.acosf(float(dot3(na, nb)))) * f32(1.0)
        if angle == f32(0.0):
            return ident
        sn, cs = f32(_libm.sinf(float(angle))), f32(_libm.cosf(float(angle)))
        omc = f32(1.0) - cs
        sqx, sqy, sqz = ux * ux, uy * uy, uz * uz
        rows = [[sqx + (f32(1.0) - sqx) * cs, ux * uy * omc - uz * sn, ux * uz * omc + uy * sn],
                [ux * uy * omc + uz * sn, sqy + (f32(1.0) - sqy) * cs, uy * uz * omc - ux * sn],
                [ux * uz * omc - uy * sn, uy * uz * omc + ux * sn, sqz + (f32(1.0) - sqz) * cs]]
        return [[rows[r][col] for r in range(3)] for col in range(3)]
    if dot3(na, nb) < 0:
        raise Panic("rotation_between(..).unwrap() on antiparallel vectors")
    return ident
