// emul.cpp -- TEST-ONLY host emulation of the GPU algorithm (not part of the product library).
//
// Compiles the product's host/device stage functions (tiny_renderer_amd/csrc/tr_shaders.h,
// tr_prepare.cpp) with g++ and runs them through the same decomposition the kernels use:
// setup -> per-tile bins -> order-independent 64-bit max over (order(z), index) -> shade the
// survivors.  tests/test_emulation.py compares the result with the oracle on the CPU, so that
// arithmetic or algorithm mistakes are caught without a GPU.  It deliberately visits bin
// entries in REVERSE polygon order to prove the resolve does not depend on order.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "tiny_renderer.h"
#include "tr_prepare.h"
#include "tr_shaders.h"

using namespace tr;

namespace {

struct PassDesc {
    int prepare_kind, vs, fs;
};

struct Pipe {
    const char *name;
    int n;
    PassDesc p[2];
};

const Pipe kPipes[] = {
    { "default", 1, { { 0, VS_DEFAULT, FS_DEFAULT }, {} } },
    { "phong", 1, { { 0, VS_PHONG, FS_PHONG }, {} } },
    { "normal_map", 1, { { 0, VS_PLAIN, FS_NORMAL_MAP }, {} } },
    { "specular", 1, { { 0, VS_PLAIN, FS_SPECULAR }, {} } },
    { "darboux", 1, { { 0, VS_DARBOUX, FS_DARBOUX }, {} } },
    { "shadow", 2, { { 1, VS_DEPTH, FS_DEPTH }, { 2, VS_PHONG, FS_SHADOW2 } } },
    { "occlusion", 2, { { 1, VS_DEPTH, FS_DEPTH }, { 2, VS_PLAIN, FS_OCCLUSION2 } } },
};

template <int VS>
bool vs_call(const DevMesh &m, const DevUniforms &u, uint32_t t, RasterRec &r, float *v, uint32_t &e)
{
    return vertex_stage<VS>(m, u, t, r, v, e);
}

bool run_vs(int vs, const DevMesh &m, const DevUniforms &u, uint32_t t, RasterRec &r, float *v, uint32_t &e)
{
    switch (vs) {
    case VS_DEFAULT: return vs_call<VS_DEFAULT>(m, u, t, r, v, e);
    case VS_PHONG: return vs_call<VS_PHONG>(m, u, t, r, v, e);
    case VS_PLAIN: return vs_call<VS_PLAIN>(m, u, t, r, v, e);
    case VS_DARBOUX: return vs_call<VS_DARBOUX>(m, u, t, r, v, e);
    default: return vs_call<VS_DEPTH>(m, u, t, r, v, e);
    }
}

uint32_t run_fs(int fs, const DevUniforms &u, const DevTextures &tx, const float *vary, vec3 bar, uint32_t x,
                uint32_t y, float z, const float *shadow, uint32_t W, uint32_t H, uint32_t &err)
{
    switch (fs) {
    case FS_DEFAULT: return fragment_stage<FS_DEFAULT>(u, tx, vary, bar, x, y, z, shadow, W, H, err);
    case FS_PHONG: return fragment_stage<FS_PHONG>(u, tx, vary, bar, x, y, z, shadow, W, H, err);
    case FS_NORMAL_MAP: return fragment_stage<FS_NORMAL_MAP>(u, tx, vary, bar, x, y, z, shadow, W, H, err);
    case FS_SPECULAR: return fragment_stage<FS_SPECULAR>(u, tx, vary, bar, x, y, z, shadow, W, H, err);
    case FS_DARBOUX: return fragment_stage<FS_DARBOUX>(u, tx, vary, bar, x, y, z, shadow, W, H, err);
    case FS_SHADOW2: return fragment_stage<FS_SHADOW2>(u, tx, vary, bar, x, y, z, shadow, W, H, err);
    case FS_OCCLUSION2: return fragment_stage<FS_OCCLUSION2>(u, tx, vary, bar, x, y, z, shadow, W, H, err);
    default: return 0;
    }
}

}  // namespace

// Renders one frame the way the kernels do.  Buffers are in/out (the caller clears or keeps
// them): z / shadow W*H floats (internal layout), fb 3*W*H bytes with row 0 = TOP, winner W*H.
// `fresh` plays the role of the scene's lazy clear.  Returns the DevErr bits.
extern "C" uint32_t tr_emul_render(uint32_t W, uint32_t H, const tr_mesh *mesh, const tr_image_rgb8 tex[4],
                                   const char *pipeline, const float light[3], const float from[3],
                                   const float at[3], const float up[3], int fresh, float *zbuf, float *shadow,
                                   uint8_t *fb, uint32_t *winner, uint32_t band_row0, uint32_t band_row1)
{
    const Pipe *pipe = nullptr;
    for (const Pipe &p : kPipes)
        if (!strcmp(p.name, pipeline)) pipe = &p;
    if (!pipe) return 0x80000000u;

    std::vector<std::vector<uint32_t>> texel(4);
    DevTextures tx;
    for (int k = 0; k < 4; k++) {
        size_t n = (size_t)tex[k].w * tex[k].h;
        texel[k].resize(n);
        for (size_t i = 0; i < n; i++)
            texel[k][i] = tex[k].rgb[3 * i] | (tex[k].rgb[3 * i + 1] << 8) | (tex[k].rgb[3 * i + 2] << 16);
        tx.texel[k] = texel[k].data();
        tx.w[k] = tex[k].w;
        tx.h[k] = tex[k].h;
    }
    std::vector<float> rows((size_t)mesh->n_tri * TRI_FLOATS);
    for (uint32_t t = 0; t < mesh->n_tri; t++)
        gather_polygon(mesh->pos, mesh->tex, mesh->nrm, mesh->idx + 9u * (size_t)t, &rows[(size_t)t * TRI_FLOATS]);
    DevMesh dm = { rows.data(), mesh->n_tri };
    if (band_row0 == 0 && band_row1 == 0) band_row1 = H;

    tr_uniforms un;
    memset(&un, 0, sizeof un);
    uint32_t err = 0;
    for (int pi = 0; pi < pipe->n; pi++) {
        const PassDesc &p = pipe->p[pi];
        if (prepare_uniforms(p.prepare_kind, &un, W, H, light, from, at, up) != TR_OK) return 0x40000000u;
        DevUniforms du;
        memset(&du, 0, sizeof du);
        memcpy(du.vpmv, un.vpmv, 64);
        memcpy(du.m, un.m, 64);
        memcpy(du.it_m, un.it_m, 64);
        memcpy(du.shadow_matrix, un.shadow_matrix, 64);
        memcpy(du.i_vpmv, un.i_vpmv, 64);
        memcpy(du.camera_direction, un.camera_direction, 12);
        memcpy(du.t_light, un.t_light_direction, 12);
        if (p.fs == FS_SHADOW2 || p.fs == FS_OCCLUSION2) shadow_times_inverse(&un, du.sm_ivpmv);
        if (p.fs == FS_OCCLUSION2 && occlusion_steps(&un, du.occl_steps) != TR_OK) return 0x20000000u;

        const bool depth = p.fs == FS_DEPTH;
        DevFrame f;
        f.width = W;
        f.height = H;
        f.band_y0 = depth ? 0 : (int32_t)(H - band_row1);
        f.band_y1 = depth ? (int32_t)H : (int32_t)(H - band_row0);
        f.ntx = (W + TILE_W - 1) / TILE_W;
        f.ty_base = f.band_y0 / TILE_H;
        f.nty = (uint32_t)((f.band_y1 - 1) / TILE_H - f.ty_base + 1);

        // setup
        std::vector<RasterRec> rast(mesh->n_tri);
        std::vector<float> vary((size_t)mesh->n_tri * VARY_STRIDE, 0.0f);
        std::vector<std::vector<uint32_t>> bins((size_t)f.ntx * f.nty);
        for (uint32_t t = 0; t < mesh->n_tri; t++) {
            RasterRec &r = rast[t];
            if (run_vs(p.vs, dm, du, t, r, &vary[(size_t)t * VARY_STRIDE], err))
                finish_raster_rec(r, f);
            else
                mark_rejected(r);
            if (r.bx0 <= r.bx1)
                for (int32_t ty = r.by0 / TILE_H; ty <= r.by1 / TILE_H; ty++)
                    for (int32_t tx_ = r.bx0 / TILE_W; tx_ <= r.bx1 / TILE_W; tx_++)
                        bins[(size_t)(ty - f.ty_base) * f.ntx + tx_].push_back(t);
        }

        float *target = depth ? shadow : zbuf;
        struct Key {
            float z;
            uint32_t slot1;
        };
        std::vector<Key> key((size_t)TILE_W * TILE_H);
        for (uint32_t tile = 0; tile < f.ntx * f.nty; tile++) {
            const int32_t tile_x0 = (int32_t)(tile % f.ntx) * TILE_W;
            const int32_t tile_y0 = (f.ty_base + (int32_t)(tile / f.ntx)) * TILE_H;
            // initial keys
            for (int32_t j = 0; j < TILE_H; j++)
                for (int32_t i = 0; i < TILE_W; i++) {
                    const int32_t px = tile_x0 + i, py = tile_y0 + j;
                    Key k = { bits_f32(TR_F32_MIN_BITS), 0u };
                    if (!fresh && px < (int32_t)W && py >= f.band_y0 && py < f.band_y1)
                        k.z = target[(size_t)py * W + px];
                    key[(size_t)j * TILE_W + i] = k;
                }
            // coverage; the bin is visited in REVERSE polygon order on purpose
            std::vector<uint32_t> bin(bins[tile].rbegin(), bins[tile].rend());
            for (size_t bi = 0; bi < bin.size(); bi++) {
                const uint32_t tri = bin[bi];
                const RasterRec &r = rast[tri];
                const int32_t bx0 = imax(r.bx0, tile_x0), bx1 = imin(r.bx1, tile_x0 + TILE_W - 1);
                const int32_t by0 = imax(r.by0, tile_y0), by1 = imin(r.by1, tile_y0 + TILE_H - 1);
                const Edge e = edge_setup(r);
                const Recip rz = recip_of(e.cz);
                for (int32_t py = by0; py <= by1; py++)
                    for (int32_t px = bx0; px <= bx1; px++) {
                        float cx, cy;
                        edge_cross(e, px, py, cx, cy);
                        if (!covers(cx, cy, e.cz)) continue;
                        const vec3 bar = barycentric_by(cx, cy, rz);
                        const float z = dot3(bar, make3(r.z0, r.z1, r.z2));
                        Key &cur = key[(size_t)(py - tile_y0) * TILE_W + (px - tile_x0)];
                        bool win = z > cur.z;
                        if (z == cur.z) {
                            if (cur.slot1 == 0u)
                                win = depth;
                            else
                                win = depth ? tri > bin[cur.slot1 - 1u] : tri < bin[cur.slot1 - 1u];
                        }
                        if (win) {
                            cur.z = z;
                            cur.slot1 = (uint32_t)bi + 1u;
                        }
                    }
            }
            // shade + write
            for (int32_t j = 0; j < TILE_H; j++)
                for (int32_t i = 0; i < TILE_W; i++) {
                    const int32_t px = tile_x0 + i, py = tile_y0 + j;
                    if (!(px < (int32_t)W && py >= f.band_y0 && py < f.band_y1)) continue;
                    const uint32_t s1 = key[(size_t)j * TILE_W + i].slot1;
                    const bool won = s1 != 0u;
                    if (!won && !fresh) continue;
                    float zout = bits_f32(TR_F32_MIN_BITS);
                    uint32_t rgb = 0, tri = 0xFFFFFFFFu;
                    if (won) {
                        tri = bin[s1 - 1u];
                        const RasterRec &r = rast[tri];
                        const Edge e = edge_setup(r);
                        float cx, cy;
                        edge_cross(e, px, py, cx, cy);
                        const vec3 bar = barycentric(cx, cy, e.cz);
                        zout = dot3(bar, make3(r.z0, r.z1, r.z2));
                        if (!depth)
                            rgb = run_fs(p.fs, du, tx, &vary[(size_t)tri * VARY_STRIDE], bar, (uint32_t)px,
                                         (uint32_t)py, zout, shadow, W, H, err);
                    }
                    target[(size_t)py * W + px] = zout;
                    if (!depth) {
                        uint8_t *o = fb + ((size_t)(H - 1 - py) * W + px) * 3;
                        o[0] = rgb & 0xFF;
                        o[1] = (rgb >> 8) & 0xFF;
                        o[2] = (rgb >> 16) & 0xFF;
                        if (winner) winner[(size_t)py * W + px] = tri;
                    }
                }
        }
        // a colour pass consumed the cleared state of its targets only
    }
    return err;
}

// Direct taps for the coverage-arithmetic test: the product's division-free inside test and
// its barycentric coordinates for one pixel.
extern "C" int tr_emul_covers(const int32_t raster[6], int32_t px, int32_t py, float bar_out[3])
{
    RasterRec r;
    memset(&r, 0, sizeof r);
    r.x0 = raster[0]; r.y0 = raster[1]; r.x1 = raster[2]; r.y1 = raster[3]; r.x2 = raster[4]; r.y2 = raster[5];
    const Edge e = edge_setup(r);
    if (fabsf(e.cz) < 1.0f) return -1;  // degenerate: never drawn
    float cx, cy;
    edge_cross(e, px, py, cx, cy);
    const vec3 b = barycentric(cx, cy, e.cz);
    bar_out[0] = b.x; bar_out[1] = b.y; bar_out[2] = b.z;
    // bit 1: the tile kernel's orientation-normalised form of the same test
    Edge n = e;
    if (n.cz < 0.0f) {
        n.a0 = -n.a0; n.a1 = -n.a1; n.b0 = -n.b0; n.b1 = -n.b1;
        n.cz = -n.cz;
    }
    float nx, ny;
    edge_cross(n, px, py, nx, ny);
    return (covers(cx, cy, e.cz) ? 1 : 0) | (covers_oriented(nx, ny, n.cz) ? 2 : 0);
}

extern "C" uint32_t tr_emul_depth_order_key(float z) { return depth_order_key(z); }

// the specular closure's powf (tr_powf.h) for n argument pairs; returns TR_POWF_EXACT
extern "C" int tr_emul_powf(const float *x, const float *y, float *out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) out[i] = tr_powf(x[i], y[i]);
    return TR_POWF_EXACT;
}

// x / d through the shared-reciprocal division of tr_math.h
extern "C" float tr_emul_div_by(float x, float d) { return div_by(x, recip_of(d)); }
