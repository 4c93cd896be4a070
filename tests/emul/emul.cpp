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
#include <stdlib.h>
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

uint32_t run_fs_color(int fs, const DevUniforms &u, const DevTextures &tx, const float *vary, vec3 bar, float uu, float vv,
                      uint32_t x, uint32_t y, float z, const float *shadow, uint32_t W, uint32_t H, uint32_t &err)
{
    switch (fs) {
    case FS_DEFAULT: return fragment_color<FS_DEFAULT>(u, tx, vary, bar, uu, vv, x, y, z, shadow, W, H, err);
    case FS_PHONG: return fragment_color<FS_PHONG>(u, tx, vary, bar, uu, vv, x, y, z, shadow, W, H, err);
    case FS_NORMAL_MAP: return fragment_color<FS_NORMAL_MAP>(u, tx, vary, bar, uu, vv, x, y, z, shadow, W, H, err);
    case FS_SPECULAR: return fragment_color<FS_SPECULAR>(u, tx, vary, bar, uu, vv, x, y, z, shadow, W, H, err);
    case FS_DARBOUX: return fragment_color<FS_DARBOUX>(u, tx, vary, bar, uu, vv, x, y, z, shadow, W, H, err);
    case FS_SHADOW2: return fragment_color<FS_SHADOW2>(u, tx, vary, bar, uu, vv, x, y, z, shadow, W, H, err);
    case FS_OCCLUSION2: return fragment_color<FS_OCCLUSION2>(u, tx, vary, bar, uu, vv, x, y, z, shadow, W, H, err);
    default: return 0;
    }
}

struct VaryPair {
    const float *a, *b;
    f2 operator()(int k) const { return mk2(a[k], b[k]); }
};

uint64_t g_pair_fast = 0, g_pair_plain = 0;
uint64_t g_mask_violations = 0, g_mask_cells_live = 0, g_mask_cells_box = 0;  // pair_masks: covered pixels in dead cells; culling rate  // how often the two-pixel closures' guard let the fast form stand

// The kernel's use of the two-pixel closures: both pixels together; if either survivor left the guarded
// range (`plain` stays true) the caller runs the plain closure for both.
template <int FS>
void fs_pair(const DevUniforms &u, const DevTextures &tx, const float *va, const float *vb, const Bary2 &bar, f2 uu, f2 vv,
             uint32_t &ca, uint32_t &cb, uint32_t &ea, uint32_t &eb, const bool won[2], bool &plain)
{
    VaryPair v = { va, vb };
    vec3p barp;
    barp.x = bar.x; barp.y = bar.y; barp.z = bar.z;
    bool bad_a, bad_b;
    fragment_color_pair<FS>(u, tx, v, barp, uu, vv, ca, cb, ea, eb, bad_a, bad_b);
    plain = (bad_a && won[0]) || (bad_b && won[1]);
    (plain ? g_pair_plain : g_pair_fast)++;
}

bool run_fs_pair(int fs, const DevUniforms &u, const DevTextures &tx, const float *va, const float *vb, const Bary2 &bar,
                 f2 uu, f2 vv, uint32_t &ca, uint32_t &cb, uint32_t &ea, uint32_t &eb, const bool won[2], bool &plain)
{
    if (!has_pair_closure(fs)) return false;
    switch (fs) {
    case FS_NORMAL_MAP: fs_pair<FS_NORMAL_MAP>(u, tx, va, vb, bar, uu, vv, ca, cb, ea, eb, won, plain); break;
    case FS_SPECULAR: fs_pair<FS_SPECULAR>(u, tx, va, vb, bar, uu, vv, ca, cb, ea, eb, won, plain); break;
    default: fs_pair<FS_DARBOUX>(u, tx, va, vb, bar, uu, vv, ca, cb, ea, eb, won, plain); break;
    }
    return true;
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
    // the closure's images as one interleaved, tiled array, as the scene builds it (tr_texels.h) -- TR_EMUL_PLAIN_TEXELS=1:
    // the plain images only
    std::vector<uint32_t> packed;
    {
        bool same = true;
        for (int k = 1; k < 4; k++) same = same && tex[k].w == tex[0].w && tex[k].h == tex[0].h;
        const char *plain = getenv("TR_EMUL_PLAIN_TEXELS");
        if (same && !(plain && atoi(plain))) {
            const uint32_t *const image[4] = { texel[0].data(), texel[1].data(), texel[2].data(), texel[3].data() };
            packed = pack_texels(pipe->p[pipe->n - 1].fs, image, tex[0].w, tex[0].h, tx.packed_bpr);
            tx.packed = packed.data();
        }
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
                // the tile kernel's forms: orientation-normalised edge constants, two pixels per step in
                // packed arithmetic, the three-way-minimum inside test, compare-only barycentrics
                const Edge e = edge_setup(r);
                float la0 = e.a0, la1 = e.a1, lb0 = e.b0, lb1 = e.b1, lcz = la0 * lb1 - la1 * lb0, lry = record_recip(r);
                if (lcz < 0.0f) {
                    la0 = -la0; la1 = -la1; lb0 = -lb0; lb1 = -lb1;
                    lcz = -lcz;
                    lry = -lry;
                }
                Edge2 e2;
                e2.a0 = splat2(la0); e2.a1 = splat2(la1); e2.b0 = splat2(lb0); e2.b1 = splat2(lb1);
                e2.cz = splat2(lcz);
                e2.y = splat2(lry);
                // what k_setup stores with this (polygon, tile) pair: the cells / block columns that can hold a
                // fragment.  Pixels outside them are skipped, as the tile kernel skips them; a covered pixel
                // found there is a violation of the masks' conservativeness (counted, and the frame is wrong).
                uint32_t mlo = 0, mhi = 0, blo = 0, bhi = 0;
                pair_masks(r, tile_x0, tile_y0, true, mlo, mhi);    // a pass whose tile kernel runs the shared form
                pair_masks(r, tile_x0, tile_y0, false, blo, bhi);   // ... owns columns: block columns for every pair
                const PairBox pb = pair_box(r.bx0, r.bx1, r.by0, r.by1, tile_x0, tile_y0);
                const bool small_pair = pb.nch <= SCAN_MAX_CHUNKS;
                const uint32_t cols = pair_block_columns(mlo, mhi, true, pb, tile_x0) & pair_block_columns(blo, bhi, false, pb, tile_x0);
                if (small_pair) {
                    g_mask_cells_live += (uint64_t)(__builtin_popcount(mlo) + __builtin_popcount(mhi));
                    g_mask_cells_box += (uint64_t)(pb.nch * (pb.ay1 - pb.ay0 + 1));
                }
                for (int32_t py = by0; py <= by1; py++)
                    for (int32_t px = bx0; px <= bx1; px += 2) {
                        const bool second = px + 1 <= bx1;  // the pair's second pixel (px + 1, py)
                        f2 cx, cy;
                        edge_cross2(e2, mk2((float)isub(r.x0, px), (float)isub(r.x0, px + 1)), splat2((float)isub(r.y0, py)), cx, cy);
                        const f2 rest = e2.cz - (cx + cy);
                        bool hit[2] = { covers_oriented(cx.x, cy.x, lcz), second && covers_oriented(cx.y, cy.y, lcz) };
                        (void)rest;
                        for (int h = 0; h < 2; h++) {
                            const int32_t x = px + h;
                            const uint32_t cell = (uint32_t)((py - pb.ay0) * 4 + ((x - pb.xs) >> 3));
                            const bool in_cell = !small_pair || (((cell < 32u ? mlo >> cell : mhi >> (cell - 32u)) & 1u) != 0u);
                            const bool in_col = ((cols >> ((x - tile_x0) >> 3)) & 1u) != 0u;
                            if (hit[h] && !(in_cell && in_col)) g_mask_violations++;
                            hit[h] = hit[h] && in_cell && in_col;
                        }
                        if (!hit[0] && !hit[1]) continue;
                        const Bary2 bar = barycentric2_for_compare(cx, cy, e2);
                        const f2 z2 = dot3_2(bar.x, bar.y, bar.z, splat2(r.z0), splat2(r.z1), splat2(r.z2));
                        for (int h = 0; h < 2; h++) {
                            if (!hit[h]) continue;
                            const float z = h ? z2.y : z2.x;
                            Key &cur = key[(size_t)(py - tile_y0) * TILE_W + (px + h - tile_x0)];
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
            }
            // shade + write: two pixels at a time ((i, j) and (i + 1, j)), each against its own survivor,
            // through the kernel's packed forms -- barycentric2, the two-pixel closures with their
            // range guard and the plain closure as fallback
            for (int32_t j = 0; j < TILE_H; j++)
                for (int32_t i = 0; i < TILE_W; i += 2) {
                    const int32_t py = tile_y0 + j;
                    bool live[2], won[2];
                    uint32_t tri[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu }, rgb[2] = { 0u, 0u };
                    float zout[2] = { bits_f32(TR_F32_MIN_BITS), bits_f32(TR_F32_MIN_BITS) };
                    const RasterRec *rr[2];
                    for (int h = 0; h < 2; h++) {
                        const int32_t px = tile_x0 + i + h;
                        live[h] = px < (int32_t)W && py >= f.band_y0 && py < f.band_y1;
                        const uint32_t s1 = live[h] ? key[(size_t)j * TILE_W + i + h].slot1 : 0u;
                        won[h] = s1 != 0u;
                        tri[h] = won[h] ? bin[s1 - 1u] : 0xFFFFFFFFu;
                        rr[h] = &rast[won[h] ? tri[h] : (bin.empty() ? 0u : bin[0])];  // lanes without a survivor run on some record
                    }
                    if ((won[0] || won[1]) && !rast.empty()) {
                        const Edge ea = edge_setup(*rr[0]), eb = edge_setup(*rr[1]);
                        Edge2 e2;
                        e2.a0 = mk2(ea.a0, eb.a0); e2.a1 = mk2(ea.a1, eb.a1); e2.b0 = mk2(ea.b0, eb.b0); e2.b1 = mk2(ea.b1, eb.b1);
                        e2.cz = e2.a0 * e2.b1 - e2.a1 * e2.b0;
                        e2.y = mk2(record_recip(*rr[0]), record_recip(*rr[1]));
                        const int32_t pxa = tile_x0 + i, pxb = pxa + 1;
                        f2 cx, cy;
                        edge_cross2(e2, mk2((float)isub(rr[0]->x0, pxa), (float)isub(rr[1]->x0, pxb)),
                                    mk2((float)isub(rr[0]->y0, py), (float)isub(rr[1]->y0, py)), cx, cy);
                        const Bary2 bar = barycentric2(cx, cy, e2);
                        const f2 z = dot3_2(bar.x, bar.y, bar.z, mk2(rr[0]->z0, rr[1]->z0), mk2(rr[0]->z1, rr[1]->z1),
                                            mk2(rr[0]->z2, rr[1]->z2));
                        const float *va = &vary[(size_t)(won[0] ? tri[0] : 0u) * VARY_STRIDE];
                        const float *vb = &vary[(size_t)(won[1] ? tri[1] : 0u) * VARY_STRIDE];
                        uint32_t ca = 0, cb = 0, ea_ = 0, eb_ = 0;
                        if (!depth) {
                            f2 uu = mk2(va[0], vb[0]) * bar.x, vv = mk2(va[1], vb[1]) * bar.x;
                            uu = mk2(va[2], vb[2]) * bar.y + uu;
                            vv = mk2(va[3], vb[3]) * bar.y + vv;
                            uu = mk2(va[4], vb[4]) * bar.z + uu;
                            vv = mk2(va[5], vb[5]) * bar.z + vv;
                            bool plain = true;
                            if (run_fs_pair(p.fs, du, tx, va, vb, bar, uu, vv, ca, cb, ea_, eb_, won, plain)) {}
                            if (plain) {
                                ea_ = eb_ = 0;
                                if (won[0])
                                    ca = run_fs_color(p.fs, du, tx, va, make3(bar.x.x, bar.y.x, bar.z.x), uu.x, vv.x,
                                                      (uint32_t)pxa, (uint32_t)py, z.x, shadow, W, H, ea_);
                                if (won[1])
                                    cb = run_fs_color(p.fs, du, tx, vb, make3(bar.x.y, bar.y.y, bar.z.y), uu.y, vv.y,
                                                      (uint32_t)pxb, (uint32_t)py, z.y, shadow, W, H, eb_);
                            }
                        }
                        if (won[0]) { zout[0] = z.x; rgb[0] = ca; err |= ea_; }
                        if (won[1]) { zout[1] = z.y; rgb[1] = cb; err |= eb_; }
                    }
                    for (int h = 0; h < 2; h++) {
                        const int32_t px = tile_x0 + i + h;
                        if (!live[h] || (!won[h] && !fresh)) continue;
                        target[(size_t)py * W + px] = zout[h];
                        if (!depth) {
                            uint8_t *o = fb + ((size_t)(H - 1 - py) * W + px) * 3;
                            o[0] = rgb[h] & 0xFF;
                            o[1] = (rgb[h] >> 8) & 0xFF;
                            o[2] = (rgb[h] >> 16) & 0xFF;
                            if (winner) winner[(size_t)py * W + px] = won[h] ? tri[h] : 0xFFFFFFFFu;
                        }
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
    // the shading phase's barycentrics: exact quotients through the record's reciprocal
    Edge2 e2;
    e2.a0 = splat2(e.a0); e2.a1 = splat2(e.a1); e2.b0 = splat2(e.b0); e2.b1 = splat2(e.b1);
    e2.cz = splat2(e.cz);
    e2.y = splat2(record_recip(r));
    f2 cx, cy;
    edge_cross2(e2, splat2((float)isub(r.x0, px)), splat2((float)isub(r.y0, py)), cx, cy);
    const Bary2 b = barycentric2(cx, cy, e2);
    bar_out[0] = b.x.x; bar_out[1] = b.y.x; bar_out[2] = b.z.y;
    // the coverage phase's inside test on the orientation-normalised polygon
    Edge2 n = e2;
    float ncz = e.cz;
    if (ncz < 0.0f) {
        n.a0 = -n.a0; n.a1 = -n.a1; n.b0 = -n.b0; n.b1 = -n.b1;
        ncz = -ncz;
    }
    f2 nx, ny;
    edge_cross2(n, splat2((float)isub(r.x0, px)), splat2((float)isub(r.y0, py)), nx, ny);
    return covers_oriented(nx.x, ny.y, ncz) ? 3 : 0;
}

// the specular closure's powf (tr_powf.h) for n argument pairs; returns TR_POWF_EXACT
extern "C" int tr_emul_powf(const float *x, const float *y, float *out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) out[i] = tr_powf(x[i], y[i]);
    return TR_POWF_EXACT;
}

// x / d through the shared-reciprocal forms: which = 0 div_by (coverage / barycentrics: keeps the sign of
// a zero numerator), 1 div_by2_nonzero with y = rcp2(d) (the two-pixel closures)
extern "C" void tr_emul_div(int which, const float *x, const float *d, float *out, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++)
        out[i] = which == 0 ? div_by(x[i], recip_of(d[i]))
                            : div_by2_nonzero(mk2(x[i], x[i]), mk2(d[i], d[i]), rcp2(mk2(d[i], d[i]))).y;
}

// color_blend(c, black, t) per channel: the product's form (tr_math.h blend_black: the (1 - t) * 0.0 term as a select)
// and the literal one; returns the number of (t, c) pairs -- every channel value 0..255 -- whose bytes differ
extern "C" uint64_t tr_emul_blend_mismatches(const float *t, uint64_t n)
{
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; i++)
        for (uint32_t c = 0; c < 256u; c++) bad += blend_black(c, t[i]) != blend_black_literal(c, t[i]);
    return bad;
}
extern "C" uint32_t tr_emul_blend(uint32_t c, float t, int literal) { return literal ? blend_black_literal(c, t) : blend_black(c, t); }

// decode_normal for two texels at once against the plain form; returns the number of differing components
// shadow_fetch with fast-clear flags against the plain lookup in a fully materialised buffer: n lookups at
// (x[i], y[i]); returns how many differ (value bits or error bits).
extern "C" uint32_t tr_emul_shadow_fetch_mismatches(const float *plain, const float *stale, const uint32_t *sclean,
                                                    uint32_t W, uint32_t H, const float *x, const float *y, uint32_t n)
{
    uint32_t bad = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t e0 = 0, e1 = 0;
        const float a = shadow_fetch(plain, nullptr, W, H, make3(x[i], y[i], 0.0f), e0);
        const float b = shadow_fetch(stale, sclean, W, H, make3(x[i], y[i], 0.0f), e1);
        if (f32_bits(a) != f32_bits(b) || e0 != e1) bad++;
    }
    return bad;
}

extern "C" uint32_t tr_emul_decode_normal_mismatches(uint32_t first, uint32_t count)
{
    uint32_t bad = 0;
    for (uint32_t t = first; t < first + count; t++) {
        const uint32_t other = (t * 2654435761u) & 0xFFFFFFu;
        const vec3p p = decode_normal_p(t, other);
        const vec3 a = decode_normal(t), b = decode_normal(other);
        bad += memcmp(&p.x.x, &a.x, 4) != 0 || memcmp(&p.y.x, &a.y, 4) != 0 || memcmp(&p.z.x, &a.z, 4) != 0 ||
               memcmp(&p.x.y, &b.x, 4) != 0 || memcmp(&p.y.y, &b.y, 4) != 0 || memcmp(&p.z.y, &b.z, 4) != 0;
    }
    return bad;
}

// pair_masks statistics since the library was loaded: {covered pixels found outside the masks (must be 0),
// live cells of small pairs, cells of their boxes}
extern "C" void tr_emul_mask_counts(uint64_t out[3])
{
    out[0] = g_mask_violations;
    out[1] = g_mask_cells_live;
    out[2] = g_mask_cells_box;
}

extern "C" void tr_emul_pair_counts(uint64_t out[2])
{
    out[0] = g_pair_fast;
    out[1] = g_pair_plain;
}

// ---------------------------------------------------------------------------------------------------------------------
// The host's planner (tiny_renderer_amd/csrc/tr_plan.h: the decisions of tr_scene.cpp, free of HIP), bound for
// tests/test_planner.py.
// ---------------------------------------------------------------------------------------------------------------------
#include "tr_plan.h"

extern "C" {

// shape: n_tiles, frames_per_launch, max_slots, forced, winner_tap, tile_stamps, no_long_runs, n_passes, then two u64:
// pool_bytes_per_pass, pixels.  out: group_size, long_run_group_size
void tr_emul_plan_sizes(const uint32_t *shape, const uint64_t *big, uint32_t *out)
{
    tr::plan::SceneShape s;
    s.n_tiles = shape[0]; s.frames_per_launch = shape[1]; s.max_slots = shape[2]; s.forced = shape[3];
    s.winner_tap = shape[4] != 0; s.tile_stamps = shape[5] != 0; s.no_long_runs = shape[6] != 0; s.n_passes = shape[7];
    s.pool_bytes_per_pass = big[0]; s.pixels = big[1];
    out[0] = tr::plan::group_size(s);
    out[1] = tr::plan::long_run_group_size(s);
}

// returns the number of groups; sizes[] (cap entries), info = {slots, set_frames, kept}
uint32_t tr_emul_plan_call(uint32_t n, uint32_t G, uint32_t long_run, int automatic, uint32_t growth, uint32_t short_factor,
                           uint32_t *sizes, uint32_t cap, uint32_t *info)
{
    const tr::plan::CallPlan p = tr::plan::plan_call(n, G, long_run, automatic != 0, growth, short_factor);
    for (size_t k = 0; k < p.sizes.size() && k < cap; k++) sizes[k] = p.sizes[k];
    info[0] = p.slots; info[1] = p.set_frames; info[2] = p.kept;
    return (uint32_t)p.sizes.size();
}

// fbs: g buffer ids (0 = the current slot's own buffer, 1..15 = other buffers of the scene's, >= 16 = callers' buffers);
// out: per frame {slot, fb id or 0xFFFFFFFF for "the slot's own", unreplayable}
void tr_emul_plan_deferred(const uint32_t *fbs, uint32_t g, int cur_slot, uint32_t *out)
{
    std::vector<const void *> wanted(g);
    for (uint32_t j = 0; j < g; j++) wanted[j] = (const void *)(uintptr_t)(fbs[j] + 1u);
    const std::vector<tr::plan::DeferredTarget> t = tr::plan::plan_deferred(
        wanted, cur_slot, (const void *)(uintptr_t)1u, [](const void *fb) { return (uintptr_t)fb <= 16u; });
    for (uint32_t j = 0; j < g; j++) {
        out[3 * j + 0] = (uint32_t)t[j].slot;
        out[3 * j + 1] = t[j].fb ? (uint32_t)((uintptr_t)t[j].fb - 1u) : 0xFFFFFFFFu;
        out[3 * j + 2] = t[j].unreplayable ? 1u : 0u;
    }
}

int tr_emul_plan_overflow(uint64_t first_bad, uint64_t observed, uint64_t unreplayable, int last_was_group, int last_valid,
                          int last_started_cleared)
{
    tr::plan::OverflowState o = { first_bad, observed, unreplayable, last_was_group != 0, last_valid != 0, last_started_cleared != 0 };
    return (int)tr::plan::overflow_action(o);
}

uint64_t tr_emul_plan_grown_pool(uint64_t cap, uint64_t need) { return tr::plan::grown_pool(cap, need); }

int tr_emul_plan_handover(uint32_t pending, int nothing_submitted, int newest_tile_done)
{
    return (int)tr::plan::handover(pending, nothing_submitted != 0, newest_tile_done != 0);
}

uint32_t tr_emul_plan_grid_units(const uint32_t *lengths, uint32_t frames, uint32_t n_tiles)
{
    return tr::plan::group_grid_units(lengths, frames, n_tiles);
}
uint32_t tr_emul_plan_work_units(const uint32_t *lengths) { return tr::plan::work_units(lengths); }

void tr_emul_plan_constants(int *out)
{
    out[0] = tr::plan::GROUP_MAX; out[1] = tr::plan::GROUP_SETS; out[2] = tr::plan::LOOKAHEAD; out[3] = tr::plan::BATCH;
}

}  // extern "C"
