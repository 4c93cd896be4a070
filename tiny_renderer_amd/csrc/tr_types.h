// tr_types.h -- plain-data structures shared by the host code and the HIP kernels.
#pragma once

#include <stdint.h>

namespace tr {

// shader.rs:100-109
enum Pipeline : int {
    P_DEFAULT = 0,
    P_PHONG,
    P_NORMAL_MAP,
    P_SPECULAR,
    P_DARBOUX,
    P_SHADOW,
    P_OCCLUSION,
    P_COUNT
};

// Vertex-stage variants (the `vertex` closures of shader.rs).
enum VsKind : int {
    VS_DEFAULT = 0,  // shader.rs:285-316   cull + face-normal intensity
    VS_PHONG,        // shader.rs:349-384, 711-747   cull + per-vertex intensities
    VS_PLAIN,        // shader.rs:416-437, 475-496, 849-870   cull + transform + uv
    VS_DARBOUX,      // shader.rs:549-595
    VS_DEPTH,        // shader.rs:671-692, 809-830   no cull, shadow_matrix
    VS_COUNT
};

// Fragment-stage variants (the `fragment` closures of shader.rs).
enum FsKind : int {
    FS_DEFAULT = 0,  // shader.rs:318-333
    FS_PHONG,        // shader.rs:386-401
    FS_NORMAL_MAP,   // shader.rs:439-459
    FS_SPECULAR,     // shader.rs:498-534
    FS_DARBOUX,      // shader.rs:597-655
    FS_SHADOW2,      // shader.rs:749-788
    FS_OCCLUSION2,   // shader.rs:872-947
    FS_DEPTH,        // shader.rs:694-709, 832-847 (shadow-buffer fill, draws nothing)
    FS_LIT,          // the colour of the frame's lit texel image at (u, v) (k_lit: normal-map / specular closures once per texel)
    FS_COUNT
};

// Screen tile owned by one 256-thread workgroup: 128 x 16 pixels, four 32 x 16 quadrants (one
// per wavefront), each quadrant eight 8 x 8 lane blocks.  128 px of rgb8 = 384 B = three
// 128-byte lines, 128 px of f32 depth = 512 B = four lines: every row a tile writes is made of
// whole cache lines.
constexpr int TILE_W = 128;
#ifndef TR_TILE_H
#define TR_TILE_H 16
#endif
constexpr int TILE_H = TR_TILE_H;
// Wavefronts per tile workgroup: 4, 8 or 16, chosen per launch (launch_tile).  Each owns a
// TILE_W / waves pixel wide column of the tile during coverage; for shading the tile is re-divided
// into 32-pixel wide strips so that every row a wave stores is whole cache lines.  Four waves do
// the least total work and win when the busy tiles fill the machine (4096^2 and up); more waves
// shorten the serial work per wave and win when they do not (2048^2: 42 -> 27 us with sixteen).
constexpr int STRIP = 32;

// Raster part of a polygon record (64 B).  Mirrors Buffer.vertex_t_raster / vertex_z_values
// (shader.rs:34-35) plus the clamped bounding box of scene.rs:233-239.  bx0 > bx1 marks a
// polygon that draws nothing (culled, off screen, or degenerate: |cross.z| < 1,
// scene.rs:188-191).
struct RasterRec {
    int32_t bx0, bx1, by0, by1;  // 16-byte piece 0: clamped bounding box
    int32_t x0, y0, x1, y1;      // piece 1
    int32_t x2, y2;              // piece 2
    float z0, z1;
    float z2;                    // piece 3 (continued by the first two varyings in a bin record)
    uint32_t id;                 //   polygon index
    uint32_t pad[2];
};
static_assert(sizeof(RasterRec) == 64, "RasterRec must be 64 bytes");

// Varyings, up to 24 floats per polygon:
//   [0..5]   vertex_uvs            u0,v0,u1,v1,u2,v2        (shader.rs:33)
//   [6..8]   vertex_intensities                              (shader.rs:30)      default/phong/shadow
//   darboux instead keeps, from vertex_t_positions / vertex_t_normals (shader.rs:31-32):
//   [6..8]   normalize(tpos * (-1,1,0))    row 0 of the local basis (shader.rs:612-617)
//   [9..11]  normalize(tpos * (-1,0,1))    row 1                    (shader.rs:618-623)
//   [12..20] vertex_t_normals, column major
constexpr int VARY_STRIDE = 24;

// A bin record = everything a tile needs to know about a polygon, as 16-byte pieces:
//   0: bx0 bx1 by0 by1 | 1: x0 y0 x1 y1 | 2: x2 y2 z0 z1 | 3: z2 id v0 v1 | 4: v2..v5 | 5: v6..v9
//   darboux also: 6: v10..v13 | 7: v14..v17 | 8: v18..v21
struct alignas(16) Piece {
    uint32_t x, y, z, w;
};
constexpr int REC_PIECES_SMALL = 6;  //  96 B
constexpr int REC_PIECES_LARGE = 9;  // 144 B
// Records a tile keeps resident in LDS; larger bins take the chunked path, whose survivors' records
// come from global memory when they are shaded.  Four waves per tile: 8 KiB (85 / 56 records) --
// with the 16 KiB of keys a workgroup stays under 25 KiB and six fit in a CU's LDS.  The layouts
// with more waves per tile are for frames whose tiles cannot fill the GPU; a CU holds three (8
// waves) or two (16 waves) of their workgroups by wave count alone, so LDS is plentiful and the
// heaviest tiles -- the critical path of a small frame -- stay resident (256 / 426 records).
#ifndef TR_LDS_REC4
#define TR_LDS_REC4 8192
#endif
constexpr int lds_rec_bytes(int tile_waves) { return tile_waves == 4 ? TR_LDS_REC4 : tile_waves == 8 ? 24576 : 40960; }

// Frame constants the kernels need, computed on the host by the prepares (shader.rs:183-279).
struct DevUniforms {
    float vpmv[16];
    float m[16];
    float it_m[16];
    float shadow_matrix[16];
    float sm_ivpmv[16];  // shadow_matrix * i_vpmv (shader.rs:763-764, 899-900)
    float i_vpmv[16];
    float camera_direction[3];
    float t_light[3];
    float occl_steps[48];  // rot * (sin a_k, 0, cos a_k) * 0.02, k = 0..15 (shader.rs:916-929)
};

struct DevTextures {
    const uint32_t *texel[4];  // rgba8 (a = 0), row 0 = top; texture, normal_map,
                               // normal_map_tangent, specular_map
    uint32_t w[4], h[4];
    // The images the scene's colour closure fetches, interleaved texel by texel and tiled into 128-byte blocks
    // (fetch_texels, tr_shaders.h); null: the closure fetches each image on its own (images of different sizes).
    const uint32_t *packed = nullptr;
    uint32_t packed_bpr = 0;  // blocks per row of blocks
};

// The model as the vertex stage reads it: one 96-byte row per polygon, gathered from
// obj::raw::RawObj's indexed arrays once at scene creation (the mesh never changes), so that the
// per-frame vertex stage is a single coalesced sweep instead of index -> attribute chains:
//   [0..8] positions p0 p1 p2 | [9..17] normals n0 n1 n2 | [18..23] tex coords u0 v0 u1 v1 u2 v2
constexpr int TRI_FLOATS = 24;
struct DevMesh {
    const float *tri;  // n_tri * TRI_FLOATS
    uint32_t n_tri;
};

// Fills one row of DevMesh::tri from the indexed arrays (util.rs:25-31, shader.rs:136-147,363-367).
inline void gather_polygon(const float *pos, const float *tex, const float *nrm, const uint32_t *ix, float *out)
{
    for (int i = 0; i < 3; i++) {
        for (int k = 0; k < 3; k++) {
            out[3 * i + k] = pos[3 * ix[3 * i + 0] + k];
            out[9 + 3 * i + k] = nrm[3 * ix[3 * i + 2] + k];
        }
        out[18 + 2 * i + 0] = tex[3 * ix[3 * i + 1] + 0];
        out[18 + 2 * i + 1] = tex[3 * ix[3 * i + 1] + 1];
    }
}

// Geometry of the rendered region.
struct DevFrame {
    uint32_t width, height;
    int32_t band_y0, band_y1;  // internal rows [y0, y1) (row 0 = bottom) this scene owns
    uint32_t ntx, nty;         // tile grid covering the band
    int32_t ty_base;           // first tile row (internal y / TILE_H)
};

// Device error word bits (reported as TR_E_OOB_LOOKUP / TR_E_BIN_OVERFLOW).
enum DevErr : uint32_t {
    DE_W_ZERO = 1u << 0,
    DE_TEX_OOB = 1u << 1,
    DE_SHADOW_OOB = 1u << 2,
    DE_SINGULAR = 1u << 3,
    DE_BIN_OVERFLOW = 1u << 4
};

// Triangle bins: a pass has a POOL of `pool_cap` records; k_order gives tile t the records
// [offset, offset + count) of it (WorkItem) -- exactly as many as k_setup counted.  Should the pairs of a
// whole pass exceed the pool, DE_BIN_OVERFLOW is raised and bin_need records how many the pass wanted, so that the
// host can grow the pools and render the frame again.
// One entry of the tile kernel's work lists (k_order).
struct alignas(16) WorkItem {
    uint32_t tile, count;
    uint32_t offset;  // the tile's first record in the pool
    uint32_t pad;
};

constexpr uint32_t TR_STORE_DEPTH = 1u, TR_STORE_COLOR = 2u;

struct SetupArgs {
    DevMesh mesh;
    DevFrame frame;
    DevUniforms u;
    uint32_t *tile_count;   // n_tiles counters, then 16 words (k_order's list lengths of a per-frame launch, the pool cursor)
    Piece *recs;            // n_tri records of rec_pieces x 16 B: every polygon's record, once (k_setup -> k_bin)
    Piece *bins;            // the pool: pool_cap records
    uint32_t pool_cap;
    uint32_t rec_pieces;
    uint32_t *err;
    uint32_t *alarm;    // page-locked host word (mapped): set to 1 with whatever is raised in `err` (see TileArgs)
    // 1: the tile kernel resolves small pairs as scan-line items (shared form): they get cell masks; 0: every pair
    // gets the block columns (pair_masks, tr_shaders.h)
    uint32_t cells;
    // the frame's lit texel image (k_lit; null: the closure runs per fragment): colour of every texel under this frame's
    // light and camera, from the scene's texel set `texel_set` (tex_w x tex_h texels, set_bpr / lit_bpr blocks per row)
    uint32_t *lit;
    const uint32_t *texel_set;
    uint32_t tex_w, tex_h, set_bpr, lit_bpr;
    // Fused launches: where k_bin -- the chain's last kernel -- leaves the pass's eight list lengths for the HOST
    // (page-locked, mapped), which sizes LATER tile kernels' grids by them (k_tile, "work units"; tr_scene.cpp,
    // group_units).  len_src: the lengths in the tile kernel's table entry (k_order's); len_host null: nothing is reported.
    const uint32_t *len_src;
    uint32_t *len_host;
};

struct TileArgs {
    const Piece *bins;   // the pass's pool
    uint32_t pool_cap;
    uint32_t rec_pieces;
    // The pass's work lists (k_order, from this pass's counters): eight regions of n_tiles entries, region b =
    // the tiles of weight class b as (tile, polygons in its bin), the last region the empty tiles.
    const WorkItem *order;
    // This pass's counter set: n_tiles counters (k_order leaves the end of the tile's pool range in each, k_bin counts
    // it down to the start, the tile's k_tile workgroup zeroes it for the set's next pass), then 16 words, the first
    // eight of which hold the lists' lengths of a per-frame launch, the next the 64-bit pool cursor.  A fused launch's lengths are in its table entry
    // (list_len, zeroed by the host, filled by k_order): the tile kernel finds them with its other arguments
    // instead of behind one more dependent load.
    uint32_t *tile_count;
    uint32_t list_len[8];
    // k_order: where it writes each tile's first record (k_bin reads them), and the overflow bookkeeping -- how many
    // records the pass wanted when the pool was too small, the smallest `pass_seq` of a pass that overflowed
    uint32_t *bin_need;
    unsigned long long *overflow_seq;
    unsigned long long pass_seq;
    DevFrame frame;
    DevUniforms u;
    DevTextures tex;
    float *zbuf;        // W*H, internal layout (row 0 = bottom)
    float *shadow;      // W*H, internal layout
    uint8_t *fb;        // 3*W*H, row 0 = top (already flipped: scene.rs:92-97 folded in)
    uint32_t *winner;   // W*H or nullptr
    uint32_t *err;
    // A word in page-locked HOST memory, set (plain store) whenever a bit is raised in `err`: the host looks at it
    // after waiting for the stream and copies the device words only when it is up -- a frame without errors costs
    // no device-to-host copy at its sync (10 us of the 13 an empty tr_scene_sync took).
    uint32_t *alarm;
    uint32_t fresh;     // 1: target buffers are logically cleared (scene.rs:128-137 folded in)
    // What a COLOUR pass writes to memory (TR_STORE_*; depth passes always write their shadow buffer).  Both: the
    // reference's pass.  Colour only: "transient depth" -- the z of a cleared frame is resolved in LDS and left there,
    // as a tile-based GPU leaves a depth attachment nobody loads on the chip; should a getter or an accumulating
    // render want the z buffer after all, the same pass is repeated with depth only (tr_scene.cpp, ensure_depth).
    uint32_t store;
    uint32_t aligned16; // 1: width % 16 == 0, cleared rows can be written in 16-byte pieces
    uint32_t aligned4;  // 1: width % 4 == 0, colour rows can be written as packed dwords
    uint64_t *stamps;   // diagnostic only (TR_OPT_TILE_STAMPS): per tile {start, end, polygons, hw id}; else nullptr
    // Fast depth clear: zclean[t] != 0 says every z of colour-pass tile t is logically f32::MIN and
    // its memory is stale (null for depth passes, whose shadow buffer is looked up at random).
    uint32_t *zclean;
    // Colour counterpart: fbclean[t] != 0 says every colour byte (and winner word) of tile t already holds
    // the cleared value in memory -- the tile was empty when it was last written -- so an empty tile of
    // a cleared frame has nothing to store.  Null for depth passes.
    uint32_t *fbclean;
    // The shadow buffer's fast-clear flags (one per tile of the WHOLE frame), for the colour passes that look
    // the shadow buffer up (shadow, occlusion); the depth pass that fills the buffer has them as its `zclean`.
    const uint32_t *sclean;
};

}  // namespace tr
