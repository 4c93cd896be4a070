// tr_kernels.hip -- gfx950 kernels of the triangle-fill path.
//
// The reference (src/scene.rs:151-268) is one serial loop nest: pass x polygon x bbox-x x
// bbox-y, depth-testing and shading each covered pixel in polygon order.  Here a render pass is
//
//   k_setup   one thread per polygon: vertex closure, clamped bounding box, then the wavefront
//             spreads its (polygon, tile) pairs over all 64 lanes and appends polygon ids to
//             fixed-capacity per-tile bins (one atomic per pair)
//   k_tile    one 256-thread workgroup per 128x32 screen tile: the bin's records are staged in
//             LDS, coverage + depth resolve run against LDS keys, then the winners are shaded and
//             depth and colour are streamed out once
//
// Equivalence with the serial loop: `z <= zbuf -> reject` in polygon order means the surviving
// fragment of a pixel is the one with the largest z, ties going to the lowest polygon index, and
// the frame buffer keeps the colour of the last accepted fragment = that survivor
// (shader.rs:169-180, scene.rs:259-263).  A 64-bit max over (order(z), ~index) computes the same
// survivor in any order, so only survivors are shaded.  The depth-only passes use `>=`
// (shader.rs:703): largest z, ties to the highest index.
//
// No MFMA anywhere: the path is compare/gather/stream work bound by HBM writes.
#include <hip/hip_runtime.h>

#include "tr_kernels.h"
#include "tr_shaders.h"

namespace tr {

namespace {

constexpr int NBX = QUAD / 8;    // 8x8 lane blocks per quadrant row
constexpr int NBY = TILE_H / 8;  // block rows per quadrant
constexpr int QPIX = QUAD * TILE_H;
constexpr uint32_t NO_WINNER = 0xFFFFFFFFu;
constexpr int SHADE_G = 2;  // row pairs shaded together (memory-level parallelism vs registers)
constexpr int CHUNK = 64;  // polygons staged in LDS per round (256 threads x 16 B)

__device__ __forceinline__ int32_t tile_index(const DevFrame &f, int32_t tx, int32_t ty)
{
    return (ty - f.ty_base) * (int32_t)f.ntx + tx;
}

// -----------------------------------------------------------------------------------------
// k_setup
// -----------------------------------------------------------------------------------------
template <int VS>
__global__ __launch_bounds__(256) void k_setup(SetupArgs a)
{
    // per-wave table for the pair distribution: exclusive pair offsets and tile ranges
    __shared__ int32_t s_excl[4][64], s_tx0[4][64], s_ty0[4][64], s_ntx[4][64];
    __shared__ uint32_t s_tri[4][64];

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const bool active = t < a.mesh.n_tri;

    int32_t tx0 = 0, ty0 = 0, ntx = 0, cnt = 0;
    uint32_t err = 0;
    if (active) {
        RasterRec r;
        float vary[VARY_STRIDE];
#pragma unroll
        for (int i = 0; i < VARY_STRIDE; i++) vary[i] = 0.0f;
        const bool keep = vertex_stage<VS>(a.mesh, a.u, t, r, vary, err);
        if (keep)
            finish_raster_rec(r, a.frame);
        else
            mark_rejected(r);
        r.id = t;

        uint4 *dst = reinterpret_cast<uint4 *>(a.rast + t);
        const uint4 *src = reinterpret_cast<const uint4 *>(&r);
        dst[0] = src[0];
        dst[1] = src[1];
        dst[2] = src[2];
        dst[3] = src[3];
        if (r.bx0 <= r.bx1) {
            float4 *vd = reinterpret_cast<float4 *>(a.vary + (size_t)t * VARY_STRIDE);
#pragma unroll
            for (int i = 0; i < VARY_STRIDE / 4; i++)
                vd[i] = make_float4(vary[4 * i], vary[4 * i + 1], vary[4 * i + 2], vary[4 * i + 3]);
            tx0 = r.bx0 / TILE_W;
            ty0 = r.by0 / TILE_H;
            ntx = r.bx1 / TILE_W - tx0 + 1;
            cnt = ntx * (r.by1 / TILE_H - ty0 + 1);
        }
    }
    if (err) atomicOr(a.err, err);

    // Binning.  One lane per polygon would serialise a polygon's atomics (a polygon that spans
    // 30 tiles = 30 dependent round trips); instead the wave's pairs are numbered by a prefix
    // sum and dealt round-robin to the lanes, so every lane issues ceil(pairs/64) atomics.
    int32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t v = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += v;
    }
    const int32_t total = __shfl(incl, 63, 64);
    s_excl[wave][lane] = incl - cnt;
    s_tx0[wave][lane] = tx0;
    s_ty0[wave][lane] = ty0;
    s_ntx[wave][lane] = ntx > 0 ? ntx : 1;
    s_tri[wave][lane] = t;
    __syncthreads();
    for (int32_t p = (int32_t)lane; p < total; p += 64) {
        // owner = last lane whose exclusive offset is <= p (lanes without pairs share their
        // successor's offset, so "last" skips them)
        int32_t lo = 0, hi = 63;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (s_excl[wave][mid] <= p)
                lo = mid;
            else
                hi = mid - 1;
        }
        const int32_t q = p - s_excl[wave][lo];
        const int32_t w = s_ntx[wave][lo];
        const int32_t tile = tile_index(a.frame, s_tx0[wave][lo] + q % w, s_ty0[wave][lo] + q / w);
        const uint32_t slot = atomicAdd(&a.tile_count[tile], 1u);
        if (slot < a.bin_cap) {
            a.bins[(size_t)tile * a.bin_cap + slot] = s_tri[wave][lo];
        } else {
            atomicOr(a.err, (uint32_t)DE_BIN_OVERFLOW);
            atomicMax(a.bin_need, slot + 1u);
        }
    }
}

// -----------------------------------------------------------------------------------------
// k_tile
// -----------------------------------------------------------------------------------------

// LDS index of pixel (qx, qy) of a quadrant: block-major, so that during coverage lane l of a
// wave touches slot (block*64 + l): conflict-free 8-byte accesses.
__device__ __forceinline__ uint32_t key_slot(uint32_t qx, uint32_t qy)
{
    return (((qy >> 3) * NBX + (qx >> 3)) << 6) + ((qy & 7u) << 3) + (qx & 7u);
}

// Streams the cleared value of a tile (scene.rs:128-137 folded into the render): z / shadow =
// f32::MIN, rgb = 0.  Whole-tile rows are whole cache lines; 16 B per lane when width % 16 == 0.
template <bool DEPTH>
__device__ __forceinline__ void write_cleared_tile(const TileArgs &a, int32_t tile_x0, int32_t tile_y0)
{
    const int32_t W = (int32_t)a.frame.width, H = (int32_t)a.frame.height;
    const uint32_t tid = threadIdx.x;
    float *depth = DEPTH ? a.shadow : a.zbuf;
    if (a.aligned16) {
        const uint4 zmin = make_uint4(TR_F32_MIN_BITS, TR_F32_MIN_BITS, TR_F32_MIN_BITS, TR_F32_MIN_BITS);
        // depth: TILE_H rows x 32 pieces of 16 B
        for (uint32_t c = tid; c < (uint32_t)TILE_H * 32u; c += 256u) {
            const int32_t y = tile_y0 + (int32_t)(c >> 5), x = tile_x0 + (int32_t)(c & 31u) * 4;
            if (x < W && y >= a.frame.band_y0 && y < a.frame.band_y1)
                *reinterpret_cast<uint4 *>(depth + (size_t)y * W + x) = zmin;
        }
        if (!DEPTH) {
            const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
            // colour: TILE_H rows x 24 pieces of 16 B
            for (uint32_t c = tid; c < (uint32_t)TILE_H * 24u; c += 256u) {
                const int32_t y = tile_y0 + (int32_t)(c / 24u);
                const int32_t xb = tile_x0 * 3 + (int32_t)(c % 24u) * 16;
                if (xb < W * 3 && y >= a.frame.band_y0 && y < a.frame.band_y1)
                    *reinterpret_cast<uint4 *>(a.fb + (size_t)(H - 1 - y) * W * 3 + xb) = zero;
            }
            if (a.winner) {
                const uint4 none = make_uint4(NO_WINNER, NO_WINNER, NO_WINNER, NO_WINNER);
                for (uint32_t c = tid; c < (uint32_t)TILE_H * 32u; c += 256u) {
                    const int32_t y = tile_y0 + (int32_t)(c >> 5), x = tile_x0 + (int32_t)(c & 31u) * 4;
                    if (x < W && y >= a.frame.band_y0 && y < a.frame.band_y1)
                        *reinterpret_cast<uint4 *>(a.winner + (size_t)y * W + x) = none;
                }
            }
        }
    } else {
        for (uint32_t p = tid; p < (uint32_t)(TILE_W * TILE_H); p += 256u) {
            const int32_t y = tile_y0 + (int32_t)(p / TILE_W), x = tile_x0 + (int32_t)(p % TILE_W);
            if (x < W && y >= a.frame.band_y0 && y < a.frame.band_y1) {
                depth[(size_t)y * W + x] = bits_f32(TR_F32_MIN_BITS);
                if (!DEPTH) {
                    uint8_t *px = a.fb + ((size_t)(H - 1 - y) * W + x) * 3;
                    px[0] = 0;
                    px[1] = 0;
                    px[2] = 0;
                    if (a.winner) a.winner[(size_t)y * W + x] = NO_WINNER;
                }
            }
        }
    }
}

template <int FS>
__global__ __launch_bounds__(256) void k_tile(TileArgs a)
{
    constexpr bool DEPTH = (FS == FS_DEPTH);
    // low word of the key: colour passes prefer the LOWEST polygon index on equal z
    // (0xFFFFFFFE - index; 0xFFFFFFFF = "what was there before", which wins every tie: the
    // reference rejects z <= zbuf); depth passes prefer the HIGHEST (index + 1; 0 = before).
    constexpr uint32_t PRIOR = DEPTH ? 0u : 0xFFFFFFFFu;

    __shared__ __attribute__((aligned(16))) uint64_t s_key[TILE_W * TILE_H];
    __shared__ uint4 s_rec[CHUNK * 4];  // CHUNK staged RasterRecs

    const uint32_t tile = blockIdx.x;
    const int32_t tx = (int32_t)(tile % a.frame.ntx);
    const int32_t ty = a.frame.ty_base + (int32_t)(tile / a.frame.ntx);
    const int32_t tile_x0 = tx * TILE_W, tile_y0 = ty * TILE_H;
    const int32_t W = (int32_t)a.frame.width, H = (int32_t)a.frame.height;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;

    uint32_t n = a.tile_count[tile];
    if (n > a.bin_cap) n = a.bin_cap;  // overflow: flagged by k_setup, the host renders again

    if (n == 0u && a.fresh) {
        write_cleared_tile<DEPTH>(a, tile_x0, tile_y0);
        return;
    }

    float *depth = DEPTH ? a.shadow : a.zbuf;
    const int32_t qx0 = tile_x0 + (int32_t)wave * QUAD, qy0 = tile_y0;
    uint64_t *wkey = s_key + wave * QPIX;
    const int32_t lx = (int32_t)(lane & 7u), ly = (int32_t)(lane >> 3);

    // ---- initial keys -------------------------------------------------------------------
    {
        const uint64_t cleared = ((uint64_t)depth_order_key(bits_f32(TR_F32_MIN_BITS)) << 32) | PRIOR;
#pragma unroll
        for (int b = 0; b < NBX * NBY; b++) {
            uint64_t key = cleared;
            if (!a.fresh) {
                const int32_t px = qx0 + (b % NBX) * 8 + lx, py = qy0 + (b / NBX) * 8 + ly;
                if (px < W && py >= a.frame.band_y0 && py < a.frame.band_y1)
                    key = ((uint64_t)depth_order_key(depth[(size_t)py * W + px]) << 32) | PRIOR;
            }
            wkey[(b << 6) + lane] = key;
        }
    }

    // ---- coverage + depth resolve ---------------------------------------------------------
    // The bin is consumed in chunks of CHUNK polygons.  All 256 threads stage a chunk's records
    // into LDS (thread t fetches 16-byte piece t%4 of polygon t/4: two dependent global loads
    // for the whole chunk instead of two per polygon per wave); each wave then ballots which
    // staged polygons touch its quadrant and walks only those, reading the record back from LDS
    // at a wave-uniform address.
    const uint32_t *bin = a.bins + (size_t)tile * a.bin_cap;
    for (uint32_t c0 = 0; c0 < n; c0 += CHUNK) {
        const uint32_t m = min((uint32_t)CHUNK, n - c0);
        if (c0 != 0u) __syncthreads();  // every wave is done with the previous chunk
        {
            const uint32_t j = tid >> 2, piece = tid & 3u;
            if (j < m) {
                const uint32_t id = bin[c0 + j];
                s_rec[j * 4u + piece] = reinterpret_cast<const uint4 *>(a.rast + id)[piece];
            }
        }
        __syncthreads();

        bool touch = false;
        if (lane < m) {
            const uint4 box = s_rec[lane * 4u];  // bx0 bx1 by0 by1
            touch = imax((int32_t)box.x, qx0) <= imin((int32_t)box.y, qx0 + QUAD - 1) &&
                    imax((int32_t)box.z, qy0) <= imin((int32_t)box.w, qy0 + TILE_H - 1);
        }
        unsigned long long todo = __ballot(touch);
        while (todo) {
            const uint32_t j = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint4 p0 = s_rec[j * 4u + 0u], p1 = s_rec[j * 4u + 1u], p2 = s_rec[j * 4u + 2u];
            const uint4 p3 = s_rec[j * 4u + 3u];
            RasterRec r;
            r.x0 = (int32_t)p1.x; r.y0 = (int32_t)p1.y; r.x1 = (int32_t)p1.z; r.y1 = (int32_t)p1.w;
            r.x2 = (int32_t)p2.x; r.y2 = (int32_t)p2.y;
            r.z0 = __uint_as_float(p2.z); r.z1 = __uint_as_float(p2.w); r.z2 = __uint_as_float(p3.x);
            const uint32_t tri = p3.y;
            const int32_t bx0 = imax((int32_t)p0.x, qx0), bx1 = imin((int32_t)p0.y, qx0 + QUAD - 1);
            const int32_t by0 = imax((int32_t)p0.z, qy0), by1 = imin((int32_t)p0.w, qy0 + TILE_H - 1);
            const Edge e = edge_setup(r);
            const uint32_t low = DEPTH ? tri + 1u : 0xFFFFFFFEu - tri;
            const int32_t ib0 = (bx0 - qx0) >> 3, ib1 = (bx1 - qx0) >> 3;
            const int32_t jb0 = (by0 - qy0) >> 3, jb1 = (by1 - qy0) >> 3;
            for (int32_t jb = jb0; jb <= jb1; jb++) {
                for (int32_t ib = ib0; ib <= ib1; ib++) {
                    const int32_t px = qx0 + ib * 8 + lx, py = qy0 + jb * 8 + ly;
                    float cx, cy;
                    edge_cross(e, px, py, cx, cy);
                    if (px >= bx0 && px <= bx1 && py >= by0 && py <= by1 && covers(cx, cy, e.cz)) {
                        const vec3 bar = barycentric(cx, cy, e.cz);
                        const float z = dot3(bar, make3(r.z0, r.z1, r.z2));
                        const uint64_t key = ((uint64_t)depth_order_key(z) << 32) | low;
                        uint64_t *slot = wkey + (((jb * NBX + ib) << 6) + (int32_t)lane);
                        if (key > *slot) *slot = key;
                    }
                }
            }
        }
    }

    // ---- shade the survivors and stream the tile out ------------------------------------------
    // Lanes are row-major here (lane = x within a 32-pixel row, two rows per wave step), so depth
    // is stored straight from registers as whole 128-byte lines and colour is packed to dwords
    // with two lane permutes.  The vertical flip of get_frame_buffer (scene.rs:92-97) is folded
    // into the colour address.  SHADE_G steps are processed together and branch-free (lanes
    // without a survivor run the same loads on polygon 0 and discard the result) so that the
    // dependent gathers key -> record -> varyings -> texel of different rows overlap.
    const int32_t hx = (int32_t)(lane & 31u), hrow = (int32_t)(lane >> 5);
    const uint32_t half_base = lane & 32u;
#pragma unroll 1
    for (int32_t g = 0; g < TILE_H / 2; g += SHADE_G) {
        int32_t py[SHADE_G];
        bool live[SHADE_G], won[SHADE_G];
        uint32_t tri[SHADE_G], rgb[SHADE_G];
        float zout[SHADE_G];
        const int32_t px = qx0 + hx;
        bool any_won = false;
#pragma unroll
        for (int u = 0; u < SHADE_G; u++) {
            const int32_t qy = (g + u) * 2 + hrow;
            py[u] = qy0 + qy;
            live[u] = px < W && py[u] >= a.frame.band_y0 && py[u] < a.frame.band_y1;
            const uint32_t low = (uint32_t)wkey[key_slot((uint32_t)hx, (uint32_t)qy)];
            won[u] = live[u] && low != PRIOR;
            tri[u] = won[u] ? (DEPTH ? low - 1u : 0xFFFFFFFEu - low) : NO_WINNER;
            rgb[u] = 0u;
            zout[u] = bits_f32(TR_F32_MIN_BITS);
            any_won = any_won || won[u];
        }
        if (__any(any_won)) {
            uint32_t err = 0u;
#pragma unroll
            for (int u = 0; u < SHADE_G; u++) {
                const uint32_t ts = won[u] ? tri[u] : 0u;
                const uint4 *rr = reinterpret_cast<const uint4 *>(a.rast + ts);
                const uint4 v0 = rr[1];  // x0 y0 x1 y1
                const uint4 v1 = rr[2];  // x2 y2 z0 z1
                RasterRec r;
                r.x0 = (int32_t)v0.x; r.y0 = (int32_t)v0.y; r.x1 = (int32_t)v0.z; r.y1 = (int32_t)v0.w;
                r.x2 = (int32_t)v1.x; r.y2 = (int32_t)v1.y;
                r.z0 = __uint_as_float(v1.z); r.z1 = __uint_as_float(v1.w); r.z2 = a.rast[ts].z2;
                const Edge e = edge_setup(r);
                float cx, cy;
                edge_cross(e, px, py[u], cx, cy);
                const vec3 bar = barycentric(cx, cy, e.cz);
                const float z = dot3(bar, make3(r.z0, r.z1, r.z2));
                uint32_t c = 0u, e1 = 0u;
                if (!DEPTH) {
                    const float4 *vp = reinterpret_cast<const float4 *>(a.vary + (size_t)ts * VARY_STRIDE);
                    float vary[VARY_STRIDE];
                    constexpr int NV = (FS == FS_DARBOUX) ? 6 : 3;
#pragma unroll
                    for (int i = 0; i < NV; i++) {
                        const float4 q = vp[i];
                        vary[4 * i] = q.x; vary[4 * i + 1] = q.y; vary[4 * i + 2] = q.z; vary[4 * i + 3] = q.w;
                    }
                    c = fragment_stage<FS>(a.u, a.tex, vary, bar, (uint32_t)px, (uint32_t)py[u], z, a.shadow,
                                           (uint32_t)W, (uint32_t)H, e1);
                }
                if (won[u]) {
                    zout[u] = z;
                    rgb[u] = c;
                    err |= e1;
                }
            }
            if (err) atomicOr(a.err, err);
        }

#pragma unroll
        for (int u = 0; u < SHADE_G; u++) {
            if (!DEPTH && !a.fresh && live[u] && !won[u]) {
                // untouched pixel of an accumulate render: its colour may share a dword with a
                // touched neighbour, so fetch it
                const uint8_t *old = a.fb + ((size_t)(H - 1 - py[u]) * W + px) * 3;
                rgb[u] = pack_rgb(old[0], old[1], old[2]);
            }
            // depth: only pixels that changed (or every live pixel of a fresh tile)
            const bool put = live[u] && (won[u] || a.fresh);
            if (put) depth[(size_t)py[u] * W + px] = zout[u];
            if (!DEPTH) {
                if (a.winner && put) a.winner[(size_t)py[u] * W + px] = tri[u];
                uint8_t *row = a.fb + ((size_t)(H - 1 - py[u]) * W + qx0) * 3;
                if (a.aligned4) {
                    // dword j of the 96-byte row = bytes 4j..4j+3 = pixel p0 = 4j/3 from byte (4j)%3
                    // on, topped up from pixel p0+1
                    const uint32_t j = (uint32_t)hx;
                    const uint32_t p0 = (4u * j) / 3u, o = (4u * j) % 3u;
                    const uint32_t c0 = (uint32_t)__shfl((int)rgb[u], (int)(half_base + (p0 & 31u)), 64);
                    const uint32_t c1 = (uint32_t)__shfl((int)rgb[u], (int)(half_base + ((p0 + 1u) & 31u)), 64);
                    const uint32_t dw = (c0 >> (8u * o)) | (c1 << (24u - 8u * o));
                    const bool row_live = py[u] >= a.frame.band_y0 && py[u] < a.frame.band_y1;
                    if (j < 24u && row_live && (qx0 * 3 + (int32_t)(4u * j)) < W * 3)
                        *reinterpret_cast<uint32_t *>(row + 4u * j) = dw;
                } else if (put) {
                    uint8_t *p = row + 3 * hx;
                    p[0] = (uint8_t)(rgb[u] & 0xFFu);
                    p[1] = (uint8_t)((rgb[u] >> 8) & 0xFFu);
                    p[2] = (uint8_t)((rgb[u] >> 16) & 0xFFu);
                }
            }
        }
    }

    // the bin is consumed: leave the counter at zero for the next pass / frame
    __syncthreads();
    if (tid == 0u) a.tile_count[tile] = 0u;
}

// -----------------------------------------------------------------------------------------
// Small utility kernels
// -----------------------------------------------------------------------------------------

// Scene::clear materialised (scene.rs:128-137) for the cases the render cannot fold it in.
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t *dst, uint32_t value, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = value;
}

// get_z_buffer / get_shadow_buffer (scene.rs:101-125): `v as u8` replicated to rgb, flipped.
__global__ __launch_bounds__(256) void k_depth_view(const float *src, uint8_t *dst, uint32_t W, uint32_t H)
{
    const size_t n = (size_t)W * H;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t y = (uint32_t)(i / W), x = (uint32_t)(i % W);
        const uint8_t v = (uint8_t)f32_to_u8(src[i]);
        uint8_t *p = dst + ((size_t)(H - 1u - y) * W + x) * 3;
        p[0] = v;
        p[1] = v;
        p[2] = v;
    }
}

}  // namespace

// -----------------------------------------------------------------------------------------
// Launchers
// -----------------------------------------------------------------------------------------

#define TR_LAUNCH_CHECK()                    \
    do {                                     \
        hipError_t e_ = hipGetLastError();   \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

int launch_setup(int vs, const SetupArgs &a, hipStream_t st)
{
    if (a.mesh.n_tri == 0) return 0;
    const dim3 grid((a.mesh.n_tri + 255u) / 256u), block(256);
    switch (vs) {
    case VS_DEFAULT: hipLaunchKernelGGL(k_setup<VS_DEFAULT>, grid, block, 0, st, a); break;
    case VS_PHONG: hipLaunchKernelGGL(k_setup<VS_PHONG>, grid, block, 0, st, a); break;
    case VS_PLAIN: hipLaunchKernelGGL(k_setup<VS_PLAIN>, grid, block, 0, st, a); break;
    case VS_DARBOUX: hipLaunchKernelGGL(k_setup<VS_DARBOUX>, grid, block, 0, st, a); break;
    case VS_DEPTH: hipLaunchKernelGGL(k_setup<VS_DEPTH>, grid, block, 0, st, a); break;
    default: return -1;
    }
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_tile(int fs, const TileArgs &a, hipStream_t st)
{
    const uint32_t n_tiles = a.frame.ntx * a.frame.nty;
    if (n_tiles == 0) return 0;
    const dim3 grid(n_tiles), block(256);
    switch (fs) {
    case FS_DEFAULT: hipLaunchKernelGGL(k_tile<FS_DEFAULT>, grid, block, 0, st, a); break;
    case FS_PHONG: hipLaunchKernelGGL(k_tile<FS_PHONG>, grid, block, 0, st, a); break;
    case FS_NORMAL_MAP: hipLaunchKernelGGL(k_tile<FS_NORMAL_MAP>, grid, block, 0, st, a); break;
    case FS_SPECULAR: hipLaunchKernelGGL(k_tile<FS_SPECULAR>, grid, block, 0, st, a); break;
    case FS_DARBOUX: hipLaunchKernelGGL(k_tile<FS_DARBOUX>, grid, block, 0, st, a); break;
    case FS_SHADOW2: hipLaunchKernelGGL(k_tile<FS_SHADOW2>, grid, block, 0, st, a); break;
    case FS_OCCLUSION2: hipLaunchKernelGGL(k_tile<FS_OCCLUSION2>, grid, block, 0, st, a); break;
    case FS_DEPTH: hipLaunchKernelGGL(k_tile<FS_DEPTH>, grid, block, 0, st, a); break;
    default: return -1;
    }
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_fill_u32(uint32_t *dst, uint32_t value, size_t n, hipStream_t st)
{
    if (n == 0) return 0;
    size_t blocks = (n + 255u) / 256u;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(k_fill_u32, dim3((uint32_t)blocks), dim3(256), 0, st, dst, value, n);
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_depth_view(const float *src, uint8_t *dst, uint32_t W, uint32_t H, hipStream_t st)
{
    const size_t n = (size_t)W * H;
    if (n == 0) return 0;
    size_t blocks = (n + 255u) / 256u;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(k_depth_view, dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, W, H);
    TR_LAUNCH_CHECK();
    return 0;
}

}  // namespace tr
