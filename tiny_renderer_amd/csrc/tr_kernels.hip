// tr_kernels.hip -- gfx950 kernels of the triangle-fill path.
//
// The reference (src/scene.rs:151-268) is one serial loop nest: pass x polygon x bbox-x x
// bbox-y, depth-testing and shading each covered pixel in polygon order.  Here a render pass is
// two kernels:
//
//   k_setup   one lane per polygon: vertex closure, clamped bounding box; then the wavefront
//             spreads its (polygon, tile) pairs over all 64 lanes and appends the polygon's
//             complete record to the fixed-capacity bin of every tile it touches
//   k_tile    one 256-thread workgroup per 128x16 screen tile: the bin is copied into LDS in one
//             coalesced sweep, each wave resolves coverage + depth for its 32x16 quadrant against
//             LDS keys, then the survivors are shaded from the LDS records and depth and colour
//             are streamed out once, as whole cache lines
//
// Equivalence with the serial loop: `z <= zbuf -> reject` in polygon order means the surviving
// fragment of a pixel is the one with the largest z, ties going to the lowest polygon index, and
// the frame buffer keeps the colour of the last accepted fragment = that survivor
// (shader.rs:169-180, scene.rs:259-263).  A max over (z, index) computes the same survivor in
// any order, so only survivors are shaded.  The depth-only passes use `>=` (shader.rs:703):
// largest z, ties to the highest index.
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

__device__ __forceinline__ int32_t tile_index(const DevFrame &f, int32_t tx, int32_t ty)
{
    return (ty - f.ty_base) * (int32_t)f.ntx + tx;
}

// -----------------------------------------------------------------------------------------
// k_setup
// -----------------------------------------------------------------------------------------
template <int VS>
__global__ __launch_bounds__(64) void k_setup(SetupArgs a)
{
    constexpr int P = (VS == VS_DARBOUX) ? REC_PIECES_LARGE : REC_PIECES_SMALL;
    __shared__ uint4 s_rec[64 * P];
    __shared__ int32_t s_excl[64], s_tx0[64], s_ty0[64], s_ntx[64];

    const uint32_t lane = threadIdx.x;
    const uint32_t t = blockIdx.x * 64u + lane;

    int32_t tx0 = 0, ty0 = 0, ntx = 1, cnt = 0;
    uint32_t err = 0;
    if (t < a.mesh.n_tri) {
        RasterRec r;
        float v[VARY_STRIDE];
#pragma unroll
        for (int i = 0; i < VARY_STRIDE; i++) v[i] = 0.0f;
        const bool keep = vertex_stage<VS>(a.mesh, a.u, t, r, v, err);
        if (keep)
            finish_raster_rec(r, a.frame);
        else
            mark_rejected(r);
        if (r.bx0 <= r.bx1) {
            uint4 *o = s_rec + lane * P;
            o[0] = make_uint4((uint32_t)r.bx0, (uint32_t)r.bx1, (uint32_t)r.by0, (uint32_t)r.by1);
            o[1] = make_uint4((uint32_t)r.x0, (uint32_t)r.y0, (uint32_t)r.x1, (uint32_t)r.y1);
            o[2] = make_uint4((uint32_t)r.x2, (uint32_t)r.y2, __float_as_uint(r.z0), __float_as_uint(r.z1));
            o[3] = make_uint4(__float_as_uint(r.z2), t, __float_as_uint(v[0]), __float_as_uint(v[1]));
#pragma unroll
            for (int i = 4; i < P; i++)
                o[i] = make_uint4(__float_as_uint(v[4 * i - 14]), __float_as_uint(v[4 * i - 13]),
                                  __float_as_uint(v[4 * i - 12]), __float_as_uint(v[4 * i - 11]));
            tx0 = r.bx0 / TILE_W;
            ty0 = r.by0 / TILE_H;
            ntx = r.bx1 / TILE_W - tx0 + 1;
            cnt = ntx * (r.by1 / TILE_H - ty0 + 1);
        }
    }
    if (err) atomicOr(a.err, err);

    // Binning.  One lane per polygon would serialise a polygon's atomics (a polygon spanning 30
    // tiles = 30 dependent round trips); instead the wave's (polygon, tile) pairs are numbered by
    // a prefix sum and dealt round-robin to the lanes, four per lane per trip so that the four
    // returning atomics are in flight together.
    int32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t up = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += up;
    }
    const int32_t total = __shfl(incl, 63, 64);
    s_excl[lane] = incl - cnt;
    s_tx0[lane] = tx0;
    s_ty0[lane] = ty0;
    s_ntx[lane] = ntx;
    __syncthreads();

    for (int32_t p0 = (int32_t)lane; p0 < total; p0 += 256) {
        int32_t own[4], tile[4];
        uint32_t slot[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int32_t p = p0 + 64 * k;
            own[k] = -1;
            tile[k] = 0;
            if (p < total) {
                // owner = last lane whose exclusive offset is <= p (lanes without pairs share
                // their successor's offset, so "last" skips them)
                int32_t lo = 0, hi = 63;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const int32_t mid = (lo + hi + 1) >> 1;
                    if (s_excl[mid] <= p)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                const int32_t q = p - s_excl[lo], w = s_ntx[lo];
                own[k] = lo;
                tile[k] = tile_index(a.frame, s_tx0[lo] + q % w, s_ty0[lo] + q / w);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            slot[k] = own[k] >= 0 ? atomicAdd(&a.tile_count[tile[k]], 1u) : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (own[k] < 0) continue;
            if (slot[k] < a.bin_cap) {
                uint4 *dst = reinterpret_cast<uint4 *>(a.bins) + ((size_t)tile[k] * a.bin_cap + slot[k]) * P;
                const uint4 *src = s_rec + own[k] * P;
#pragma unroll
                for (int i = 0; i < P; i++) dst[i] = src[i];
            } else {
                atomicOr(a.err, (uint32_t)DE_BIN_OVERFLOW);
                atomicMax(a.bin_need, slot[k] + 1u);
            }
        }
    }
}

// -----------------------------------------------------------------------------------------
// k_tile
// -----------------------------------------------------------------------------------------

// Bijection on [0, n): odd multiplications and xor-shifts are bijections on [0, 2^bits); values
// that fall outside [0, n) are walked through the same map again (cycle walking).
__device__ __forceinline__ uint32_t scatter_tile(uint32_t b, uint32_t n, uint32_t bits)
{
    const uint32_t mask = (1u << bits) - 1u, sh = (bits + 1u) >> 1;
    uint32_t x = b;
    do {
        x = (x * 0x9E3779B1u) & mask;
        x ^= x >> sh;
        x = (x * 0x85EBCA6Bu) & mask;
        x ^= x >> sh;
    } while (x >= n);
    return x;
}

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

__device__ __forceinline__ int32_t bcast(uint32_t v, uint32_t lane)
{
    return __builtin_amdgcn_readlane((int)v, (int)lane);
}

template <int FS>
__global__ __launch_bounds__(256) void k_tile(TileArgs a)
{
    constexpr bool DEPTH = (FS == FS_DEPTH);
    constexpr int P = (FS == FS_DARBOUX) ? REC_PIECES_LARGE : REC_PIECES_SMALL;
    constexpr int NMAX = LDS_REC_BYTES / (P * 16);  // records resident in LDS

    // Per pixel: .x = depth_order_key(z) of the best fragment so far, .y = its bin slot + 1
    // (0 = "what the buffer held before this pass").
    __shared__ uint2 s_key[TILE_W * TILE_H];
    __shared__ uint4 s_rec[NMAX * P];

    // Blocks are dealt to XCDs / shader engines / CUs round-robin in launch order, each CU
    // receiving the same number of blocks.  A model in the middle of the screen makes the busy
    // tiles periodic in any affine function of the row-major tile index, and they pile up on a
    // subset of the CUs (measured: 104 of 256 CUs received no busy tile).  A bijective hash of the
    // block index breaks the periodicity (pure placement: any order is correct).
    const uint32_t tile = scatter_tile(blockIdx.x, a.frame.ntx * a.frame.nty, a.scatter_bits);
    const int32_t tx = (int32_t)(tile % a.frame.ntx);
    const int32_t ty = a.frame.ty_base + (int32_t)(tile / a.frame.ntx);
    const int32_t tile_x0 = tx * TILE_W, tile_y0 = ty * TILE_H;
    const int32_t W = (int32_t)a.frame.width, H = (int32_t)a.frame.height;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;

    uint32_t n = a.tile_count[tile];
    if (n > a.bin_cap) n = a.bin_cap;  // overflow: flagged by k_setup, the host renders again

    // Diagnostic builds of a scene (TR_OPT_TILE_STAMPS) record when each tile ran; the stamps go
    // to a buffer of their own and nothing is computed from them.
    uint64_t t_start = 0, t_staged = 0;
    if (a.stamps) t_start = wall_clock64();

    if (n == 0u && a.fresh) {
        write_cleared_tile<DEPTH>(a, tile_x0, tile_y0);
        if (a.stamps && tid == 0u) {
            a.stamps[8u * tile + 0u] = t_start;
            a.stamps[8u * tile + 1u] = wall_clock64();
            a.stamps[8u * tile + 2u] = 0u;
            a.stamps[8u * tile + 3u] = __smid();
        }
        return;
    }

    float *depth = DEPTH ? a.shadow : a.zbuf;
    const int32_t qx0 = tile_x0 + (int32_t)wave * QUAD, qy0 = tile_y0;
    uint2 *wkey = s_key + wave * QPIX;
    const int32_t lx = (int32_t)(lane & 7u), ly = (int32_t)(lane >> 3);
    const uint4 *bin = reinterpret_cast<const uint4 *>(a.bins) + (size_t)tile * a.bin_cap * P;
    const bool resident = n <= (uint32_t)NMAX;  // the whole bin stays in LDS through shading

    // ---- initial keys -------------------------------------------------------------------
    {
        const uint32_t zmin = depth_order_key(bits_f32(TR_F32_MIN_BITS));
#pragma unroll
        for (int b = 0; b < NBX * NBY; b++) {
            uint32_t zk = zmin;
            if (!a.fresh) {
                const int32_t px = qx0 + (b % NBX) * 8 + lx, py = qy0 + (b / NBX) * 8 + ly;
                if (px < W && py >= a.frame.band_y0 && py < a.frame.band_y1)
                    zk = depth_order_key(depth[(size_t)py * W + px]);
            }
            wkey[(b << 6) + lane] = make_uint2(zk, 0u);
        }
    }

    // ---- coverage + depth resolve ---------------------------------------------------------
    // The bin (n records of P 16-byte pieces, contiguous) is copied to LDS by all 256 threads,
    // NMAX records at a time.  Each wave then takes 64 records at a time into registers (lane l
    // holds the raster part of record l), ballots which of them touch its quadrant and walks
    // those, broadcasting a record to the scalar registers with v_readlane: no memory access at
    // all per polygon.
    for (uint32_t c0 = 0; c0 < n; c0 += NMAX) {
        const uint32_t m = min((uint32_t)NMAX, n - c0);
        if (c0 != 0u) __syncthreads();  // every wave is done with the previous chunk
        for (uint32_t q = tid; q < m * P; q += 256u) s_rec[q] = bin[(size_t)c0 * P + q];
        __syncthreads();
        if (a.stamps && c0 == 0u) t_staged = wall_clock64();

        for (uint32_t j0 = 0; j0 < m; j0 += 64u) {
            const uint32_t jj = j0 + lane;
            uint4 r0 = make_uint4(1u, 0u, 1u, 0u), r1 = make_uint4(0, 0, 0, 0), r2 = r1, r3 = r1;
            if (jj < m) {
                r0 = s_rec[jj * P + 0];
                r1 = s_rec[jj * P + 1];
                r2 = s_rec[jj * P + 2];
                r3 = s_rec[jj * P + 3];
            }
            const bool touch = imax((int32_t)r0.x, qx0) <= imin((int32_t)r0.y, qx0 + QUAD - 1) &&
                               imax((int32_t)r0.z, qy0) <= imin((int32_t)r0.w, qy0 + TILE_H - 1);
            unsigned long long todo = __ballot(touch);
            while (todo) {
                const uint32_t l = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                RasterRec r;
                const int32_t bx0 = imax(bcast(r0.x, l), qx0), bx1 = imin(bcast(r0.y, l), qx0 + QUAD - 1);
                const int32_t by0 = imax(bcast(r0.z, l), qy0), by1 = imin(bcast(r0.w, l), qy0 + TILE_H - 1);
                r.x0 = bcast(r1.x, l); r.y0 = bcast(r1.y, l); r.x1 = bcast(r1.z, l); r.y1 = bcast(r1.w, l);
                r.x2 = bcast(r2.x, l); r.y2 = bcast(r2.y, l);
                r.z0 = __int_as_float(bcast(r2.z, l)); r.z1 = __int_as_float(bcast(r2.w, l));
                r.z2 = __int_as_float(bcast(r3.x, l));
                const uint32_t id = (uint32_t)bcast(r3.y, l);
                const uint32_t slot1 = c0 + j0 + l + 1u;
                const Edge e = edge_setup(r);
                const Recip rz = recip_of(e.cz);  // one IEEE division per polygon visit
                const int32_t ib0 = (bx0 - qx0) >> 3, ib1 = (bx1 - qx0) >> 3;
                const int32_t jb0 = (by0 - qy0) >> 3, jb1 = (by1 - qy0) >> 3;
                for (int32_t jb = jb0; jb <= jb1; jb++) {
                    for (int32_t ib = ib0; ib <= ib1; ib++) {
                        const int32_t px = qx0 + ib * 8 + lx, py = qy0 + jb * 8 + ly;
                        float cx, cy;
                        edge_cross(e, px, py, cx, cy);
                        if (px >= bx0 && px <= bx1 && py >= by0 && py <= by1 && covers(cx, cy, e.cz)) {
                            const vec3 bar = barycentric_by(cx, cy, rz);
                            const float z = dot3(bar, make3(r.z0, r.z1, r.z2));
                            const uint32_t zk = depth_order_key(z);
                            uint2 *slot = wkey + (((jb * NBX + ib) << 6) + (int32_t)lane);
                            const uint2 cur = *slot;
                            bool win = zk > cur.x;
                            if (zk == cur.x) {
                                // equal depth: the buffer's previous content beats a colour
                                // fragment (`z <= zbuf` rejects) and loses to a depth fragment
                                // (`z >= shadow` accepts); between two fragments of this pass
                                // the polygon index decides
                                if (cur.y == 0u) {
                                    win = DEPTH;
                                } else {
                                    const uint32_t cur_id = resident ? s_rec[(cur.y - 1u) * P + 3].y
                                                                     : bin[(size_t)(cur.y - 1u) * P + 3].y;
                                    win = DEPTH ? id > cur_id : id < cur_id;
                                }
                            }
                            if (win) *slot = make_uint2(zk, slot1);
                        }
                    }
                }
            }
        }
    }

    uint64_t t_covered = 0;
    if (a.stamps) t_covered = wall_clock64();

    // ---- shade the survivors and stream the tile out ------------------------------------------
    // Lanes are row-major here (lane = x within a 32-pixel row, two rows per wave step), so depth
    // is stored straight from registers as whole 128-byte lines and colour is packed to dwords
    // with two lane permutes.  The vertical flip of get_frame_buffer (scene.rs:92-97) is folded
    // into the colour address.  SHADE_G steps are processed together and branch-free (lanes
    // without a survivor run the same loads on record 0 and discard the result) so that the
    // record and texel fetches of different rows overlap.
    const int32_t hx = (int32_t)(lane & 31u), hrow = (int32_t)(lane >> 5);
    const uint32_t half_base = lane & 32u;
#pragma unroll 1
    for (int32_t g = 0; g < TILE_H / 2; g += SHADE_G) {
        int32_t py[SHADE_G];
        bool live[SHADE_G], won[SHADE_G];
        uint32_t wslot[SHADE_G], tri[SHADE_G], rgb[SHADE_G];
        float zout[SHADE_G];
        const int32_t px = qx0 + hx;
        bool any_won = false;
#pragma unroll
        for (int u = 0; u < SHADE_G; u++) {
            const int32_t qy = (g + u) * 2 + hrow;
            py[u] = qy0 + qy;
            live[u] = px < W && py[u] >= a.frame.band_y0 && py[u] < a.frame.band_y1;
            const uint32_t s1 = wkey[key_slot((uint32_t)hx, (uint32_t)qy)].y;
            won[u] = live[u] && s1 != 0u;
            wslot[u] = won[u] ? s1 - 1u : 0u;
            tri[u] = NO_WINNER;
            rgb[u] = 0u;
            zout[u] = bits_f32(TR_F32_MIN_BITS);
            any_won = any_won || won[u];
        }
        if (__any(any_won)) {
            uint32_t err = 0u;
#pragma unroll
            for (int u = 0; u < SHADE_G; u++) {
                uint4 q[P];
                if (resident) {
#pragma unroll
                    for (int i = 1; i < P; i++) q[i] = s_rec[wslot[u] * P + i];
                } else {
#pragma unroll
                    for (int i = 1; i < P; i++) q[i] = bin[(size_t)wslot[u] * P + i];
                }
                RasterRec r;
                r.x0 = (int32_t)q[1].x; r.y0 = (int32_t)q[1].y; r.x1 = (int32_t)q[1].z; r.y1 = (int32_t)q[1].w;
                r.x2 = (int32_t)q[2].x; r.y2 = (int32_t)q[2].y;
                r.z0 = __uint_as_float(q[2].z); r.z1 = __uint_as_float(q[2].w); r.z2 = __uint_as_float(q[3].x);
                const Edge e = edge_setup(r);
                float cx, cy;
                edge_cross(e, px, py[u], cx, cy);
                const vec3 bar = barycentric(cx, cy, e.cz);
                const float z = dot3(bar, make3(r.z0, r.z1, r.z2));
                uint32_t c = 0u, e1 = 0u;
                if (!DEPTH) {
                    float v[VARY_STRIDE];
                    v[0] = __uint_as_float(q[3].z);
                    v[1] = __uint_as_float(q[3].w);
#pragma unroll
                    for (int i = 4; i < P; i++) {
                        v[4 * i - 14] = __uint_as_float(q[i].x);
                        v[4 * i - 13] = __uint_as_float(q[i].y);
                        v[4 * i - 12] = __uint_as_float(q[i].z);
                        v[4 * i - 11] = __uint_as_float(q[i].w);
                    }
                    c = fragment_stage<FS>(a.u, a.tex, v, bar, (uint32_t)px, (uint32_t)py[u], z, a.shadow,
                                           (uint32_t)W, (uint32_t)H, e1);
                }
                if (won[u]) {
                    zout[u] = z;
                    rgb[u] = c;
                    tri[u] = q[3].y;
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
    if (tid == 0u) {
        a.tile_count[tile] = 0u;
        if (a.stamps) {
            a.stamps[8u * tile + 0u] = t_start;
            a.stamps[8u * tile + 1u] = wall_clock64();
            a.stamps[8u * tile + 2u] = n;
            a.stamps[8u * tile + 3u] = __smid();
            a.stamps[8u * tile + 4u] = t_staged;
            a.stamps[8u * tile + 5u] = t_covered;
        }
    }
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

// tr_selftest_device_math: the device forms of the casts and of the shared-reciprocal division,
// applied to caller-chosen operands so the host can compare them with its own.
__global__ __launch_bounds__(256) void k_selftest(const float *x, const float *d, uint32_t n, uint32_t *out_u32,
                                                  int32_t *out_i32, uint32_t *out_u8, float *out_div,
                                                  float *out_div_ref)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_u32[i] = f32_to_u32(x[i]);
    out_i32[i] = f32_to_i32(x[i]);
    out_u8[i] = f32_to_u8(x[i]);
    out_div[i] = div_by(x[i], recip_of(d[i]));
    out_div_ref[i] = x[i] / d[i];
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

int rec_pieces_for_vs(int vs) { return vs == VS_DARBOUX ? REC_PIECES_LARGE : REC_PIECES_SMALL; }
int rec_pieces_for_fs(int fs) { return fs == FS_DARBOUX ? REC_PIECES_LARGE : REC_PIECES_SMALL; }

int launch_setup(int vs, const SetupArgs &a, hipStream_t st)
{
    if (a.mesh.n_tri == 0) return 0;
    if ((int)a.rec_pieces != rec_pieces_for_vs(vs)) return (int)hipErrorInvalidValue;
    const dim3 grid((a.mesh.n_tri + 63u) / 64u), block(64);
    switch (vs) {
    case VS_DEFAULT: hipLaunchKernelGGL(k_setup<VS_DEFAULT>, grid, block, 0, st, a); break;
    case VS_PHONG: hipLaunchKernelGGL(k_setup<VS_PHONG>, grid, block, 0, st, a); break;
    case VS_PLAIN: hipLaunchKernelGGL(k_setup<VS_PLAIN>, grid, block, 0, st, a); break;
    case VS_DARBOUX: hipLaunchKernelGGL(k_setup<VS_DARBOUX>, grid, block, 0, st, a); break;
    case VS_DEPTH: hipLaunchKernelGGL(k_setup<VS_DEPTH>, grid, block, 0, st, a); break;
    default: return (int)hipErrorInvalidValue;
    }
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_tile(int fs, const TileArgs &a, hipStream_t st)
{
    const uint32_t n_tiles = a.frame.ntx * a.frame.nty;
    if (n_tiles == 0) return 0;
    if ((int)a.rec_pieces != rec_pieces_for_fs(fs)) return (int)hipErrorInvalidValue;
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
    default: return (int)hipErrorInvalidValue;
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

int launch_selftest(const float *x, const float *d, uint32_t n, uint32_t *out_u32, int32_t *out_i32,
                    uint32_t *out_u8, float *out_div, float *out_div_ref, hipStream_t st)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_selftest, dim3((n + 255u) / 256u), dim3(256), 0, st, x, d, n, out_u32, out_i32, out_u8,
                       out_div, out_div_ref);
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
