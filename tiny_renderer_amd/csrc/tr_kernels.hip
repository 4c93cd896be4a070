// tr_kernels.hip -- gfx950 kernels of the triangle-fill path.
//
// The reference (src/scene.rs:151-268) is one serial loop nest: pass x polygon x bbox-x x
// bbox-y, depth-testing and shading each covered pixel in polygon order.  Here a render pass is
// four kernels:
//
//   k_setup        one lane per polygon: vertex closure, clamped bounding box; then the wavefront
//                  spreads its (polygon, tile) pairs over all 64 lanes and appends the polygon's
//                  complete record to the fixed-capacity bin of every tile it touches
//   k_order_count, k_order_place
//                  one thread per tile: the tile kernel's work list, tiles sorted by polygon count
//   k_tile         one workgroup (4, 8 or 16 waves) per 128x16 screen tile, heaviest tiles first:
//                  the bin is copied into LDS in one coalesced pass, each wave resolves coverage +
//                  depth for its column of the tile against LDS keys, then the survivors are shaded
//                  from the LDS records and depth and colour are streamed out once, as whole cache
//                  lines; tiles without polygons only stream their cleared colour
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
#include <hip/hip_ext.h>

#include <type_traits>

#include "tr_kernels.h"
#include "tr_plan.h"
#include "tr_shaders.h"

namespace tr {

namespace {

constexpr int NBY = TILE_H / 8;  // block rows per quadrant
constexpr uint32_t NO_WINNER = 0xFFFFFFFFu;
static_assert(NBY == 2 && TILE_H % 4 == 0, "the pixel-pair code assumes two block rows per quadrant");

__device__ __forceinline__ int32_t tile_index(const DevFrame &f, int32_t tx, int32_t ty)
{
    return (ty - f.ty_base) * (int32_t)f.ntx + tx;
}

// -----------------------------------------------------------------------------------------
// k_setup
// -----------------------------------------------------------------------------------------
// Binning is EXACT: k_setup runs the vertex stage and counts, per tile, the polygons whose clamped box meets it;
// k_order gives every tile with polygons a range of exactly that many records in the pass's pool (one atomic per
// wave: tiles need no particular order in the pool); k_bin then fills the ranges.  No tile has a capacity: ten
// times the usual number of polygons in one tile is just a longer range.  What is bounded is the pool -- the
// pairs of the whole pass -- sized generously by the host and grown (the frame rendered again) should a pass ever
// exceed it.  (Round 2 gave every tile a fixed-capacity bin: 4 GiB of reservations at 4096^2 for 2.6 MB of records
// per frame, and TR_E_BIN_OVERFLOW whenever one tile saw more than its share.)
// Workgroups of the chain's kernels are FOUR independent waves: a compute unit that is full of tile-kernel
// workgroups admits a wave of another kernel only where a tile workgroup has left, and that wave -- on one SIMD --
// then keeps the next tile workgroup (a wave on every SIMD) out for as long as it lives.  Four waves that arrive
// together, one per SIMD, block that one slot once; as 64-thread workgroups the same waves blocked up to four
// slots, and the 4096^2 tile kernel ran 4 % slower beside them.  (A lone frame's chain has the machine to itself
// and is launched as single waves: they spread over four times as many compute units.)
constexpr uint32_t CHAIN_WAVES = 4;
constexpr uint32_t CHAIN_THREADS = 64 * CHAIN_WAVES;

// Polygon t's record (P pieces) for the pass `a` describes: vertex closure, truncating projection, clamped box.
// A polygon that draws nothing (culled, off screen, degenerate, t beyond the mesh) gets the canonical empty box
// (bx0 = 1 > bx1 = 0) and nothing else.  Returns the device error bits the vertex stage raised.
template <int VS, int P>
__device__ __forceinline__ uint32_t setup_record(const SetupArgs &a, uint32_t t, uint4 (&o)[P])
{
    uint32_t err = 0;
#pragma unroll
    for (int i = 0; i < P; i++) o[i] = make_uint4(0u, 0u, 0u, 0u);
    o[0] = make_uint4(1u, 0u, 0u, 0u);
    if (t >= a.mesh.n_tri) return 0u;
    RasterRec r;
    float v[VARY_STRIDE];
#pragma unroll
    for (int i = 0; i < VARY_STRIDE; i++) v[i] = 0.0f;
    const bool keep = vertex_stage<VS>(a.mesh, a.u, t, r, v, err);
    if (keep)
        finish_raster_rec(r, a.frame);
    else
        mark_rejected(r);
    // piece 0: the clamped box (coordinates below 2^15; bx0 = 1 > bx1 = 0: draws nothing) and, in a tile's copy,
    // the pair's coverage masks (pair_masks, tr_shaders.h: filled in by k_bin)
    // (a polygon that draws nothing may have any coordinates: it gets the canonical empty box)
    if (r.bx0 <= r.bx1) {
        o[0] = make_uint4((uint32_t)r.bx0 | ((uint32_t)r.bx1 << 16), (uint32_t)r.by0 | ((uint32_t)r.by1 << 16), 0u, 0u);
        // vertex 0 and the two edge vectors from it, the latter already as the f32 values every
        // pixel's to_barycentric_coord starts from (scene.rs:178-181: integer difference, then
        // the conversion) -- the tile kernel's shading phase used to redo these 8 subtractions
        // and 8 conversions for every pixel pair
        const Edge e = edge_setup(r);
        o[1] = make_uint4((uint32_t)r.x0, (uint32_t)r.y0, __float_as_uint(e.a0), __float_as_uint(e.b0));
        o[2] = make_uint4(__float_as_uint(e.a1), __float_as_uint(e.b1), __float_as_uint(r.z0), __float_as_uint(r.z1));
        o[3] = make_uint4(__float_as_uint(r.z2), t, __float_as_uint(v[0]), __float_as_uint(v[1]));
#pragma unroll
        for (int i = 4; i < P - 1; i++)
            o[i] = make_uint4(__float_as_uint(v[4 * i - 14]), __float_as_uint(v[4 * i - 13]),
                              __float_as_uint(v[4 * i - 12]), __float_as_uint(v[4 * i - 11]));
        // spare last word (varying 9 / 21 is unused): RN(1 / cross.z), the one IEEE division
        // per polygon; k_tile derives every per-pixel quotient from it (tr_math.h div_by)
        o[P - 1] = make_uint4(__float_as_uint(v[4 * (P - 1) - 14]), __float_as_uint(v[4 * (P - 1) - 13]),
                              __float_as_uint(v[4 * (P - 1) - 12]), __float_as_uint(record_recip(r)));
    }
    return err;
}

// The (polygon, tile) pairs of a wave whose lane l holds the box of one polygon (or the empty box): numbered by a
// prefix sum over the lanes, so that they can be dealt to all 64 lanes -- a polygon that spans 40 tiles is 40 lanes'
// work for one step, not one lane's loop of 40 while the others idle.
struct PairDeal {
    int32_t corner, ntx, excl, total;  // this lane's polygon: first tile column | first tile row << 16, tile columns, pairs before it
    __device__ __forceinline__ PairDeal(const uint4 &piece0)
    {
        const uint32_t lane = threadIdx.x & 63u;
        const int32_t bx0 = (int32_t)(piece0.x & 0xFFFFu), bx1 = (int32_t)(piece0.x >> 16);
        const int32_t by0 = (int32_t)(piece0.y & 0xFFFFu), by1 = (int32_t)(piece0.y >> 16);
        int32_t cnt = 0;
        corner = 0;
        ntx = 1;
        if (bx0 <= bx1) {
            corner = bx0 / TILE_W | (by0 / TILE_H) << 16;
            ntx = bx1 / TILE_W - bx0 / TILE_W + 1;
            cnt = ntx * (by1 / TILE_H - by0 / TILE_H + 1);
        }
        int32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t up = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += up;
        }
        total = __shfl(incl, 63, 64);
        excl = incl - cnt;
    }
    // pair p (every lane of the wave calls this together; p may lie beyond `total`): the lane that holds its polygon
    // and the tile; false beyond the last pair
    __device__ __forceinline__ bool locate(int32_t p, int32_t &owner, int32_t &ptx, int32_t &pty) const
    {
        // owner = last lane whose exclusive offset is <= p (lanes without pairs share their successor's offset, so
        // "last" skips them)
        int32_t lo = 0, hi = 63;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (__shfl(excl, mid, 64) <= p)
                lo = mid;
            else
                hi = mid - 1;
        }
        const int32_t q = p - __shfl(excl, lo, 64), w = __shfl(ntx, lo, 64), c = __shfl(corner, lo, 64);
        owner = lo;
        ptx = (c & 0xFFFF) + q % w;
        pty = (c >> 16) + q / w;
        return p < total;
    }
};

// Bumps the counter of every tile the wave's polygons' boxes meet (atomics that return nothing: nothing waits for them).
__device__ __forceinline__ void count_tiles(const SetupArgs &a, const uint4 &piece0)
{
    const PairDeal deal(piece0);
    const int32_t lane = (int32_t)(threadIdx.x & 63u);
    for (int32_t p0 = 0; p0 < deal.total; p0 += 64) {
        int32_t owner, ptx, pty;
        if (deal.locate(p0 + lane, owner, ptx, pty))
            __hip_atomic_fetch_add(&a.tile_count[tile_index(a.frame, ptx, pty)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Which polygon a lane of the chain's kernels takes.  The first `polys` lanes of every wave take one each, INTERLEAVED
// over the launch's waves (lane l of wave w: polygon l * waves + w): neighbours in a mesh are neighbours on the screen
// and of similar size -- in mesh order one wave got the eight largest polygons of the reference's model and was the
// critical path of the whole chain (k_bin's waves: median 3.4 us, slowest 13).
__device__ __forceinline__ uint32_t chain_polygon(uint32_t block, uint32_t polys)
{
    const uint32_t lane = threadIdx.x & 63u, wave = block * (blockDim.x >> 6) + (threadIdx.x >> 6), waves = gridDim.x * (blockDim.x >> 6);
    return lane < polys ? lane * waves + wave : 0xFFFFFFFFu;
}

template <int VS>
__device__ __forceinline__ void setup_body(const SetupArgs &a, uint32_t block, uint32_t polys)
{
    constexpr int P = (VS == VS_DARBOUX) ? REC_PIECES_LARGE : REC_PIECES_SMALL;
    // Lane = polygon: vertex closure, truncating projection, clamped box; the polygon's record goes to the pass's
    // record array ONCE (k_bin copies it to the tiles), its tiles' counters are bumped with atomics that return
    // nothing, so nothing waits for them.
    __builtin_amdgcn_s_setprio(3);  // (see k_order)
    // the 16 words behind this pass's counters (k_order's list lengths and pool cursor): their last readers --
    // the tile kernel of the pass that had the set before -- are done (the host orders that)
    if (block == 0u && threadIdx.x < 16u) a.tile_count[a.frame.ntx * a.frame.nty + threadIdx.x] = 0u;
    const uint32_t t = chain_polygon(block, polys);
    uint4 rec[P];
    const uint32_t err = setup_record<VS, P>(a, t, rec);
    if (t < a.mesh.n_tri) {
        uint4 *o = reinterpret_cast<uint4 *>(a.recs) + (size_t)t * P;
        o[0] = rec[0];
        if ((rec[0].x & 0xFFFFu) <= (rec[0].x >> 16)) {
#pragma unroll
            for (int i = 1; i < P; i++) o[i] = rec[i];
        }
    }
    count_tiles(a, rec[0]);
    if (err) {
        atomicOr(a.err, err);
        *a.alarm = 1u;
    }
}

template <int VS>
__global__ __launch_bounds__(CHAIN_THREADS) void k_setup(SetupArgs a, uint32_t polys)
{
    setup_body<VS>(a, blockIdx.x, polys);
}

// k_bin: copies every polygon's record into the range of each tile its box meets, with the pair's coverage masks.
// One lane per polygon would serialise a polygon's copies (a polygon spanning 30 tiles = 30 dependent round
// trips); instead the wave's (polygon, tile) pairs are numbered by a prefix sum and dealt round-robin to the
// lanes, PAIRS per lane per trip so that their returning atomics are in flight together.  A pair's place in the
// pool comes from counting the tile's counter DOWN from the end of the tile's range (k_order left it there).
// `polys`: polygons per wave (the wave with the most pairs is the kernel's critical path, and a lone frame waits
// for it: few for a small mesh).
constexpr uint32_t BIN_POLYS = 64;

__device__ __forceinline__ uint4 shuffle_piece(const uint4 &v, int32_t from)
{
    return make_uint4((uint32_t)__shfl((int)v.x, from, 64), (uint32_t)__shfl((int)v.y, from, 64), (uint32_t)__shfl((int)v.z, from, 64),
                      (uint32_t)__shfl((int)v.w, from, 64));
}

// The wave's polygons are in its lanes' registers (`mine`: lane l holds one record, or the empty box): their
// (polygon, tile) pairs are numbered by a prefix sum and dealt to all 64 lanes, which fetch the record from the
// owner lane with shuffles (no LDS: nothing to fit beside the tile kernel's workgroups, no barrier).
template <int P>
__device__ __forceinline__ void bin_deal(const SetupArgs &a, const uint4 (&mine)[P])
{
    const int32_t lane = (int32_t)(threadIdx.x & 63u);
    const PairDeal deal(mine[0]);
    constexpr int PAIRS = 2;  // pairs per lane per trip: their atomics are in flight together
    for (int32_t p0 = 0; p0 < deal.total; p0 += 64 * PAIRS) {
        int32_t own[PAIRS], origin[PAIRS];
        uint32_t at[PAIRS];
        bool have[PAIRS];
#pragma unroll
        for (int k = 0; k < PAIRS; k++) {
            have[k] = false;
            own[k] = 0;
            if (p0 + 64 * k >= deal.total) continue;  // (uniform)
            int32_t ptx, pty;
            have[k] = deal.locate(p0 + 64 * k + lane, own[k], ptx, pty);
            if (have[k]) {
                origin[k] = ptx | pty << 16;
                at[k] = atomicSub(&a.tile_count[tile_index(a.frame, ptx, pty)], 1u) - 1u;
            }
        }
#pragma unroll
        for (int k = 0; k < PAIRS; k++) {
            if (p0 + 64 * k >= deal.total) continue;  // (uniform: every lane takes part in the shuffles)
            uint4 piece[P];
#pragma unroll
            for (int i = 0; i < P; i++) piece[i] = shuffle_piece(mine[i], own[k]);
            // (pool exhausted: k_order has raised DE_BIN_OVERFLOW, the host renders again)
            if (!have[k] || at[k] >= a.pool_cap) continue;
            uint4 *dst = reinterpret_cast<uint4 *>(a.bins) + (size_t)at[k] * P;
            // which cells (small pair) or block columns (large pair) of the box inside THIS tile can hold
            // a fragment: the tile kernel evaluates no edge function to find its work
            pair_masks((int32_t)(piece[0].x & 0xFFFFu), (int32_t)(piece[0].x >> 16), (int32_t)(piece[0].y & 0xFFFFu),
                       (int32_t)(piece[0].y >> 16), (int32_t)piece[1].x, (int32_t)piece[1].y, __uint_as_float(piece[1].z),
                       __uint_as_float(piece[2].x), __uint_as_float(piece[1].w), __uint_as_float(piece[2].y),
                       (origin[k] & 0xFFFF) * TILE_W, (origin[k] >> 16) * TILE_H, a.cells != 0u, piece[0].z, piece[0].w);
#pragma unroll
            for (int i = 0; i < P; i++) dst[i] = piece[i];
        }
    }
}

template <int P>
__device__ __forceinline__ void bin_body(const SetupArgs &a, uint32_t block, uint32_t polys)
{
    __builtin_amdgcn_s_setprio(3);  // (see k_order)
    const uint32_t t = chain_polygon(block, polys);
    // the lane's polygon, in registers
    uint4 mine[P];
#pragma unroll
    for (int i = 0; i < P; i++) mine[i] = make_uint4(0u, 0u, 0u, 0u);
    mine[0] = make_uint4(1u, 0u, 0u, 0u);
    if (t < a.mesh.n_tri) {
        const uint4 *const rec = reinterpret_cast<const uint4 *>(a.recs) + (size_t)t * P;
#pragma unroll
        for (int i = 0; i < P; i++) mine[i] = rec[i];  // (a polygon that draws nothing has only its box written: the rest is not used)
    }
    bin_deal<P>(a, mine);
}

template <int P>
__global__ __launch_bounds__(CHAIN_THREADS) void k_bin(SetupArgs a, uint32_t polys)
{
    bin_body<P>(a, blockIdx.x, polys);
}

// Read-only argument tables of the fused launches: viewed in the constant address space, so that the loads
// are scalar and invariant (a member is loaded where it is used, arrays are indexed in place) -- what the
// compiler does with kernel arguments.
template <typename T>
using constant_ptr = const __attribute__((address_space(4))) T *;

// The same for a group of frames in one launch (tr_scene_render_frames): blockIdx.y = frame, whose
// arguments are entry y of a table in device memory.
template <int VS>
__global__ __launch_bounds__(CHAIN_THREADS) void k_setup_group(const SetupArgs *__restrict__ table, uint32_t polys)
{
    setup_body<VS>(*(const SetupArgs *)((constant_ptr<SetupArgs>)table + blockIdx.y), blockIdx.x, polys);
}

template <int P>
__global__ __launch_bounds__(CHAIN_THREADS) void k_bin_group(const SetupArgs *__restrict__ table, uint32_t polys)
{
    const SetupArgs &a = *(const SetupArgs *)((constant_ptr<SetupArgs>)table + blockIdx.y);
    // the pass's list lengths (k_order has completed) to the host, which sizes later tile kernels' grids by them
    constexpr uint32_t N_LISTS = 8u;  // (ORDER_BUCKETS, below)
    if (blockIdx.x == 0u && threadIdx.x < N_LISTS && a.len_host)
        __hip_atomic_store(a.len_host + threadIdx.x, gload(a.len_src + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    bin_body<P>(a, blockIdx.x, polys);
}

// -----------------------------------------------------------------------------------------
// k_lit
// -----------------------------------------------------------------------------------------
// The normal-map and specular closures use nothing of a fragment but its texel (shade_texel, tr_shaders.h): under one
// frame's light and camera the colour of a fragment is a function of (u, v)'s texel alone.  Where a frame shades many
// more fragments than the images have texels -- the x64 grid at 8192^2: ~20 M fragments, 1 M texels -- the closure
// runs ONCE PER TEXEL here, in the chain in front of the tile kernel, and the tile kernel's fragment stage is a fetch
// from the frame's lit image (FS_LIT).  The arithmetic is the fragment stage's own -- the two-texel closure with its
// guard, the plain closure (IEEE division and square root) behind it, the exact powf: what a fragment finds is what it
// would have computed.  Thread = two texels of the scene's texel set (four-word texels in 4x2 blocks); the lit image is
// a one-word image in 8x4 blocks (tr_texels.h).
template <int FS>
__device__ __forceinline__ void lit_body(const SetupArgs &a)
{
    // two texels per thread -- neighbours in a row of a 4x2 block -- through the two-texel closure (packed arithmetic,
    // shared-reciprocal divisions: shade_texel_pair); a texel that leaves its guarded range is redone by the plain one
    const uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 2u;
    const uint32_t block = i >> 3, within = i & 7u;
    const uint32_t x = (block % a.set_bpr) * 4u + (within & 3u), y = (block / a.set_bpr) * 2u + (within >> 2);
    if (x >= a.tex_w || y >= a.tex_h) return;
    const bool second = x + 1u < a.tex_w;
    const Texel4 q0 = reinterpret_cast<const Texel4 *>(a.texel_set)[i], q1 = reinterpret_cast<const Texel4 *>(a.texel_set)[i + 1u];
    const vec3 n0 = make3(bits_f32(q0.y), bits_f32(q0.z), bits_f32(q0.w)), n1 = make3(bits_f32(q1.y), bits_f32(q1.z), bits_f32(q1.w));
    PairGuard g = guard_init();
    uint32_t c0 = 0u, c1 = 0u;
    const f2 r = shade_texel_pair<FS>(a.u, q0.x & 0xFFFFFFu, q1.x & 0xFFFFFFu, make3p(mk2(n0.x, n1.x), mk2(n0.y, n1.y), mk2(n0.z, n1.z)),
                                      q0.x >> 24, q1.x >> 24, g, c0, c1);
    if (guard_bad(g, 0) || !(r.x == r.x)) c0 = shade_texel<FS>(a.u, q0.x & 0xFFFFFFu, n0, q0.x >> 24);
    if (second && (guard_bad(g, 1) || !(r.y == r.y))) c1 = shade_texel<FS>(a.u, q1.x & 0xFFFFFFu, n1, q1.x >> 24);
    const uint32_t at = packed_index(1, a.lit_bpr, x, y);  // (x even: x + 1 is the next texel of the same 8x4 block row)
    a.lit[at] = c0;
    if (second) a.lit[at + 1u] = c1;
}

template <int FS>
__global__ __launch_bounds__(256) void k_lit(SetupArgs a)
{
    lit_body<FS>(a);
}

template <int FS>
__global__ __launch_bounds__(256) void k_lit_group(const SetupArgs *__restrict__ table)
{
    lit_body<FS>(table[blockIdx.y]);
}

// -----------------------------------------------------------------------------------------
// k_order
// -----------------------------------------------------------------------------------------
// Turns the per-tile polygon counts k_setup left into the tile kernel's WORK LISTS: every tile once, as
// (tile, count), in one of eight lists by count -- >= 64, >= 32, ... , 1, and the empty tiles -- each with a
// region of its own (n_tiles entries), so that ONE sweep suffices: a wave counts its members of a list with a
// ballot, one lane per list reserves the wave's range with a single atomic, members take base + rank.  (Round 2
// needed a counting sweep first, to know where each list starts in a common array: two dependent kernels in
// front of every tile kernel.)  The tile kernel walks the lists heaviest first (longest-processing-time-first
// packing) and handles the empty tiles in batches.  Inside a list the tiles come in a hashed order: tiles that
// are neighbours on the screen cost about the same and read the same texture region -- in row-major order the
// tile kernel was 15 % slower.  A tile's counter leaves this kernel as the END of the tile's range in the pool
// (k_bin counts it down to the start; the tile's own k_tile workgroup zeroes it for the set's next pass); the 16
// words behind the counters -- the list lengths, the pool cursor -- are zeroed by k_setup's first workgroup
// earlier in the same stream.
constexpr int ORDER_BUCKETS = 8;
constexpr int ORDER_THREADS = 256;
constexpr int ORDER_EMPTY = ORDER_BUCKETS - 1;  // the list of the tiles without polygons
static_assert(ORDER_BUCKETS == 8, "k_bin_group reports eight list lengths");
// The tile kernel's WORK UNITS: one per tile with polygons (lists 0..6, heaviest first), then one per EMPTY_CHUNK
// entries of the empty list.  A workgroup per EMPTY tile -- three quarters of a 4096^2 frame of the reference's model --
// is four waves and 20 KiB of LDS dispatched to read one flag: 3.3 of a frame's 22.8 us went into dispatching them
// (measured with a grid cut behind the busy tiles, profiles/r04_notes.md).  A launch needs
// units(lengths) = busy + ceil(empty / EMPTY_CHUNK) workgroups per frame; any larger grid is correct (the surplus
// exits at once), n_tiles always suffices, and a host that knows the lengths (SetupArgs::len_host) asks for no more.
using plan::EMPTY_CHUNK;
static_assert(plan::ORDER_LISTS_ == (uint32_t)ORDER_BUCKETS, "tr_plan.h counts the lists the tile kernel walks");

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

__device__ __forceinline__ uint32_t order_bucket(uint32_t n)
{
    if (n == 0u) return ORDER_EMPTY;
    const uint32_t lg = 31u - (uint32_t)__builtin_clz(n);  // floor(log2 n)
    return lg >= (uint32_t)(ORDER_BUCKETS - 2) ? 0u : (uint32_t)(ORDER_BUCKETS - 2) - lg;
}

// The pool cursor: a 64-bit word among the eight words behind the list lengths (at the first 8-byte boundary) -- the
// pairs a pass WANTS may exceed 2^32 (a million polygons that each cross a large frame) even though no pool can
// hold them: ranges are formed in 64 bits and saturate, so that "how many records the pass wanted" stays meaningful
constexpr int ORDER_POOL = ORDER_BUCKETS;
__device__ __forceinline__ unsigned long long *order_pool_cursor(uint32_t *words_behind_counters)
{
    return reinterpret_cast<unsigned long long *>(((uintptr_t)(words_behind_counters + ORDER_POOL) + 7u) & ~(uintptr_t)7u);
}
__device__ __forceinline__ uint32_t saturate_u32(unsigned long long v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; }

// One workgroup's share of the sweep: tiles block * ORDER_THREADS .. of the pass `a` describes; `lengths`: the eight
// list lengths (a fused launch keeps them in its table entry, a per-frame launch behind the pass's counters).
__device__ __forceinline__ void order_body(const TileArgs &a, uint32_t *lengths, uint32_t n_tiles, uint32_t bits, uint32_t block)
{
    uint32_t *const tile_count = a.tile_count;
    WorkItem *const order = const_cast<WorkItem *>(a.order);
    unsigned long long *const pool_cursor = order_pool_cursor(tile_count + n_tiles);
    const uint32_t i = block * ORDER_THREADS + threadIdx.x, lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    const bool live = i < n_tiles;
    const uint32_t t = live ? scatter_tile(i, n_tiles, bits) : 0u;
    // (agent-scope load and store of the counter: inside k_chain they are what other workgroups' atomics made and will
    // count down, with no kernel boundary in between)
    const uint32_t n = live ? __hip_atomic_load(&tile_count[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const uint32_t bk = order_bucket(n);
    uint32_t mine = 0u, rank = 0u;
#pragma unroll
    for (int b = 0; b < ORDER_BUCKETS; b++) {
        const unsigned long long m = __ballot(live && bk == (uint32_t)b);
        if (lane == (uint32_t)b) mine = (uint32_t)__builtin_popcountll(m);
        if (bk == (uint32_t)b) rank = (uint32_t)__builtin_popcountll(m & below);
    }
    // the wave's tiles take one contiguous piece of the pool: running sum over the lanes, the waves' sums over
    // the workgroup, ONE atomic per workgroup (every tile of every frame of a group passes through this one
    // word, and a single address takes ~90 atomics per microsecond)
    unsigned long long incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long up = (unsigned long long)__shfl_up((long long)incl, d, 64);
        if ((int)lane >= d) incl += up;
    }
    const unsigned long long wave_total = (unsigned long long)__shfl((long long)incl, 63, 64);
    __shared__ unsigned long long s_total[ORDER_THREADS / 64], s_base;
    if (lane == 0u) s_total[threadIdx.x >> 6] = wave_total;
    __syncthreads();
    unsigned long long block_total = 0u, before = 0u;
#pragma unroll
    for (int w = 0; w < ORDER_THREADS / 64; w++) {
        if ((uint32_t)w < (threadIdx.x >> 6)) before += s_total[w];
        block_total += s_total[w];
    }
    // (the list atomics and the pool atomic are in flight together)
    uint32_t base = 0u;
    unsigned long long pool_base = 0u;
    if (lane < (uint32_t)ORDER_BUCKETS && mine) base = atomicAdd(&lengths[lane], mine);
    if (threadIdx.x == 0u && block_total) pool_base = atomicAdd(pool_cursor, block_total);
    if (threadIdx.x == 0u) s_base = pool_base;
    const uint32_t pos = (uint32_t)__shfl((int)base, (int)bk, 64) + rank;
    __syncthreads();
    const unsigned long long wave_base = s_base + before;
    const unsigned long long offset = wave_base + (incl - n);
    if (lane == 0u && wave_total && wave_base + wave_total > a.pool_cap) {
        // the pass wants more records than the pool holds: the frame is truncated, the host grows the pools to at
        // least what has been asked for so far and renders again
        atomicOr(a.err, (uint32_t)DE_BIN_OVERFLOW);
        *a.alarm = 1u;
        atomicMax(a.bin_need, saturate_u32(wave_base + wave_total));
        atomicMin(a.overflow_seq, a.pass_seq);
    }
    if (live) {
        WorkItem w;
        w.tile = t;
        // a tile whose range leaves the pool: what fits (k_bin writes no further)
        w.count = offset >= a.pool_cap ? 0u : min(n, a.pool_cap - (uint32_t)offset);
        w.offset = saturate_u32(offset);
        w.pad = 0u;
        order[(size_t)bk * n_tiles + pos] = w;
        // k_bin counts the tile's counter down from the END of its range: what an atomic returns is the record's
        // place in the pool (a place beyond the pool -- pools hold fewer than 2^31 records -- is not written); the
        // tile kernel's workgroup for the tile zeroes the counter for the set's next pass
        if (n) __hip_atomic_store(&tile_count[t], saturate_u32(offset + n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// `group` != null: blockIdx.y = frame of a group, whose counters, work lists and pool are in entry y of the tile
// kernel's argument table; else `one` describes the pass.
__global__ __launch_bounds__(ORDER_THREADS) void k_order(TileArgs one, uint32_t n_tiles, uint32_t bits, const TileArgs *__restrict__ group)
{
    // The chain in front of a tile kernel is a few hundred waves that share their SIMDs with six tile-kernel waves
    // each -- at equal priority every instruction of theirs waits its turn behind six others, and the NEXT pass's
    // tile kernel waits for them.  They are raised: the tile kernel hardly notices a few hundred short waves.
    __builtin_amdgcn_s_setprio(3);
    const TileArgs &a = group ? group[blockIdx.y] : one;
    // (a fused launch's list lengths live in its table entry, zeroed by the host with the table)
    uint32_t *const lengths = group ? const_cast<uint32_t *>(group[blockIdx.y].list_len) : a.tile_count + n_tiles;
    order_body(a, lengths, n_tiles, bits, blockIdx.x);
}

// -----------------------------------------------------------------------------------------
// k_tile
// -----------------------------------------------------------------------------------------

// LDS index of pixel (qx, qy) of a quadrant: block-major, so that during coverage lane l of a
// wave touches slot (block*64 + l): conflict-free 8-byte accesses.
// Inside a block the rows are exchanged by the block's column within its 32-pixel strip (row ^ column & 3): the shading
// phase reads a ROW of 32 keys per half-wave -- four blocks' eight-key pieces, which without the exchange start in the
// same bank (blocks are 512 bytes apart): a four-way conflict on every survivor read, 1.7 M of the headline launch's
// 2.8 M conflict cycles.  During coverage a wave's 64 lanes still cover exactly one block: any order of its rows is
// conflict-free there.
#ifndef TR_KEY_SWIZZLE
#define TR_KEY_SWIZZLE 1
#endif
__device__ __forceinline__ uint32_t key_row_swizzle(uint32_t tx)  // tx: x within the tile
{
    return TR_KEY_SWIZZLE ? ((tx >> 3) & 3u) << 3 : 0u;
}
template <int QUAD>
__device__ __forceinline__ uint32_t key_slot(uint32_t tx, uint32_t qy)
{
    constexpr int NBX = QUAD / 8, QPIX = QUAD * TILE_H;
    const uint32_t qx = tx % (uint32_t)QUAD;
    return (tx / (uint32_t)QUAD) * (uint32_t)QPIX + (((qy >> 3) * NBX + (qx >> 3)) << 6) + ((((qy & 7u) << 3) + (qx & 7u)) ^ key_row_swizzle(tx));
}

// Key layout of the SHARED resolve: row-major, the 8-pixel groups of a row rotated by the row number.
// Pixels x and x + 1 (x even) are neighbours -- a scan-line item reads both keys with one 16-byte
// access -- and a wave that sweeps a block column (lane = 8 x 8 pixels) still touches every bank once.
__device__ __forceinline__ uint32_t shared_key_slot(uint32_t tx, uint32_t qy)
{
    return qy * (uint32_t)TILE_W + (tx ^ ((qy & 7u) << 3));
}

// Streams the cleared value of a tile (scene.rs:128-137 folded into the render): z / shadow =
// f32::MIN, rgb = 0.  Whole-tile rows are whole cache lines; 16 B per lane when width % 16 == 0.
template <bool DEPTH, int TILE_THREADS, bool WINNER>
__device__ __forceinline__ void write_cleared_tile(const TileArgs &a, int32_t tile_x0, int32_t tile_y0,
                                                   bool with_depth)
{
    const int32_t W = (int32_t)a.frame.width, H = (int32_t)a.frame.height;
    const uint32_t tid = threadIdx.x;
    float *depth = DEPTH ? a.shadow : a.zbuf;
    if (a.aligned16) {
        const uint4 zmin = make_uint4(TR_F32_MIN_BITS, TR_F32_MIN_BITS, TR_F32_MIN_BITS, TR_F32_MIN_BITS);
        // depth: TILE_H rows x 32 pieces of 16 B
        for (uint32_t c = tid; c < (uint32_t)TILE_H * 32u && with_depth; c += (uint32_t)TILE_THREADS) {
            const int32_t y = tile_y0 + (int32_t)(c >> 5), x = tile_x0 + (int32_t)(c & 31u) * 4;
            if (x < W && y >= a.frame.band_y0 && y < a.frame.band_y1)
                *reinterpret_cast<uint4 *>(depth + (size_t)y * W + x) = zmin;
        }
        if (!DEPTH) {
            const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
            // colour: TILE_H rows x 24 pieces of 16 B
            for (uint32_t c = tid; c < (uint32_t)TILE_H * 24u; c += (uint32_t)TILE_THREADS) {
                const int32_t y = tile_y0 + (int32_t)(c / 24u);
                const int32_t xb = tile_x0 * 3 + (int32_t)(c % 24u) * 16;
                if (xb < W * 3 && y >= a.frame.band_y0 && y < a.frame.band_y1)
                    *reinterpret_cast<uint4 *>(a.fb + (size_t)(H - 1 - y) * W * 3 + xb) = zero;
            }
            if (WINNER && a.winner) {
                const uint4 none = make_uint4(NO_WINNER, NO_WINNER, NO_WINNER, NO_WINNER);
                for (uint32_t c = tid; c < (uint32_t)TILE_H * 32u; c += (uint32_t)TILE_THREADS) {
                    const int32_t y = tile_y0 + (int32_t)(c >> 5), x = tile_x0 + (int32_t)(c & 31u) * 4;
                    if (x < W && y >= a.frame.band_y0 && y < a.frame.band_y1)
                        *reinterpret_cast<uint4 *>(a.winner + (size_t)y * W + x) = none;
                }
            }
        }
    } else {
        for (uint32_t p = tid; p < (uint32_t)(TILE_W * TILE_H); p += (uint32_t)TILE_THREADS) {
            const int32_t y = tile_y0 + (int32_t)(p / TILE_W), x = tile_x0 + (int32_t)(p % TILE_W);
            if (x < W && y >= a.frame.band_y0 && y < a.frame.band_y1) {
                if (with_depth) depth[(size_t)y * W + x] = bits_f32(TR_F32_MIN_BITS);
                if (!DEPTH) {
                    uint8_t *px = a.fb + ((size_t)(H - 1 - y) * W + x) * 3;
                    px[0] = 0;
                    px[1] = 0;
                    px[2] = 0;
                    if (WINNER && a.winner) a.winner[(size_t)y * W + x] = NO_WINNER;
                }
            }
        }
    }
}

// Keeps a wave-uniform value in scalar registers from this point on.  A member of the argument table is an invariant
// scalar load the compiler prefers to REPEAT next to each use -- inside the shading loop that was 14 loads per step,
// each with its own s_waitcnt (band rows four times, the texture's size, pitch and address twice): a round trip to
// the scalar cache in the wave's serial path every few dozen instructions.  An empty asm makes the value opaque.
#define TR_KEEP_SCALAR(x) asm volatile("" : "+s"(x))

__device__ __forceinline__ int32_t bcast(uint32_t v, uint32_t lane)
{
    return __builtin_amdgcn_readlane((int)v, (int)lane);
}

// Waves per SIMD each k_tile variant is compiled for (= its VGPR budget: 8 -> 64, 6 -> 80, 5 -> 96,
// 4 -> 128).  Six for the light closures (walked both ways in round 1: five lost 5 %, seven lost 9 %);
// the closures that run two pixels' normalisations and a 3x3 inverse in packed registers need more
// room than that leaves them (80 VGPRs: 30-50 spilled dwords per lane, specular 446 -> 559 us on the
// x64 grid): normal_map and specular five waves (96), darboux four (128) -- profiles/r02_notes.md.
#ifndef TR_WPE_SPECULAR
#define TR_WPE_SPECULAR 5
#endif
#ifndef TR_WPE_DARBOUX
#define TR_WPE_DARBOUX 4
#endif
#ifndef TR_WPE_NORMAL_MAP
#define TR_WPE_NORMAL_MAP 6
#endif
#ifndef TR_WPE_LIGHT
#define TR_WPE_LIGHT 6
#endif
// The fused launches' kernels (GROUP) load their arguments from the table where they are used and spill far fewer
// scalar registers into vector ones (specular: 16-22 against 101-120), which is what the closures' budgets had been
// paying for: there specular fits six waves (x64 grid 349.9 -> 335.8 us per frame, 4096^2 45.0 -> 42.6) and darboux
// five (52.3 -> 48.6).
#ifndef TR_WPE_SPECULAR_GROUP
#define TR_WPE_SPECULAR_GROUP 6
#endif
#ifndef TR_WPE_DARBOUX_GROUP
#define TR_WPE_DARBOUX_GROUP 5
#endif
// Round 4: with the depth stores and the accumulate paths compiled out, the fused kernels of the light closures need
// 62 vector registers -- and a wave of the tile kernel spends nine tenths of its life waiting (LDS, the staged bin, its
// texels, two barriers): MORE waves per SIMD pay where round 1's register-starved kernels lost (seven: - 9 %).  What
// limits a four-wave tile's workgroups per compute unit beyond six is LDS: 16 KiB of keys + 8 KiB of resident records;
// with fewer resident records (lds_rec_bytes below: 72 at seven workgroups per CU, 40 at eight -- a tile of the reference's
// model holds six on average at 4096^2; larger bins take the chunked path as before) seven or eight fit.  Same box, 4096^2,
// k_tile per frame at 6 / 7 / 8: phong 25.3 / 23.9 / 23.6; shadow's colour pass 34.2 / 32.1 / 36.1 (64 registers: spills);
// its depth pass 29.7 / 28.1 / 27.1.  Only four-wave tiles of fused launches: tiles with eight waves are at their LDS
// limit already and lose registers (2048^2: 8.65 -> 9.05).
#ifndef TR_WPE_LIGHT_GROUP
#define TR_WPE_LIGHT_GROUP 8
#endif
#ifndef TR_WPE_SHADOW_GROUP
#define TR_WPE_SHADOW_GROUP 7
#endif
#ifndef TR_WPE_DEPTH_GROUP
#define TR_WPE_DEPTH_GROUP 8
#endif
#ifndef TR_WPE_LIT_GROUP
#define TR_WPE_LIT_GROUP 8
#endif
// (the fetch-only fragment stage of the lit texel path: specular 4096^2, columns, 6 / 7 / 8: 30.1 / 29.2 / 28.7 us per step;
// the x64 grid at 8192^2, SHARED resolve -- whose scan-line sums take another KiB of LDS, 31 resident records at eight --
// 203.4 / 198.4 / 203.0: the shared resolve stops at seven)
#ifndef TR_WPE_SHARED_MAX
#define TR_WPE_SHARED_MAX 7
#endif
// (occlusion's colour pass -- seventeen shadow-buffer lookups per fragment, 60 registers in the fused kernel: 6 / 7 / 8:
// 191 / 180 / 171 us per frame at 4096^2; the per-frame kernels of the light closures, which hold their arguments in
// registers: seven -- the unfused loop 34.2 -> 33.0)
#ifndef TR_WPE_OCCL_GROUP
#define TR_WPE_OCCL_GROUP 8
#endif
#ifndef TR_WPE_LIGHT4
#define TR_WPE_LIGHT4 7
#endif
// Closures whose tile kernels run their per-pixel arithmetic on plain scalar pairs instead of packed ones (tr_pk.h, f2s):
// a bit per FsKind (coverage, barycentrics, uv; a two-pixel closure itself is written for the packed type and stays packed).
#ifndef TR_SCALAR_FS
#define TR_SCALAR_FS ((1 << FS_DEFAULT) | (1 << FS_PHONG) | (1 << FS_LIT) | (1 << FS_SHADOW2) | (1 << FS_NORMAL_MAP) | (1 << FS_SPECULAR))
#endif
constexpr bool tile_scalar_pairs(int fs) { return ((TR_SCALAR_FS) >> fs & 1) != 0; }
constexpr int tile_waves_per_eu(int fs, int tile_waves, bool group = false, bool shared = false)
{
    int want = fs == FS_DARBOUX ? (group ? TR_WPE_DARBOUX_GROUP : TR_WPE_DARBOUX)
               : fs == FS_SPECULAR ? (group ? TR_WPE_SPECULAR_GROUP : TR_WPE_SPECULAR)
               : fs == FS_NORMAL_MAP ? TR_WPE_NORMAL_MAP : TR_WPE_LIGHT;
    if (!group && tile_waves == 4 && fs != FS_DARBOUX && fs != FS_SPECULAR && fs != FS_NORMAL_MAP) want = TR_WPE_LIGHT4;
    if (group && tile_waves == 4) {
        if (fs == FS_DEFAULT || fs == FS_PHONG) want = TR_WPE_LIGHT_GROUP;
        if (fs == FS_SHADOW2) want = TR_WPE_SHADOW_GROUP;
        if (fs == FS_DEPTH) want = TR_WPE_DEPTH_GROUP;
        if (fs == FS_LIT) want = TR_WPE_LIT_GROUP;
        if (fs == FS_OCCLUSION2) want = TR_WPE_OCCL_GROUP;
        if (shared && want > TR_WPE_SHARED_MAX) want = TR_WPE_SHARED_MAX;
    }
    // sixteen waves per tile: a workgroup brings four waves to every SIMD, so 8 (two workgroups per
    // CU) or 4 (one) are the only useful budgets
    if (tile_waves == 16) return want >= 6 ? 8 : 4;
    return want;
}
// Bytes of a tile's bin that stay resident in LDS: what the workgroups of a compute unit leave of its 160 KiB beside
// their keys (16 KiB) and, in the shared resolve, the scan-line sums (4 B per thread + per wave).  Four-wave tiles:
// as many workgroups per CU as waves per SIMD.
constexpr int lds_rec_bytes_for(int tile_waves, int waves_per_eu, bool shared)
{
    if (tile_waves != 4) return lds_rec_bytes(tile_waves);
    // (a workgroup's share, rounded down to a multiple of 2 560 B: whatever the allocation granule -- 512 B, 1 280 B --
    // the workgroups' rounded-up allocations then still fit)
    const int per_wg = 160 * 1024 / waves_per_eu / 2560 * 2560;
    const int left = per_wg - TILE_W * TILE_H * 8 - (shared ? 4 * 64 * tile_waves + 4 * tile_waves : 8) - 64;
    const int cap = lds_rec_bytes(4);
    return (left < cap ? left : cap) / 16 * 16;
}

// Keys of the SHARED resolve (below): one 64-bit word per pixel, compared as an unsigned integer by
// an LDS atomic maximum.  High half: the depth as an order-preserving integer (-0.0 folded onto +0.0:
// the reference's `z <= zbuf` treats them as equal).  Low half: who wins among equal depths --
//   a colour fragment: (2^20 - 1 - polygon id) << 12 | bin slot + 1   -> the lowest polygon index
//   what the buffer held before the pass: all ones in a colour pass (`z <= zbuf` rejects: it beats
//   every fragment of equal depth), zero in a depth pass (`z >= shadow` accepts: it loses)
// Field limits: polygon ids below 2^20, bin slots below 4093 (launch_tile falls back to the column
// mode beyond them).
constexpr uint32_t SHARED_MAX_POLYGONS = 1u << 20, SHARED_MAX_SLOTS = 4093u;
// Shared resolve: small polygons (pair_masks, tr_shaders.h) are resolved as scan-line items (k_tile);
// 0 = every polygon is visited by a whole wave (round 2's only form).
#ifndef TR_SCAN_ITEMS
#define TR_SCAN_ITEMS 1
#endif
// (round 4: the survivor's key read whole, profiles/r04_notes.md)
#ifndef TR_KEY_B64
#define TR_KEY_B64 1
#endif
// (a depth pass stores the depth its resolve compared: k_tile, shade_steps)
#ifndef TR_DEPTH_FROM_KEYS
#define TR_DEPTH_FROM_KEYS 1
#endif
constexpr bool SCAN_ITEMS = TR_SCAN_ITEMS != 0;
// Measurement builds only (wrong frames): leave a phase out to time the others (profiles/r03_notes.md)
#ifndef TR_DBG_SKIP
#define TR_DBG_SKIP 0  // 1: no fragment stage, 2: no item passes, 4: no visits
#endif
__device__ __forceinline__ uint32_t depth_order_bits(float z)
{
    uint32_t b = __float_as_uint(z);
    b = b == 0x80000000u ? 0u : b;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// SHARED = false: every wave owns a column of the tile (32 / 16 / 8 pixels wide) and sees the whole
// bin; its pixels' keys are private, plain LDS reads and writes.
// SHARED = true: every wave sees the whole tile and owns an interleaved share of the BIN (record j
// belongs to wave j mod waves); keys are shared and updated with 64-bit LDS atomic maxima.  Waves then
// carry equal loads wherever the polygons cluster (the eyes of a head at 800^2: one 8-pixel column
// held 80 of a tile's 120 polygons and its wave ran 20 us while the others idled), and a polygon is
// visited once per tile instead of once per column it touches.
#define TR_TILE_KERNEL_ATTRS \
    __global__ __launch_bounds__(64 * TILE_WAVES) \
        __attribute__((amdgpu_waves_per_eu(tile_waves_per_eu(FS, TILE_WAVES, MODE != 0, SHARED), tile_waves_per_eu(FS, TILE_WAVES, MODE != 0, SHARED))))

// GROUP = false: one frame, arguments by value.
// GROUP = true: a fused launch over a group of n_frames frames (tr_scene_render_frames): workgroup b renders
// entry b / n_frames of the work list of frame b % n_frames, whose arguments are entry b % n_frames of a
// table in device memory -- so the heavy tiles of ALL the frames come first and the light and empty ones
// fill the slots they free: one frame's drain (a third of a lone 4096^2 launch runs on a machine that is
// emptying, and a small frame never fills it at all) is the next frame's start.
// MODE 0: one frame, arguments by value; what the pass writes is TileArgs::store, looked up at run time (the depth-only
// repeat of a colour pass is such a launch).  MODE 1 / 2: a fused launch; 2 = transient depth, decided at COMPILE time --
// a run-time flag skipped the stores but kept the arithmetic behind them (the survivors' z, the depth addresses), and
// with it the gain (same box: 28.5 -> 28.3 us per frame against 27.4 for the compiled-out form).
template <int FS, int TILE_WAVES, bool SHARED, int MODE>
TR_TILE_KERNEL_ATTRS void k_tile(TileArgs args, const TileArgs *__restrict__ table, uint32_t n_frames)
{
    constexpr bool GROUP = MODE != 0;
    static_assert(TILE_WAVES == 4 || TILE_WAVES == 8 || TILE_WAVES == 16, "a wave covers a 32, 16 or 8 pixel wide column of the tile");
    constexpr int TILE_THREADS = 64 * TILE_WAVES;
    constexpr int QUAD_COLUMN = TILE_W / TILE_WAVES;  // width of a wave's column when columns are owned
    constexpr int WAVES_PER_STRIP = TILE_WAVES / (TILE_W / STRIP);  // shading: waves sharing a 32-pixel strip
    constexpr int STRIP_ROWS = TILE_H / WAVES_PER_STRIP;            // rows of the strip each of them shades
    static_assert(STRIP_ROWS % 4 == 0, "a shading step covers four rows");
    constexpr bool DEPTH = (FS == FS_DEPTH);
    constexpr int P = (FS == FS_DARBOUX) ? REC_PIECES_LARGE : REC_PIECES_SMALL;
    constexpr int NMAX = lds_rec_bytes_for(TILE_WAVES, tile_waves_per_eu(FS, TILE_WAVES, MODE != 0, SHARED), SHARED) / (P * 16);  // records resident in LDS
    // the per-pixel arithmetic on plain pairs of scalars, or packed (tr_pk.h): per closure, measured
    using V2 = std::conditional_t<tile_scalar_pairs(FS), f2s, f2>;

    // Per pixel, column mode: .x = z of the best fragment so far (f32 bits; compared as floats, so
    // -0.0 and +0.0 tie exactly like the reference's `z <= zbuf`), .y = its bin slot + 1 (0 = "what
    // the buffer held before this pass").  Shared mode: the 64-bit key described above (.y = depth
    // order bits, .x = tie-break word whose low 12 bits are the bin slot + 1).
    __shared__ uint2 s_key[TILE_W * TILE_H];
    __shared__ uint4 s_rec[NMAX * P];
    // shared resolve: running sums of the scan-line items per thread, and per wave
    __shared__ uint32_t s_incl[SHARED ? TILE_THREADS : 1];
    __shared__ uint32_t s_wtot[SHARED ? TILE_WAVES : 1];

    // One workgroup per tile and frame, in the order of the work lists (k_order): blockIdx = b * n_frames + f is
    // entry b of frame f's lists walked heaviest first, so that the heavy tiles of ALL frames of a group are
    // dispatched first -- the long, VALU-bound ones (longest-processing-time-first packing by the hardware's
    // own dispatcher, and the machine is full of them from the first microsecond) -- then the light ones and
    // the empty tiles, whose workgroups only check flags or stream the cleared colour of a fresh frame: short
    // work that fills the slots the busy tiles free.  Measured alternatives: interleaving the two kinds, or
    // letting the busy blocks issue those stores themselves, was 5-10 % slower (profiles/r01_notes.md); fewer
    // workgroups that each take several tiles (a loop around this body) cost registers -- 7-13 spilled
    // vector registers in the light closures' kernels -- and 4-12 % (profiles/r03_notes.md).  Any order is correct.
    const uint32_t n_fr = GROUP ? n_frames : 1u;
    const uint32_t frame_of_group = blockIdx.x % n_fr;
    // (the table entry is not copied: a member is loaded where it is used, arrays are indexed in place)
    // (a fused-mode launch of ONE frame -- a per-frame pass that starts from cleared targets, launch_tile's `fused_single` --
    // has no table in memory: its arguments are `args`, the kernel's first parameter, read in place in the kernel-argument
    // segment like a table entry)
    const TileArgs &a = GROUP ? *(const TileArgs *)(table ? (constant_ptr<TileArgs>)table + frame_of_group
                                                         : (constant_ptr<TileArgs>)__builtin_amdgcn_kernarg_segment_ptr())
                              : args;
    // a fused launch has the eight lengths in its table entry, a per-frame launch (fused-mode or not) behind the pass's
    // counters (one more scalar load)
    const bool table_lengths = GROUP && table != nullptr;
    // (every frame of a fused launch starts from cleared targets and has no winner tap -- run_group, tr_scene.cpp: known
    // when the kernel is compiled, so the accumulate paths and the tap's stores are not even there)
    const bool fresh = GROUP || a.fresh != 0u;
    const bool st_z = FS == FS_DEPTH || (MODE == 0 ? (a.store & TR_STORE_DEPTH) != 0u : MODE == 1);
    const bool st_c = FS == FS_DEPTH || MODE != 0 || (a.store & TR_STORE_COLOR) != 0u;
    const uint32_t n_tiles = a.frame.ntx * a.frame.nty;
    // Work units (EMPTY_CHUNK above): workgroup blockIdx / frames of a frame takes that unit.  (A loop that lets a
    // workgroup take further units -- so that ANY grid would do and the host could size it by an estimate -- costs the
    // body registers it does not have: 48-68 bytes of spills per lane in the light closures' kernels, 22.7 -> 29.1 us per
    // frame with one workgroup per tile and 23.6 with the estimate; profiles/r04_notes.md.  So the grid is exact or full.)
    const uint32_t tid = threadIdx.x;
    __shared__ uint32_t s_stale[EMPTY_CHUNK];  // empty chunk: tile + 1 of the chunk's tiles whose colour must be written, else 0
    uint32_t entry = blockIdx.x / n_fr;
    uint32_t list = 0u;
    {
        // which list, which entry: a fused launch has the eight lengths in its table entry, a per-frame launch
        // behind the pass's counters (one more scalar load)
        constant_ptr<uint32_t> lengths = table_lengths ? (constant_ptr<uint32_t>)a.list_len : (constant_ptr<uint32_t>)a.tile_count + n_tiles;
#pragma unroll
        for (int b = 0; b < ORDER_BUCKETS - 1; b++) {
            const uint32_t len = lengths[b];
            if (list == (uint32_t)b && entry >= len) {
                entry -= len;
                list = (uint32_t)b + 1u;
            }
        }
    }
    uint64_t *const stamps = frame_of_group == 0u ? a.stamps : nullptr;  // the diagnostic stamps follow a group's first frame
    if (list == (uint32_t)ORDER_EMPTY) {
        // ---- a chunk of the empty list: EMPTY_CHUNK tiles without polygons (work unit `entry` behind the busy tiles) ----
        constant_ptr<uint32_t> lens = table_lengths ? (constant_ptr<uint32_t>)a.list_len : (constant_ptr<uint32_t>)a.tile_count + n_tiles;
        const uint32_t n_empty = lens[ORDER_EMPTY], first = entry * EMPTY_CHUNK;
        if (first >= n_empty || !fresh) return;  // (surplus workgroup of a grid sized for the worst case; an accumulating render leaves empty tiles alone)
        const uint32_t cnt = min(EMPTY_CHUNK, n_empty - first);
        // An empty tile of a cleared frame: its colour is zeros -- stored unless the tile's memory already holds them
        // (it was empty the last time it was written, too: most of a frame, most of the time); its z stays unwritten
        // behind the tile's fast-clear flag (depth passes write their f32::MIN).  (A pass that leaves the colour or the
        // depth out -- TileArgs::store -- leaves the respective memory and flags of the tile alone.)
        uint32_t my_tile = 0u;
        bool my_stale = false;
        if (tid < cnt) {
            my_tile = a.order[(size_t)ORDER_EMPTY * n_tiles + first + tid].tile;
            my_stale = st_c && (DEPTH ? a.zclean == nullptr : (a.fbclean == nullptr || gload(a.fbclean + my_tile) == 0u));
            s_stale[tid] = my_stale ? my_tile + 1u : 0u;
        }
        // (every flag has been read before the first one is raised)
        __syncthreads();
        for (uint32_t i = 0; i < cnt; i++) {
            const uint32_t t1 = s_stale[i];
            if (t1 == 0u) continue;
            const uint32_t t = t1 - 1u;
            write_cleared_tile<DEPTH, TILE_THREADS, !GROUP>(a, (int32_t)(t % a.frame.ntx) * TILE_W,
                                      (a.frame.ty_base + (int32_t)(t / a.frame.ntx)) * TILE_H, a.zclean == nullptr);
        }
        if (tid < cnt) {
            if (!DEPTH && my_stale && a.fbclean) a.fbclean[my_tile] = 1u;
            if (a.zclean && st_z) a.zclean[my_tile] = 1u;
        }
        return;
    }
    const WorkItem work = a.order[(size_t)list * n_tiles + entry];
    const uint32_t tile = work.tile;
    uint32_t n = work.count;
    // (k_bin has counted the tile's counter down to the start of its range; the set's next pass counts from zero)
    if (tid == 0u) a.tile_count[tile] = 0u;
    if (n == 0u) {
        if (fresh) {
            // an empty tile of a cleared frame: its colour is zeros -- stored unless the tile's memory
            // already holds them (it was empty the last time it was written, too: most of a frame,
            // most of the time); its z stays unwritten behind the tile's fast-clear flag (depth passes
            // write their f32::MIN).  (A pass that leaves the colour or the depth out -- TileArgs::store -- leaves the
            // respective memory and flags of the tile alone.)
            const bool stale = st_c && (DEPTH || a.fbclean == nullptr || a.fbclean[tile] == 0u);
            // every wave has read the flag before the first one raises it (a wave that came late would find
            // it up and leave its share of the tile unwritten)
            __syncthreads();
            if (stale) {
                write_cleared_tile<DEPTH, TILE_THREADS, !GROUP>(a, (int32_t)(tile % a.frame.ntx) * TILE_W,
                                          (a.frame.ty_base + (int32_t)(tile / a.frame.ntx)) * TILE_H, a.zclean == nullptr);
                if (!DEPTH && tid == 0u && a.fbclean) a.fbclean[tile] = 1u;
            }
            if (tid == 0u && a.zclean && st_z) a.zclean[tile] = 1u;
        }
        return;
    }
    const int32_t tx = (int32_t)(tile % a.frame.ntx);
    const int32_t ty = a.frame.ty_base + (int32_t)(tile / a.frame.ntx);
    const int32_t tile_x0 = tx * TILE_W, tile_y0 = ty * TILE_H;
    int32_t W = (int32_t)a.frame.width, H = (int32_t)a.frame.height;
    // what every shading step reads of the frame's arguments: into scalar registers once (TR_KEEP_SCALAR)
    int32_t band_y0 = a.frame.band_y0, band_y1 = a.frame.band_y1;
    uint32_t aligned4 = a.aligned4;
    uint8_t *fb = a.fb;
    DevTextures tex = a.tex;
    TR_KEEP_SCALAR(W); TR_KEEP_SCALAR(H); TR_KEEP_SCALAR(band_y0); TR_KEEP_SCALAR(band_y1); TR_KEEP_SCALAR(aligned4); TR_KEEP_SCALAR(fb);
    TR_KEEP_SCALAR(tex.packed); TR_KEEP_SCALAR(tex.packed_bpr); TR_KEEP_SCALAR(tex.w[0]); TR_KEEP_SCALAR(tex.h[0]);
    // the wave index is uniform: say so, so that quadrant bounds and the block loop stay scalar
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63u;

    // Diagnostic builds of a scene (TR_OPT_TILE_STAMPS) record when each busy tile ran; the stamps
    // go to a buffer of their own and nothing is computed from them.
    uint64_t t_start = 0, t_staged = 0;
    if (stamps) t_start = wall_clock64();

    float *depth = DEPTH ? a.shadow : a.zbuf;
    // is every z of this tile logically f32::MIN (cleared frame, or fast-clear flag still set)?
    // then nothing is read and every live z is written, after which the flag is down
    const bool zfresh = fresh || (a.zclean && a.zclean[tile] != 0u);
    const int32_t qy0 = tile_y0;
    constexpr uint32_t PREV_TAG = DEPTH ? 0u : 0xFFFFFFFFu;  // shared mode: tie-break word of the buffer's old content
    // A launch compiled for the shared resolve still gives tiles with few polygons to the column form:
    // sharing deals whole polygons to waves, and four polygons that each cross the entire tile would
    // occupy four waves for 16 column pairs each while the others idle (2048^2: such tiles were the
    // slowest of the frame, 19-21 us); columns split exactly that work evenly.
    const bool shared_tile = SHARED && n >= 2u * (uint32_t)TILE_WAVES && n <= SHARED_MAX_SLOTS;  // (the key's slot field)
    const int32_t lx = (int32_t)(lane & 7u), ly = (int32_t)(lane >> 3);
    const uint4 *bin = reinterpret_cast<const uint4 *>(a.bins) + (size_t)work.offset * P;  // the tile's records in the pool
    const bool resident = n <= (uint32_t)NMAX;  // the whole bin stays in LDS through shading

    // ---- initial keys, coverage + depth resolve: one body for both forms ------------------------
    auto resolve = [&](auto shared_tag) {
        constexpr bool SH = decltype(shared_tag)::value;
        constexpr int QUAD = SH ? TILE_W : QUAD_COLUMN;  // width of the region a wave resolves
        constexpr int NBX = QUAD / 8;                    // 8x8 lane blocks per row of it
        constexpr int QPIX = QUAD * TILE_H;
        const int32_t qx0 = SH ? tile_x0 : tile_x0 + (int32_t)wave * QUAD;
        uint2 *wkey = SH ? s_key : s_key + wave * QPIX;
        // this lane's key in block column 0 (block row 0; block row 1 follows SH ? 8 rows : NBX blocks later)
        const uint32_t key_lane = SH ? shared_key_slot((uint32_t)lx, (uint32_t)ly) : lane;
        // ---- initial keys -------------------------------------------------------------------
        if (SH) {
            // the waves initialise the tile's blocks between them
            for (int b = (int)wave; b < NBX * NBY; b += TILE_WAVES) {
                uint32_t zb = TR_F32_MIN_BITS;
                if (!zfresh) {
                    const int32_t px = qx0 + (b % NBX) * 8 + lx, py = qy0 + (b / NBX) * 8 + ly;
                    if (px < W && py >= band_y0 && py < band_y1)
                        zb = __float_as_uint(gload(depth + ((size_t)py * W + px)));
                }
                wkey[shared_key_slot((uint32_t)((b % NBX) * 8 + lx), (uint32_t)((b / NBX) * 8 + ly))] =
                    make_uint2(PREV_TAG, depth_order_bits(__uint_as_float(zb)));
            }
        } else {
    #pragma unroll
            for (int b = 0; b < NBX * NBY; b++) {
                uint32_t zb = TR_F32_MIN_BITS;
                if (!zfresh) {
                    const int32_t px = qx0 + (b % NBX) * 8 + lx, py = qy0 + (b / NBX) * 8 + ly;
                    if (px < W && py >= band_y0 && py < band_y1)
                        zb = __float_as_uint(gload(depth + ((size_t)py * W + px)));
                }
                wkey[(b << 6) + (lane ^ key_row_swizzle((uint32_t)((int32_t)wave * QUAD + (b % NBX) * 8)))] = make_uint2(zb, 0u);
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
            for (uint32_t q = tid; q < m * P; q += (uint32_t)TILE_THREADS) s_rec[q] = gload(bin + ((size_t)c0 * P + q));
            __syncthreads();
            if (stamps && c0 == 0u) t_staged = wall_clock64();

            // column mode: every wave takes all m records, 64 per round; shared mode: record jj belongs to
            // wave jj mod TILE_WAVES, a round covers 64 * TILE_WAVES records
            for (uint32_t j0 = 0; j0 < m; j0 += SH ? 64u * (uint32_t)TILE_WAVES : 64u) {
                const uint32_t jj = SH ? j0 + lane * (uint32_t)TILE_WAVES + wave : j0 + lane;
                uint4 r0 = make_uint4(1u, 0u, 0u, 0u), r1 = make_uint4(0, 0, 0, 0), r2 = r1, r3 = r1;  // (no record: an empty box)
                uint32_t ry = 0u;
                if (jj < m) {
                    r0 = s_rec[jj * P + 0];
                    r1 = s_rec[jj * P + 1];
                    r2 = s_rec[jj * P + 2];
                    r3 = s_rec[jj * P + 3];
                    ry = s_rec[jj * P + (P - 1)].w;
                }
                // piece 0: the polygon's clamped box and what k_setup found out about it inside THIS tile
                // (pair_masks, tr_shaders.h): the cells of a small pair, the block columns of a large one
                const int32_t rbx0 = (int32_t)(r0.x & 0xFFFFu), rbx1 = (int32_t)(r0.x >> 16);
                const int32_t rby0 = (int32_t)(r0.y & 0xFFFFu), rby1 = (int32_t)(r0.y >> 16);
                const PairBox pb = pair_box(rbx0, rbx1, rby0, rby1, tile_x0, qy0);
                const bool some = rbx0 <= rbx1;
                // Shared form: a SMALL polygon -- its box inside the tile at most SCAN_MAX_CHUNKS 8-pixel chunks
                // wide -- is not visited by a whole wave (135 instructions of broadcast and prologue for a
                // polygon of a hundred pixels, then 128 pixel slots per step of which a third lie in its
                // box): the live cells of its box (one row x one chunk each) become ITEMS, and the items of
                // all the tile's small polygons are dealt to the lanes of all its waves (below).
                const bool small = SH && SCAN_ITEMS && some && pb.nch <= SCAN_MAX_CHUNKS;
                const uint32_t items = small ? (uint32_t)(__popc(r0.z) + __popc(r0.w)) : 0u;
                // block columns of this wave's region (bit i = column pair i: block rows 0 / 1) with a live
                // block: what a visit iterates over
                uint32_t lmask = (some && !small) ? pair_block_columns(r0.z, r0.w, SHARED && SCAN_ITEMS, pb, tile_x0) : 0u;
                if (!SH) lmask = (lmask >> (NBX * wave)) & ((1u << NBX) - 1u);
                // The polygon's part of to_barycentric_coord (scene.rs:178-187), for the 64 records
                // of this round at once (lane l = record l).  Orientation is normalised so that
                // cross.z > 0: negating a0, a1, b0, b1 flips the sign of cross.x and cross.y exactly,
                // and dividing them by -cross.z (reciprocal -y) gives bit-identical quotients, so one
                // branch-free form of the inside test serves both windings.
                float la0 = __uint_as_float(r1.z), la1 = __uint_as_float(r2.x);
                float lb0 = __uint_as_float(r1.w), lb1 = __uint_as_float(r2.y);
                float lcz = la0 * lb1 - la1 * lb0;
                float lry = __uint_as_float(ry);
                if (lcz < 0.0f) {
                    la0 = -la0; la1 = -la1; lb0 = -lb0; lb1 = -lb1;
                    lcz = -lcz;
                    lry = -lry;
                }
                unsigned long long todo = (TR_DBG_SKIP & 4) ? 0ull : __ballot(lmask != 0u);
                while (todo) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const uint32_t boxx = (uint32_t)bcast(r0.x, l), boxy = (uint32_t)bcast(r0.y, l);
                    const int32_t bx0 = imax((int32_t)(boxx & 0xFFFFu), qx0), bx1 = imin((int32_t)(boxx >> 16), qx0 + QUAD - 1);
                    const int32_t by0 = imax((int32_t)(boxy & 0xFFFFu), qy0), by1 = imin((int32_t)(boxy >> 16), qy0 + TILE_H - 1);
                    const int32_t x0 = bcast(r1.x, l), y0 = bcast(r1.y, l);
                    const float z0 = __int_as_float(bcast(r2.z, l)), z1 = __int_as_float(bcast(r2.w, l));
                    const float z2 = __int_as_float(bcast(r3.x, l));
                    const uint32_t id = (uint32_t)bcast(r3.y, l);
                    const uint32_t slot1 = SH ? c0 + j0 + l * (uint32_t)TILE_WAVES + wave + 1u : c0 + j0 + l + 1u;
                    uint32_t cols = (uint32_t)bcast(lmask, l);
                    Edge2T<V2> e;
                    e.a0 = splat2v<V2>(__int_as_float(bcast(__float_as_uint(la0), l)));
                    e.a1 = splat2v<V2>(__int_as_float(bcast(__float_as_uint(la1), l)));
                    e.b0 = splat2v<V2>(__int_as_float(bcast(__float_as_uint(lb0), l)));
                    e.b1 = splat2v<V2>(__int_as_float(bcast(__float_as_uint(lb1), l)));
                    const float cz = __int_as_float(bcast(__float_as_uint(lcz), l));
                    e.cz = splat2v<V2>(cz);
                    e.y = splat2v<V2>(__int_as_float(bcast(__float_as_uint(lry), l)));
                    // this lane's two pixels: (px, pya) in block row 0 and (px, pyb) in block row 1
                    const int32_t pya = qy0 + ly, pyb = qy0 + 8 + ly;
                    const V2 b2 = mk2v<V2>((float)isub(y0, pya), (float)isub(y0, pyb));
                    // the inside test (scene.rs:245-247 on to_barycentric_coord's cross products, divided
                    // by cross.z > 0): cross.x >= 0, cross.y >= 0, cross.x + cross.y <= cross.z.  The last is
                    // taken as cross.z - (cross.x + cross.y) >= 0 (same truth value: a difference of two
                    // floats has the sign of the exact difference) so that one three-way minimum and one
                    // compare decide a pixel; a row outside the clamped box gets -inf for cross.z.
                    const bool rowa = pya >= by0 && pya <= by1, rowb = pyb >= by0 && pyb <= by1;
                    const V2 czp = mk2v<V2>(rowa ? cz : -__builtin_inff(), rowb ? cz : -__builtin_inff());
                    while (cols) {
                        const int32_t ib = (int32_t)__builtin_ctz(cols);
                        cols &= cols - 1u;
                        // the two pixels' current keys, requested before the arithmetic that decides
                        // whether they are needed (LDS latency hidden inside the wave)
                        uint2 *slot_a = SH ? wkey + (key_lane ^ ((uint32_t)ib << 3))
                                           : wkey + (((uint32_t)ib << 6) + (lane ^ key_row_swizzle((uint32_t)((int32_t)wave * QUAD + ib * 8))));
                        uint2 *slot_b = slot_a + (SH ? 8 * TILE_W : NBX << 6);
                        const uint2 cur_a = *slot_a, cur_b = *slot_b;
                        const int32_t px = qx0 + ib * 8 + lx;
                        const bool inx = (uint32_t)isub(px, bx0) <= (uint32_t)isub(bx1, bx0);
                        V2 cx, cy;
                        edge_cross2(e, splat2v<V2>((float)isub(x0, px)), b2, cx, cy);
                        const V2 rest = czp - (cx + cy);
                        const bool hita = inx && __builtin_fminf(__builtin_fminf(cx.x, cy.x), rest.x) >= 0.0f;
                        const bool hitb = inx && __builtin_fminf(__builtin_fminf(cx.y, cy.y), rest.y) >= 0.0f;
                        if (SH) {
                            if (hita || hitb) {
                                // shared keys: the candidate (depth order bits, tie-break word) against what the
                                // pixel holds now -- a plain read, possibly stale, but keys only grow, so a
                                // candidate that does not beat it can never win -- then one atomic maximum
                                const Bary2T<V2> bar = barycentric2_for_compare(cx, cy, e);
                                const V2 z = dot3_2(bar.x, bar.y, bar.z, splat2v<V2>(z0), splat2v<V2>(z1), splat2v<V2>(z2));
                                const uint32_t tag = (((SHARED_MAX_POLYGONS - 1u) - id) << 12) | slot1;
                                const unsigned long long ka = ((unsigned long long)depth_order_bits(z.x) << 32) | tag;
                                const unsigned long long kb = ((unsigned long long)depth_order_bits(z.y) << 32) | tag;
                                const unsigned long long ca_ = ((unsigned long long)cur_a.y << 32) | cur_a.x;
                                const unsigned long long cb_ = ((unsigned long long)cur_b.y << 32) | cur_b.x;
                                if (hita && ka > ca_)
                                    __hip_atomic_fetch_max(reinterpret_cast<unsigned long long *>(slot_a), ka, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (hitb && kb > cb_)
                                    __hip_atomic_fetch_max(reinterpret_cast<unsigned long long *>(slot_b), kb, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        } else if (hita || hitb) {
                            // depth of both fragments; only compared here (the survivor's stored z is
                            // recomputed with exact zero signs when it is shaded)
                            const Bary2T<V2> bar = barycentric2_for_compare(cx, cy, e);
                            const V2 z = dot3_2(bar.x, bar.y, bar.z, splat2v<V2>(z0), splat2v<V2>(z1), splat2v<V2>(z2));
                            // both comparisons first, branch-free, so that the two key reads above are
                            // consumed together after the arithmetic; equal depths (shared vertices and
                            // edges, or the buffer's previous content) are the rare divergent path
                            const float zca = __uint_as_float(cur_a.x), zcb = __uint_as_float(cur_b.x);
                            bool wina = hita && z.x > zca, winb = hitb && z.y > zcb;
                            const bool tiea = hita && z.x == zca, tieb = hitb && z.y == zcb;
                            if (tiea || tieb) {
                                // equal depth: the buffer's previous content beats a colour fragment
                                // (`z <= zbuf` rejects) and loses to a depth fragment (`z >= shadow`
                                // accepts); between two fragments of this pass the polygon index decides
    #pragma unroll
                                for (int h = 0; h < 2; h++) {
                                    if (!(h == 0 ? tiea : tieb)) continue;
                                    const uint32_t cs = h == 0 ? cur_a.y : cur_b.y;
                                    bool w = DEPTH;
                                    if (cs != 0u) {
                                        const uint32_t cur_id = resident ? s_rec[(cs - 1u) * P + 3].y
                                                                         : gload(bin + ((size_t)(cs - 1u) * P + 3)).y;
                                        w = DEPTH ? id > cur_id : id < cur_id;
                                    }
                                    if (h == 0) wina = w; else winb = w;
                                }
                            }
                            if (wina) *slot_a = make_uint2(__float_as_uint(z.x), slot1);
                            if (winb) *slot_b = make_uint2(__float_as_uint(z.y), slot1);
                        }
                    }
                }

                if (SH && SCAN_ITEMS) {
                    // ---- scan-line items of the round's small polygons --------------------------------
                    // Items are numbered over the whole workgroup (thread order), 64 of them make a pass and
                    // pass p belongs to wave p mod TILE_WAVES: every wave gets the same share wherever the
                    // polygons sit in the bin.  A lane finds the record its item belongs to by a binary search
                    // in the running sums and reads the record from LDS itself -- no broadcast at all -- then
                    // tests the item's 8 pixels two at a time: (x, x + 1) in packed arithmetic with exactly
                    // the operations of the visit above (scene.rs:178-187, 245-247, shader.rs:174).
                    uint32_t incl = items;
    #pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
                        if ((int)lane >= d) incl += up;
                    }
                    if (lane == 63u) s_wtot[wave] = incl;
                    __syncthreads();  // (also: every wave is done with the previous round's sums)
                    uint32_t before = 0u, total = 0u;
    #pragma unroll
                    for (int w = 0; w < TILE_WAVES; w++) {
                        const uint32_t t = s_wtot[w];
                        before += (uint32_t)w < wave ? t : 0u;
                        total += t;
                    }
                    total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
                    s_incl[tid] = before + incl;
                    __syncthreads();
                    for (uint32_t pass = wave; !(TR_DBG_SKIP & 2) && pass * 64u < total; pass += (uint32_t)TILE_WAVES) {
                        const uint32_t item = pass * 64u + lane;
                        const bool act = item < total;
                        // owner = the first thread whose running sum exceeds the item's number
                        uint32_t lo = 0u, hi = (uint32_t)TILE_THREADS - 1u;
    #pragma unroll
                        for (int i = 0; i < 6 + (TILE_WAVES == 4 ? 2 : TILE_WAVES == 8 ? 3 : 4); i++) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (s_incl[mid] > item)
                                hi = mid;
                            else
                                lo = mid + 1u;
                        }
                        const uint32_t q = item - (lo ? s_incl[lo - 1u] : 0u);
                        // thread lo = lane lo & 63 of wave lo >> 6 holds record j0 + lane * TILE_WAVES + wave
                        const uint32_t rec = act ? j0 + (lo & 63u) * (uint32_t)TILE_WAVES + (lo >> 6) : 0u;
                        const uint4 *const R = s_rec + mul24(rec, (uint32_t)P);
                        const uint4 q0 = R[0], q1 = R[1], q2 = R[2];
                        const uint2 q3 = *reinterpret_cast<const uint2 *>(R + 3);
                        const float ryv = __uint_as_float(R[P - 1].w);
                        // the item = the q-th live cell of the pair's mask: row `cell / 4` of the box inside the
                        // tile, chunk `cell % 4` = pixels xs .. xs + 7 of that row (xs even)
                        const PairBox ib = pair_box((int32_t)(q0.x & 0xFFFFu), (int32_t)(q0.x >> 16), (int32_t)(q0.y & 0xFFFFu),
                                                    (int32_t)(q0.y >> 16), tile_x0, qy0);
                        uint32_t cell;
                        {
                            const uint32_t nlo = (uint32_t)__popc(q0.z);
                            const bool upper = q >= nlo;
                            uint32_t w = upper ? q0.w : q0.z, k = upper ? q - nlo : q;
                            cell = upper ? 32u : 0u;
    #pragma unroll
                            for (int sh = 16; sh >= 1; sh >>= 1) {
                                const uint32_t below = (uint32_t)__popc(w & ((1u << sh) - 1u));
                                const bool up = k >= below;
                                k -= up ? below : 0u;
                                w = up ? w >> sh : w;
                                cell += up ? (uint32_t)sh : 0u;
                            }
                        }
                        const int32_t py = ib.ay0 + (int32_t)(cell >> 2), xs = ib.xs + 8 * (int32_t)(cell & 3u);
                        // pixels of the chunk inside the box, as a bit mask (bit i = pixel xs + i)
                        const uint32_t first = (uint32_t)imax(isub(ib.ax0, xs), 0) & 7u, last = (uint32_t)imin(isub(ib.ax1, xs), 7) & 7u;  // (in 0..7 for a live item)
                        const uint32_t inmask = act ? (2u << last) - (1u << first) : 0u;
                        // the polygon's constants, orientation-normalised as above
                        float a0 = __uint_as_float(q1.z), b0 = __uint_as_float(q1.w);
                        float a1 = __uint_as_float(q2.x), b1 = __uint_as_float(q2.y);
                        float cz = a0 * b1 - a1 * b0, yv = ryv;
                        if (cz < 0.0f) {
                            a0 = -a0; a1 = -a1; b0 = -b0; b1 = -b1;
                            cz = -cz;
                            yv = -yv;
                        }
                        const int32_t x0 = (int32_t)q1.x;
                        const float bf = (float)isub((int32_t)q1.y, py);
                        const V2 a1b = splat2v<V2>(a1 * bf), a0b = splat2v<V2>(a0 * bf);  // e.a1 * b2, e.a0 * b2 of edge_cross2
                        Edge2T<V2> e;
                        e.cz = splat2v<V2>(cz);
                        e.y = splat2v<V2>(yv);
                        const V2 z0 = splat2v<V2>(__uint_as_float(q2.z)), z1 = splat2v<V2>(__uint_as_float(q2.w));
                        const V2 z2 = splat2v<V2>(__uint_as_float(q3.x));
                        const uint32_t tag = (((SHARED_MAX_POLYGONS - 1u) - q3.y) << 12) | (c0 + rec + 1u);
                        const uint32_t yrel = (uint32_t)isub(py, qy0) & (uint32_t)(TILE_H - 1);
                        const uint32_t xrel = (uint32_t)isub(xs, qx0) & (uint32_t)(TILE_W - 2);
                        const uint32_t key_row = yrel * (uint32_t)TILE_W, key_sw = (yrel & 7u) << 3;
    #pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t kx = (xrel + 2u * k) & (uint32_t)(TILE_W - 1);  // (a chunk may start up to 6 pixels from the tile's end: wraps, masked out)
                            uint4 *const slot = reinterpret_cast<uint4 *>(wkey + (key_row + (kx ^ key_sw)));
                            const uint4 cur = *slot;
                            const int32_t px = xs + 2 * k;
                            const V2 a2 = mk2v<V2>((float)isub(x0, px), (float)isub(x0, px + 1));
                            const V2 cx = a1b - a2 * splat2v<V2>(b1), cy = a2 * splat2v<V2>(b0) - a0b;
                            const V2 rest = e.cz - (cx + cy);
                            const bool hita = ((inmask >> (2 * k)) & 1u) && __builtin_fminf(__builtin_fminf(cx.x, cy.x), rest.x) >= 0.0f;
                            const bool hitb = ((inmask >> (2 * k + 1)) & 1u) && __builtin_fminf(__builtin_fminf(cx.y, cy.y), rest.y) >= 0.0f;
                            if (hita || hitb) {
                                const Bary2T<V2> bar = barycentric2_for_compare(cx, cy, e);
                                const V2 z = dot3_2(bar.x, bar.y, bar.z, z0, z1, z2);
                                const unsigned long long ka = ((unsigned long long)depth_order_bits(z.x) << 32) | tag;
                                const unsigned long long kb = ((unsigned long long)depth_order_bits(z.y) << 32) | tag;
                                const unsigned long long ca_ = ((unsigned long long)cur.y << 32) | cur.x;
                                const unsigned long long cb_ = ((unsigned long long)cur.w << 32) | cur.z;
                                if (hita && ka > ca_)
                                    __hip_atomic_fetch_max(reinterpret_cast<unsigned long long *>(slot), ka, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (hitb && kb > cb_)
                                    __hip_atomic_fetch_max(reinterpret_cast<unsigned long long *>(slot) + 1, kb, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
            }
        }

    };
    if (shared_tile)
        resolve(std::bool_constant<SHARED>{});
    else
        resolve(std::false_type{});

    uint64_t t_covered = 0;
    if (stamps) t_covered = wall_clock64();

    // ---- shade the survivors and stream the tile out ------------------------------------------
    // Lanes are row-major here: lane = x within a 32-pixel row; a step covers four rows, each
    // lane carrying the two pixels (x, 4s + half) and (x, 4s + 2 + half) through packed
    // arithmetic.  Depth goes out straight from registers as whole 128-byte lines and colour is
    // packed to dwords with two lane permutes; the vertical flip of get_frame_buffer
    // (scene.rs:92-97) is folded into the colour address.  Lanes without a survivor run the same
    // loads on record 0 and discard the result, so the code is branch-free inside a step.
    // Addresses are a per-wave base (scalar registers) plus a 32-bit lane offset built with 24-bit
    // multiplies: 64-bit and 32 x 32 multiplies run at quarter rate and were a third of this
    // phase's vector-ALU time.
    if (SHARED || TILE_WAVES != TILE_W / STRIP) __syncthreads();  // a strip may hold pixels other waves resolved
    const int32_t hx = (int32_t)(lane & 31u), hrow = (int32_t)(lane >> 5);
    const uint32_t half_base = lane & 32u;
    const int32_t strip_x = (int32_t)(wave / (uint32_t)WAVES_PER_STRIP) * STRIP;  // within the tile
    const int32_t strip_y = (int32_t)(wave % (uint32_t)WAVES_PER_STRIP) * STRIP_ROWS;
    const int32_t sx0 = tile_x0 + strip_x, sy0 = qy0 + strip_y;
    const int32_t px = sx0 + hx;
    constexpr int NSTEP = STRIP_ROWS / 4;
    // strip origins: depth/winner at row sy0, colour at the strip's last row (lowest address of
    // the flipped image), so every lane offset is non-negative.  Rows outside the frame or the
    // band give addresses that are formed but never used.
    float *const depth_strip = depth + ((int64_t)sy0 * W + sx0);
    uint32_t *const winner_strip = (!GROUP && a.winner) ? a.winner + ((int64_t)sy0 * W + sx0) : nullptr;
    uint8_t *const fb_strip = DEPTH ? nullptr : fb + ((int64_t)(H - sy0 - STRIP_ROWS) * W + sx0) * 3;
    const uint32_t Wu = (uint32_t)W, W3 = 3u * (uint32_t)W;
    const bool col_live = px < W;

    // Reads a pixel's key and returns the bin slot + 1 of its survivor (0: none)
    // (zbits: the f32 bits of the depth the resolve compared for that pixel -- what a depth pass stores, below)
    auto survivor_slot = [&](uint32_t sx, uint32_t sy, uint32_t &zbits) -> uint32_t {
        if (SHARED && shared_tile) {
            // tie-break word: low 12 bits = bin slot + 1 of a fragment, 0xFFF / 0 = the buffer's old content
            const uint2 key = s_key[shared_key_slot((uint32_t)strip_x + sx, (uint32_t)strip_y + sy)];
            const uint32_t f = key.x & 0xFFFu;
            zbits = (key.y & 0x80000000u) ? key.y ^ 0x80000000u : ~key.y;  // (depth_order_bits undone; both zeros read +0)
            return (f == 0xFFFu) ? 0u : f;
        }
        // (the whole 8-byte key: 64 lanes x 8 bytes in a row are conflict-free as ds_read_b64, the upper dwords alone as
        // ds_read_b32 are a two-way bank conflict)
#if TR_KEY_B64
        const uint2 key = s_key[key_slot<QUAD_COLUMN>((uint32_t)strip_x + sx, (uint32_t)strip_y + sy)];
        uint32_t keep = key.x;
        asm volatile("" : "+v"(keep));   // (keeps the compiler from narrowing the read again)
        zbits = keep;
        return key.y;
#else
        zbits = 0u;
        return s_key[key_slot<QUAD_COLUMN>((uint32_t)strip_x + sx, (uint32_t)strip_y + sy)].y;
#endif
    };
    // Two pixels of a lane through the fragment stage, each against its own polygon: pixel u at (pxs[u], pys[u]),
    // survivor in bin slot wslot[u] if won[u].  Lanes without a survivor run the same loads on record 0 and
    // discard the result, so the code is branch-free.  RESIDENT: the bin's records are in LDS (nearly always);
    // PAIR: the two-pixel closures (tr_shaders.h) -- returns true when a surviving pixel left their guarded
    // range anywhere in the wave (the caller then runs the step again with the plain closures; keeping those
    // out of the fast loop's body matters: inline, as the fallback of each step, their registers were live
    // across the fast path and cost 5-9 % although they almost never ran).
    auto shade_two = [&](auto in_lds, auto pair_tag, const int32_t (&pxs)[2], const int32_t (&pys)[2], const bool (&won)[2],
                         const uint32_t (&wslot)[2], float (&zout)[2], uint32_t (&rgb)[2], uint32_t (&tri)[2]) -> bool {
        constexpr bool RESIDENT = decltype(in_lds)::value;
        constexpr bool PAIR = decltype(pair_tag)::value;
        bool redo = false;
        // The two survivors' records.  Pieces 1..TOP-1 (raster part, uv and, for the 6-piece record,
        // everything else) are taken for both pixels at once; the darboux record's further 16
        // varyings per pixel are fetched when that pixel's closure runs, one pixel after the other,
        // so that 32 fewer registers are live (123 -> under 96: a fifth wave per SIMD).
        constexpr int TOP = P < 6 ? P : 6;
        uint4 qa[TOP], qb[TOP];
        const uint4 *const ra = RESIDENT ? s_rec + mul24(wslot[0], (uint32_t)P) : bin + (size_t)wslot[0] * P;
        const uint4 *const rb = RESIDENT ? s_rec + mul24(wslot[1], (uint32_t)P) : bin + (size_t)wslot[1] * P;
#pragma unroll
        for (int i = 1; i < TOP; i++) {
            qa[i] = RESIDENT ? ra[i] : gload(ra + i);
            qb[i] = RESIDENT ? rb[i] : gload(rb + i);
        }
        const uint32_t rya = P > TOP ? (RESIDENT ? ra[P - 1].w : gload(ra + (P - 1)).w) : qa[TOP - 1].w;
        const uint32_t ryb = P > TOP ? (RESIDENT ? rb[P - 1].w : gload(rb + (P - 1)).w) : qb[TOP - 1].w;
        // to_barycentric_coord for both pixels (each against its own polygon)
        Edge2T<V2> e;
        e.a0 = mk2v<V2>(__uint_as_float(qa[1].z), __uint_as_float(qb[1].z));
        e.a1 = mk2v<V2>(__uint_as_float(qa[2].x), __uint_as_float(qb[2].x));
        e.b0 = mk2v<V2>(__uint_as_float(qa[1].w), __uint_as_float(qb[1].w));
        e.b1 = mk2v<V2>(__uint_as_float(qa[2].y), __uint_as_float(qb[2].y));
        e.cz = e.a0 * e.b1 - e.a1 * e.b0;
        e.y = mk2v<V2>(__uint_as_float(rya), __uint_as_float(ryb));
        const V2 a2 = mk2v<V2>((float)isub((int32_t)qa[1].x, pxs[0]), (float)isub((int32_t)qb[1].x, pxs[1]));
        const V2 b2 = mk2v<V2>((float)isub((int32_t)qa[1].y, pys[0]), (float)isub((int32_t)qb[1].y, pys[1]));
        V2 cx, cy;
        edge_cross2(e, a2, b2, cx, cy);
        const Bary2T<V2> bar = barycentric2(cx, cy, e);
        const V2 z = dot3_2(bar.x, bar.y, bar.z, mk2v<V2>(__uint_as_float(qa[2].z), __uint_as_float(qb[2].z)),
                            mk2v<V2>(__uint_as_float(qa[2].w), __uint_as_float(qb[2].w)),
                            mk2v<V2>(__uint_as_float(qa[3].x), __uint_as_float(qb[3].x)));
        uint32_t ca = 0u, cb = 0u, ea = 0u, eb = 0u;
        if (!DEPTH && st_c) {   // (a depth-only repeat of a colour pass: the z, no closure)
            // uv = vertex_uvs * bar (2x3 gemv), both pixels
            V2 uu = mk2v<V2>(__uint_as_float(qa[3].z), __uint_as_float(qb[3].z)) * bar.x;
            V2 vv = mk2v<V2>(__uint_as_float(qa[3].w), __uint_as_float(qb[3].w)) * bar.x;
            uu = mk2v<V2>(__uint_as_float(qa[4].x), __uint_as_float(qb[4].x)) * bar.y + uu;
            vv = mk2v<V2>(__uint_as_float(qa[4].y), __uint_as_float(qb[4].y)) * bar.y + vv;
            uu = mk2v<V2>(__uint_as_float(qa[4].z), __uint_as_float(qb[4].z)) * bar.z + uu;
            vv = mk2v<V2>(__uint_as_float(qa[4].w), __uint_as_float(qb[4].w)) * bar.z + vv;
            if (FS == FS_LIT) {
                // the frame's lit texel image (k_lit): the closure has run for this texel already
                uint32_t unused1, unused2;
                vec3 unused3;
                fetch_texels<FS>(tex, uu.x, vv.x, ea, ca, unused1, unused2, unused3);
                fetch_texels<FS>(tex, uu.y, vv.y, eb, cb, unused1, unused2, unused3);
            } else if (FS == FS_DEFAULT || FS == FS_PHONG) {
                // shader.rs:318-333 / 386-401 for both pixels at once: texel, diffuse term,
                // color_blend(c, 0, t) = (t * c + (1 - t) * 0.0) as u8 per channel
                uint32_t ta, tb, unused1, unused2;
                vec3 unused3;
                fetch_texels<FS>(tex, uu.x, vv.x, ea, ta, unused1, unused2, unused3);
                fetch_texels<FS>(tex, uu.y, vv.y, eb, tb, unused1, unused2, unused3);
                V2 t = mk2v<V2>(__uint_as_float(qa[5].x), __uint_as_float(qb[5].x));
                if (FS == FS_PHONG)
                    t = dot3_2(bar.x, bar.y, bar.z, t, mk2v<V2>(__uint_as_float(qa[5].y), __uint_as_float(qb[5].y)),
                               mk2v<V2>(__uint_as_float(qa[5].z), __uint_as_float(qb[5].z)));
#if TR_BLEND_FAST
                t = mk2v<V2>(blend_weight(t.x), blend_weight(t.y));  // (blend_black, tr_math.h: the (1 - t) * 0.0 term as a select)
#else
                const V2 k = (splat2v<V2>(1.0f) - t) * splat2v<V2>(0.0f);
#endif
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
#if TR_BLEND_FAST
                    const V2 v = t * mk2v<V2>((float)((ta >> (8 * ch)) & 0xFFu), (float)((tb >> (8 * ch)) & 0xFFu));
#else
                    const V2 v = t * mk2v<V2>((float)((ta >> (8 * ch)) & 0xFFu), (float)((tb >> (8 * ch)) & 0xFFu)) + k;
#endif
                    ca = pack_u8(v.x, (uint32_t)ch, ca);   // (`as u8` and the byte's place in one step, tr_math.h)
                    cb = pack_u8(v.y, (uint32_t)ch, cb);
                }
            } else {
                // one pixel's closure after the other's (interleaved they need twice the registers)
                auto closure = [&](const uint4 (&q)[TOP], const uint4 *rec, vec3 b, float u_, float v_, int32_t px_, int32_t py_,
                                   float z_, uint32_t &e_) {
                    float v[VARY_STRIDE];
                    v[0] = __uint_as_float(q[3].z); v[1] = __uint_as_float(q[3].w);
#pragma unroll
                    for (int i = 4; i < P; i++) {
                        const uint4 piece = i < TOP ? q[i] : (RESIDENT ? rec[i] : gload(rec + i));
                        v[4 * i - 14] = __uint_as_float(piece.x); v[4 * i - 13] = __uint_as_float(piece.y);
                        v[4 * i - 12] = __uint_as_float(piece.z); v[4 * i - 11] = __uint_as_float(piece.w);
                    }
                    return fragment_color<FS>(a.u, tex, v, b, u_, v_, (uint32_t)px_, (uint32_t)py_, z_, a.shadow,
                                              (uint32_t)W, (uint32_t)H, e_, a.sclean);
                };
                if constexpr (PAIR) {
                    // both pixels through the closure together in packed arithmetic with shared
                    // reciprocals (tr_shaders.h, fragment_color_pair); a step in which a surviving pixel's
                    // operands leave the range that form is proven on is run again with the plain
                    // closure (rare: exact zeros among the normalised components, a degenerate basis)
                    auto vary2 = [&](int k) -> f2 {
                        const int i = k < 2 ? 3 : (k + 14) / 4, c = k < 2 ? k + 2 : (k + 14) % 4;
                        const uint4 pa = i < TOP ? qa[i] : (RESIDENT ? ra[i] : gload(ra + i)), pb = i < TOP ? qb[i] : (RESIDENT ? rb[i] : gload(rb + i));
                        const uint32_t wa = c == 0 ? pa.x : c == 1 ? pa.y : c == 2 ? pa.z : pa.w;
                        const uint32_t wb = c == 0 ? pb.x : c == 1 ? pb.y : c == 2 ? pb.z : pb.w;
                        return mk2(__uint_as_float(wa), __uint_as_float(wb));
                    };
                    bool bad_a, bad_b;
                    vec3p barp;
                    // (the two-pixel closures are written for the packed type)
                    barp.x = to_f2(bar.x); barp.y = to_f2(bar.y); barp.z = to_f2(bar.z);
                    fragment_color_pair<FS>(a.u, tex, vary2, barp, to_f2(uu), to_f2(vv), ca, cb, ea, eb, bad_a, bad_b);
                    if (__any((bad_a && won[0]) || (bad_b && won[1]))) {
                        redo = true;
                        ea = eb = 0u;  // the second run reports this step's lookups
                    }
                } else {
                    ca = closure(qa, ra, make3(bar.x.x, bar.y.x, bar.z.x), uu.x, vv.x, pxs[0], pys[0], z.x, ea);
                    __builtin_amdgcn_sched_barrier(0);
                    cb = closure(qb, rb, make3(bar.x.y, bar.y.y, bar.z.y), uu.y, vv.y, pxs[1], pys[1], z.y, eb);
                }
            }
        }
        uint32_t err = 0u;
        if (won[0]) {
            zout[0] = z.x;
            rgb[0] = ca;
            tri[0] = qa[3].y;
            err |= ea;
        }
        if (won[1]) {
            zout[1] = z.y;
            rgb[1] = cb;
            tri[1] = qb[3].y;
            err |= eb;
        }
        if (err) {
            atomicOr(a.err, err);
            *a.alarm = 1u;
        }
        return redo;
    };

    // One pixel of a lane to memory: row `row` of the strip (this lane's column hx), `live` = inside frame and band,
    // `put` = its depth (and winner word) changes.  Depth goes out straight from registers as whole 128-byte lines;
    // colour is packed to dwords with two lane permutes and stored flipped (scene.rs:92-97 folded in).
    auto store_pixel = [&](int32_t row, int32_t py_, bool live, bool won, float zv, uint32_t rgbv, uint32_t triv, bool with_winner) {
        const uint32_t zoff = mul24((uint32_t)row, Wu) + (uint32_t)hx;
        const uint32_t coff = mul24((uint32_t)(STRIP_ROWS - 1 - row), W3);  // the row's first byte
        if (!DEPTH && !fresh && live && !won && st_c) {
            // untouched pixel of an accumulate render: its colour may share a dword with a
            // touched neighbour, so fetch it
            const uint8_t *old = fb_strip + (coff + 3u * (uint32_t)hx);
            rgbv = pack_rgb(gload(old), gload(old + 1), gload(old + 2));
        }
        // depth: only pixels that changed (or every live pixel of a fresh tile)
        const bool put = live && (won || zfresh);
        // (TileArgs::store: a cleared frame's colour pass may leave its depth on the chip -- nothing reads the z buffer
        // of such a frame unless a getter or an accumulating render asks, and then the pass is repeated for the depth alone)
        if (put && st_z) gstore(depth_strip + zoff, zv);
        if (!DEPTH && st_c) {
            if (winner_strip && put && with_winner) gstore(winner_strip + zoff, triv);
            if (aligned4) {
                // dword j of the 96-byte row = bytes 4j..4j+3 = pixel p0 = 4j/3 from byte (4j)%3
                // on, topped up from pixel p0+1
                const uint32_t j = (uint32_t)hx;
                const uint32_t p0 = (4u * j) / 3u, o = (4u * j) % 3u;
                const uint32_t c0 = (uint32_t)__shfl((int)rgbv, (int)(half_base + (p0 & 31u)), 64);
                const uint32_t c1 = (uint32_t)__shfl((int)rgbv, (int)(half_base + ((p0 + 1u) & 31u)), 64);
                const uint32_t dw = (c0 >> (8u * o)) | (c1 << (24u - 8u * o));
                const bool row_live = py_ >= band_y0 && py_ < band_y1;
                if (j < 24u && row_live && (sx0 * 3 + (int32_t)(4u * j)) < W * 3)
                    gstore(reinterpret_cast<uint32_t *>(fb_strip + (coff + 4u * j)), dw);
            } else if (put) {
                uint8_t *p = fb_strip + (coff + 3u * (uint32_t)hx);
                gstore(p, (uint8_t)(rgbv & 0xFFu));
                gstore(p + 1, (uint8_t)((rgbv >> 8) & 0xFFu));
                gstore(p + 2, (uint8_t)((rgbv >> 16) & 0xFFu));
            }
        }
    };

    // ---- row-major form: lane = x within a 32-pixel row; a step covers four rows, each lane carrying the
    // two pixels (x, 4s + half) and (x, 4s + 2 + half) through the fragment stage and straight to memory.
    // Exists for bins that stay resident in LDS and for the rare larger ones, whose survivors' records come
    // from global memory -- a run-time choice inside the loop cost a dozen register moves per step where the
    // two paths merge.  Returns the steps to run again with the plain closures (every store of a step is
    // repeated, so the second run simply overwrites the first).
    auto shade_steps = [&](auto in_lds, auto pair_tag, uint32_t step_mask) -> uint32_t {
    uint32_t redo = 0u;
#pragma unroll 1
    for (int32_t sstep = 0; sstep < NSTEP; sstep++) {
        if (!((step_mask >> sstep) & 1u)) continue;
        int32_t row[2], py[2];
        bool live[2], won[2];
        uint32_t wslot[2], zkey[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            row[u] = sstep * 4 + u * 2 + hrow;  // within the strip
            py[u] = sy0 + row[u];
            live[u] = col_live && py[u] >= band_y0 && py[u] < band_y1;
            const uint32_t s1 = survivor_slot((uint32_t)hx, (uint32_t)row[u], zkey[u]);
            won[u] = live[u] && s1 != 0u;
            wslot[u] = won[u] ? s1 - 1u : 0u;
        }
        uint32_t tri[2] = { NO_WINNER, NO_WINNER }, rgb[2] = { 0u, 0u };
        float zout[2] = { bits_f32(TR_F32_MIN_BITS), bits_f32(TR_F32_MIN_BITS) };
        // A depth pass (shader.rs:694-709, 832-847) stores the survivor's z and nothing else -- and the resolve has
        // computed exactly that value: the same polygon, the same pixel, the same operations (barycentric2_for_compare
        // differs from barycentric2 only in the SIGN of a zero quotient, which reaches the sum only when the sum is itself
        // a zero).  So the key's depth IS the stored depth unless it is a zero; a step with a zero among its survivors'
        // depths takes the fragment stage as before.  No record gather, no cross products, no quotients for the rest:
        // the depth pass of the reference's model at 4096^2 22.5 -> see profiles/r04_notes.md.
        bool from_keys = false;
        if (DEPTH && TR_DEPTH_FROM_KEYS && TR_KEY_B64) {
            from_keys = !__any((won[0] && (zkey[0] << 1) == 0u) || (won[1] && (zkey[1] << 1) == 0u));
            if (from_keys) {
                if (won[0]) zout[0] = __uint_as_float(zkey[0]);
                if (won[1]) zout[1] = __uint_as_float(zkey[1]);
            }
        }
        if (!(TR_DBG_SKIP & 1) && !from_keys && __any(won[0] || won[1])) {
            const int32_t pxs[2] = { px, px };
            if (shade_two(in_lds, pair_tag, pxs, py, won, wslot, zout, rgb, tri)) redo |= 1u << sstep;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) store_pixel(row[u], py[u], live[u], won[u], zout[u], rgb[u], tri[u], true);
    }
    return redo;
    };

    constexpr uint32_t ALL_STEPS = (1u << NSTEP) - 1u;
    constexpr bool HAS_PAIR = has_pair_closure(FS);
    uint32_t redo;
    if (resident)
        redo = shade_steps(std::true_type{}, std::bool_constant<HAS_PAIR>{}, ALL_STEPS);
    else
        redo = shade_steps(std::false_type{}, std::bool_constant<HAS_PAIR>{}, ALL_STEPS);
    if (HAS_PAIR && redo != 0u) {
        if (resident)
            shade_steps(std::true_type{}, std::false_type{}, redo);
        else
            shade_steps(std::false_type{}, std::false_type{}, redo);
    }

    if (a.zclean && tid == 0u && st_z) a.zclean[tile] = 0u;
    if (!DEPTH && a.fbclean && tid == 0u && st_c) a.fbclean[tile] = 0u;

    if (stamps) {
        __syncthreads();
        if (tid == 0u) {
            stamps[8u * tile + 0u] = t_start;
            stamps[8u * tile + 1u] = wall_clock64();
            stamps[8u * tile + 2u] = n;
            stamps[8u * tile + 3u] = __smid();
            stamps[8u * tile + 4u] = t_staged;
            stamps[8u * tile + 5u] = t_covered;
        }
    }
}

// -----------------------------------------------------------------------------------------
// Small utility kernels
// -----------------------------------------------------------------------------------------

// Writes the f32::MIN of every colour-pass tile whose fast-clear flag is up and lowers the flag:
// run before anything reads the z buffer as plain memory (tr_scene_read_z_f32, get_z_buffer).
__global__ __launch_bounds__(256) void k_materialize_depth(float *zbuf, uint32_t *zclean, DevFrame frame)
{
    const uint32_t t = blockIdx.x;
    if (zclean[t] == 0u) return;
    const int32_t x0 = (int32_t)(t % frame.ntx) * TILE_W, y0 = (frame.ty_base + (int32_t)(t / frame.ntx)) * TILE_H;
    for (uint32_t p = threadIdx.x; p < (uint32_t)(TILE_W * TILE_H); p += 256u) {
        const int32_t y = y0 + (int32_t)(p / TILE_W), x = x0 + (int32_t)(p % TILE_W);
        if (x < (int32_t)frame.width && y >= frame.band_y0 && y < frame.band_y1)
            zbuf[(size_t)y * frame.width + x] = bits_f32(TR_F32_MIN_BITS);
    }
    __syncthreads();
    if (threadIdx.x == 0u) zclean[t] = 0u;
}

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

// Streaming a frame out to the host (app.rs:213-218 hands every frame to the window): one workgroup per tile of
// the scene's band writes the tile's rows straight into the page-locked HOST buffer -- or nothing at all: a tile
// whose colour-clean flag is up holds the cleared colour on the device, and if the host buffer's own flag says
// the tile was zeros the last time it was written there, nothing has to cross PCIe (three quarters of a
// 4096^2 frame of the reference's model).  `host_clean` belongs to (scene, host buffer); 16-byte pieces, so
// the launcher requires width % 16 == 0.
__global__ __launch_bounds__(256) void k_read_back(const uint8_t *__restrict__ fb, uint8_t *__restrict__ host,
                                                   const uint32_t *__restrict__ fb_clean, uint32_t *host_clean, DevFrame frame)
{
    const uint32_t t = blockIdx.x;
    const bool zeros = fb_clean[t] != 0u;
    const bool host_zeros = host_clean[t] != 0u;
    __syncthreads();  // (every thread has read the host flag before thread 0 changes it)
    if (zeros && host_zeros) return;
    const int32_t W = (int32_t)frame.width, H = (int32_t)frame.height;
    const int32_t x0 = (int32_t)(t % frame.ntx) * TILE_W, y0 = (frame.ty_base + (int32_t)(t / frame.ntx)) * TILE_H;
    // TILE_H rows x 24 pieces of 16 B
    for (uint32_t c = threadIdx.x; c < (uint32_t)TILE_H * 24u; c += 256u) {
        const int32_t y = y0 + (int32_t)(c / 24u);
        const int32_t xb = x0 * 3 + (int32_t)(c % 24u) * 16;
        if (xb < W * 3 && y >= frame.band_y0 && y < frame.band_y1) {
            const size_t at = (size_t)(H - 1 - y) * W * 3 + xb;
            const uint4 v = zeros ? make_uint4(0u, 0u, 0u, 0u) : *reinterpret_cast<const uint4 *>(fb + at);
            *reinterpret_cast<uint4 *>(host + at) = v;
        }
    }
    if (threadIdx.x == 0u) host_clean[t] = zeros ? 1u : 0u;
}

// The sparse form of the frame exchange between GPUs (tr_exchange_all_gather_tiles): this rank's band goes to a
// peer's copy of the frame tile by tile, like k_read_back to the host -- a tile whose colour-clean flag is up holds
// zeros here, and if this rank's record of the PEER's copy (`remote_clean`, kept by the exchange per slot and peer)
// says the tile was zeros the last time it was written there, nothing crosses the link: three quarters of a 4096^2
// frame of the reference's model.  `peer` is the peer's frame slot mapped into this process (HIP IPC); the stores
// are plain stores over xGMI, made visible by the kernel's end and the system-scope flag store that follows it on
// the stream.  `poisoned`: the exchange's error word -- after a peer failed to open its slot in time nothing is
// written into it.  `bytes`: running count of what was pushed.
__global__ __launch_bounds__(256) void k_push_tiles(const uint8_t *__restrict__ fb, uint8_t *__restrict__ peer,
                                                    const uint32_t *__restrict__ fb_clean, uint32_t *remote_clean, DevFrame frame,
                                                    const uint32_t *poisoned, unsigned long long *bytes)
{
    // one load of the error word per workgroup, then a uniform branch: a time-out raised on another copy stream
    // while this kernel runs must not split the workgroup around the barrier
    __shared__ uint32_t s_poisoned;
    if (threadIdx.x == 0u) s_poisoned = __hip_atomic_load(poisoned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t t = blockIdx.x;
    const bool zeros = fb_clean[t] != 0u;
    const bool remote_zeros = remote_clean[t] != 0u;
    __syncthreads();  // (every thread has read the remote flag before thread 0 changes it)
    if (s_poisoned != 0u) return;
    if (zeros && remote_zeros) return;
    const int32_t W = (int32_t)frame.width, H = (int32_t)frame.height;
    const int32_t x0 = (int32_t)(t % frame.ntx) * TILE_W, y0 = (frame.ty_base + (int32_t)(t / frame.ntx)) * TILE_H;
    uint32_t pieces = 0;
    // TILE_H rows x 24 pieces of 16 B
    for (uint32_t c = threadIdx.x; c < (uint32_t)TILE_H * 24u; c += 256u) {
        const int32_t y = y0 + (int32_t)(c / 24u);
        const int32_t xb = x0 * 3 + (int32_t)(c % 24u) * 16;
        if (xb < W * 3 && y >= frame.band_y0 && y < frame.band_y1) {
            const size_t at = (size_t)(H - 1 - y) * W * 3 + xb;
            const uint4 v = zeros ? make_uint4(0u, 0u, 0u, 0u) : *reinterpret_cast<const uint4 *>(fb + at);
            *reinterpret_cast<uint4 *>(peer + at) = v;
            pieces++;
        }
    }
    if (pieces) atomicAdd(bytes, (unsigned long long)pieces * 16ull);
    if (threadIdx.x == 0u) remote_clean[t] = zeros ? 1u : 0u;
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

// tr_selftest_shadow_fetch: the DEVICE form of shadow_fetch (tr_shaders.h) on caller-chosen coordinates, once
// through the fast-clear flags on a buffer whose flagged tiles hold stale values, once as a plain lookup in the
// materialised buffer: value bits and error bits of both, for the host to compare.
__global__ __launch_bounds__(256) void k_selftest_shadow(const float *plain, const float *stale, const uint32_t *sclean,
                                                         uint32_t W, uint32_t H, const float *x, const float *y, uint32_t n,
                                                         uint32_t *out_plain, uint32_t *out_flagged, uint32_t *err_plain,
                                                         uint32_t *err_flagged)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t e0 = 0u, e1 = 0u;
    const float a = shadow_fetch(plain, nullptr, W, H, make3(x[i], y[i], 0.0f), e0);
    const float b = shadow_fetch(stale, sclean, W, H, make3(x[i], y[i], 0.0f), e1);
    out_plain[i] = __float_as_uint(a);
    out_flagged[i] = __float_as_uint(b);
    err_plain[i] = e0;
    err_flagged[i] = e1;
}

// tr_selftest_device_unary: rcp2 / sqrt2 (tr_pk.h) against the compiler's correctly rounded
// 1.0f / x and sqrtf for every f32 whose bits lie in [first, first + count).
__global__ __launch_bounds__(256) void k_selftest_unary(int which, uint32_t first, uint64_t count,
                                                        unsigned long long *n_bad, uint32_t *bad_bits)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float x = __uint_as_float(first + (uint32_t)i);
        float got, want;
        if (which == 2) {
            // pack_u8 (v_cvt_pk_u8_f32) against the cast it replaces, for x and -x, into every byte of a word
            const uint32_t into = 0xA5C3E17Bu, b = (uint32_t)(i & 3u);
            const uint32_t g0 = pack_u8(x, b, into), w0 = (into & ~(0xFFu << (8u * b))) | (f32_to_u8(x) << (8u * b));
            const uint32_t g1 = pack_u8(-x, b, into), w1 = (into & ~(0xFFu << (8u * b))) | (f32_to_u8(-x) << (8u * b));
            got = __uint_as_float(g0 ^ (g1 << 1) ^ (g1 >> 31));
            want = __uint_as_float(w0 ^ (w1 << 1) ^ (w1 >> 31));
            if (g0 != w0 || g1 != w1) {
                const unsigned long long k = atomicAdd(n_bad, 1ull);
                if (k < 16ull) bad_bits[k] = __float_as_uint(x);
            }
            continue;
        }
        if (which == 0) {
            got = rcp2(mk2(x, -x)).x;
            want = 1.0f / x;
            if (__float_as_uint(rcp2(mk2(x, -x)).y) != __float_as_uint(-1.0f / x)) got = __uint_as_float(~__float_as_uint(want));
        } else {
            got = sqrt2(mk2(x, x)).y;
            want = sqrtf(x);
        }
        if (__float_as_uint(got) != __float_as_uint(want)) {
            const unsigned long long k = atomicAdd(n_bad, 1ull);
            if (k < 16ull) bad_bits[k] = __float_as_uint(x);
        }
    }
}

// -----------------------------------------------------------------------------------------
// Peer exchange flags (tr_exchange.cpp): generation counters in uncached memory, possibly another
// GPU's.  Stores and loads are system scope; a waiter sleeps between polls and gives up after
// `timeout_ticks` of the 100 MHz wall clock (ten seconds unless the host says otherwise), raising the
// error word instead of hanging the queue.
// -----------------------------------------------------------------------------------------
// (`unless`: the exchange's error word -- after a wait on this queue has timed out, what was to be announced may
// not have happened: the peer is not told)
__global__ __launch_bounds__(64) void k_flag_store(uint32_t *flag, uint32_t value, const uint32_t *unless)
{
    if (threadIdx.x == 0u) {
        if (unless && __hip_atomic_load(unless, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// one launch tells every peer: lane p stores into peer p's block
__global__ __launch_bounds__(64) void k_flags_store_all(uint32_t *const *flags, uint32_t n, uint32_t skip, uint32_t value)
{
    if (threadIdx.x < n && threadIdx.x != skip) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(flags[threadIdx.x], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ __forceinline__ void spin_until_at_least(uint32_t *flag, uint32_t value, uint32_t *error, uint64_t timeout_ticks)
{
    const uint64_t t0 = wall_clock64();
    // generations are compared as a signed distance so that the counter may wrap
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

__global__ __launch_bounds__(64) void k_flag_wait(uint32_t *flag, uint32_t value, uint32_t *error, uint64_t timeout_ticks)
{
    if (threadIdx.x == 0u) spin_until_at_least(flag, value, error, timeout_ticks);
}

__global__ __launch_bounds__(64) void k_flags_wait_all(uint32_t *const *flags, uint32_t n, uint32_t skip, uint32_t value,
                                                       uint32_t *error, uint64_t timeout_ticks)
{
    if (threadIdx.x < n && threadIdx.x != skip) spin_until_at_least(flags[threadIdx.x], value, error, timeout_ticks);
}

}  // namespace

int launch_flag_store_unless(uint32_t *flag, uint32_t value, const uint32_t *unless, hipStream_t st)
{
    hipLaunchKernelGGL(k_flag_store, dim3(1), dim3(64), 0, st, flag, value, unless);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

int launch_flag_store(uint32_t *flag, uint32_t value, hipStream_t st) { return launch_flag_store_unless(flag, value, nullptr, st); }

int launch_flags_store_all(uint32_t *const *flags, uint32_t n, uint32_t skip, uint32_t value, hipStream_t st)
{
    if (n > 64u) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_flags_store_all, dim3(1), dim3(64), 0, st, flags, n, skip, value);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

int launch_flag_wait(uint32_t *flag, uint32_t value, uint32_t *error, uint64_t timeout_ticks, hipStream_t st)
{
    hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, st, flag, value, error, timeout_ticks);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

int launch_flags_wait_all(uint32_t *const *flags, uint32_t n, uint32_t skip, uint32_t value, uint32_t *error,
                          uint64_t timeout_ticks, hipStream_t st)
{
    if (n > 64u) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_flags_wait_all, dim3(1), dim3(64), 0, st, flags, n, skip, value, error, timeout_ticks);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

int launch_selftest_unary(int which, uint32_t first, uint64_t count, unsigned long long *n_bad, uint32_t *bad_bits,
                          hipStream_t st)
{
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_selftest_unary, dim3(4096), dim3(256), 0, st, which, first, count, n_bad, bad_bits);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

// -----------------------------------------------------------------------------------------
// Launchers
// -----------------------------------------------------------------------------------------

#define TR_LAUNCH_CHECK()                    \
    do {                                     \
        hipError_t e_ = hipGetLastError();   \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

int rec_pieces_for_vs(int vs) { return vs == VS_DARBOUX ? REC_PIECES_LARGE : REC_PIECES_SMALL; }

// Polygons per wave of the chain's kernels: all 64 lanes beside a running tile kernel (throughput: the fewer waves, the
// less it is disturbed); when the caller's tile kernel WAITS for the chain (`hurry`, see launch_setup) and the mesh is
// small, 8 -- the wave with the most (polygon, tile) pairs is the critical path
static uint32_t chain_polys(uint32_t n_tri, uint32_t frames, bool hurry)
{
    uint32_t polys = BIN_POLYS;
    if (hurry && (uint64_t)((n_tri + polys - 1u) / polys) * frames < 2048u) polys = 8u;
    return polys;
}
int rec_pieces_for_fs(int fs) { return fs == FS_DARBOUX ? REC_PIECES_LARGE : REC_PIECES_SMALL; }

int launch_setup(int vs, const SetupArgs &a, const SetupArgs *group, uint32_t n_frames, bool hurry, hipStream_t st,
                 hipEvent_t start, hipEvent_t done)
{
    if ((int)a.rec_pieces != rec_pieces_for_vs(vs)) return (int)hipErrorInvalidValue;
    if (group && (n_frames == 0 || n_frames > 65535u)) return (int)hipErrorInvalidValue;
    // (an empty mesh: the first workgroup still zeroes the pass's list words)
    // (`hurry`: nothing else is on the GPU and the caller's tile kernel waits for this chain -- a lone frame, or the
    // first group after a synchronisation: single waves, spread over four times as many compute units.  A per-frame
    // launch always: its chain is as long as the tile kernel it has to hide behind -- the throughput shapes made the
    // unfused per-frame loop 65 us per frame at 4096^2 where these give 38)
    hurry = hurry || !group;
    // polygons per wave: as k_bin (launch_bin), so that a wave of k_bin finds the boxes a wave of k_setup counted
    const uint32_t polys = chain_polys(a.mesh.n_tri, group ? n_frames : 1u, hurry);
    const uint32_t per_group = hurry ? 1u : CHAIN_WAVES, waves = (a.mesh.n_tri + polys - 1u) / polys;
    const dim3 grid(a.mesh.n_tri ? (waves + per_group - 1u) / per_group : 1u, group ? n_frames : 1u), block(64u * per_group);
#define TR_SETUP_CASE(V)                                                                   \
    case V:                                                                                \
        if (group)                                                                         \
            hipExtLaunchKernelGGL(k_setup_group<V>, grid, block, 0, st, start, done, 0, group, polys); \
        else                                                                               \
            hipExtLaunchKernelGGL(k_setup<V>, grid, block, 0, st, start, done, 0, a, polys);      \
        break;
    switch (vs) {
    TR_SETUP_CASE(VS_DEFAULT)
    TR_SETUP_CASE(VS_PHONG)
    TR_SETUP_CASE(VS_PLAIN)
    TR_SETUP_CASE(VS_DARBOUX)
    TR_SETUP_CASE(VS_DEPTH)
    default: return (int)hipErrorInvalidValue;
    }
#undef TR_SETUP_CASE
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_bin(const SetupArgs &a, const SetupArgs *group, uint32_t n_frames, bool hurry, hipStream_t st, hipEvent_t start,
               hipEvent_t done)
{
    if (a.mesh.n_tri == 0) return 0;
    if (group && (n_frames == 0 || n_frames > 65535u)) return (int)hipErrorInvalidValue;
    // polygons per wave: all 64 lanes beside a running tile kernel (throughput: the fewer waves, the less it is
    // disturbed); when the caller's tile kernel WAITS for this chain (`hurry`, see launch_setup) and the mesh is
    // small, 8 -- the wave with the most (polygon, tile) pairs is the critical path (628 waves for 5 022 polygons:
    // 13 us instead of 39)
    hurry = hurry || !group;  // (see launch_setup)
    const uint32_t polys = chain_polys(a.mesh.n_tri, group ? n_frames : 1u, hurry);
    const uint32_t lds = 0u, waves = (a.mesh.n_tri + polys - 1u) / polys;
    const uint32_t per_group = hurry ? 1u : CHAIN_WAVES;
    const dim3 grid((waves + per_group - 1u) / per_group, group ? n_frames : 1u), block(64u * per_group);
    if (a.rec_pieces == (uint32_t)REC_PIECES_LARGE) {
        if (group)
            hipExtLaunchKernelGGL(k_bin_group<REC_PIECES_LARGE>, grid, block, lds, st, start, done, 0, group, polys);
        else
            hipExtLaunchKernelGGL(k_bin<REC_PIECES_LARGE>, grid, block, lds, st, start, done, 0, a, polys);
    } else if (a.rec_pieces == (uint32_t)REC_PIECES_SMALL) {
        if (group)
            hipExtLaunchKernelGGL(k_bin_group<REC_PIECES_SMALL>, grid, block, lds, st, start, done, 0, group, polys);
        else
            hipExtLaunchKernelGGL(k_bin<REC_PIECES_SMALL>, grid, block, lds, st, start, done, 0, a, polys);
    } else {
        return (int)hipErrorInvalidValue;
    }
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_lit(int fs, const SetupArgs &a, const SetupArgs *group, uint32_t n_frames, hipStream_t st, hipEvent_t start, hipEvent_t done)
{
    if (!a.lit || !a.texel_set || a.tex_w == 0 || a.tex_h == 0) return (int)hipErrorInvalidValue;
    if (group && (n_frames == 0 || n_frames > 65535u)) return (int)hipErrorInvalidValue;
    const uint32_t rows = (a.tex_h + 1u) / 2u;
    const uint64_t texels = (uint64_t)a.set_bpr * rows * 8u;
    const dim3 grid((uint32_t)((texels / 2u + 255u) / 256u), group ? n_frames : 1u), block(256);  // (two texels per thread)
    if (fs == FS_SPECULAR) {
        if (group) hipExtLaunchKernelGGL(k_lit_group<FS_SPECULAR>, grid, block, 0, st, start, done, 0, group);
        else hipExtLaunchKernelGGL(k_lit<FS_SPECULAR>, grid, block, 0, st, start, done, 0, a);
    } else if (fs == FS_NORMAL_MAP) {
        if (group) hipExtLaunchKernelGGL(k_lit_group<FS_NORMAL_MAP>, grid, block, 0, st, start, done, 0, group);
        else hipExtLaunchKernelGGL(k_lit<FS_NORMAL_MAP>, grid, block, 0, st, start, done, 0, a);
    } else {
        return (int)hipErrorInvalidValue;
    }
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_order(const TileArgs &one, uint32_t n_tiles, const TileArgs *group, uint32_t n_frames, hipStream_t st, hipEvent_t start,
                 hipEvent_t done)
{
    if (n_tiles == 0) return 0;
    if (group && (n_frames == 0 || n_frames > 65535u)) return (int)hipErrorInvalidValue;
    const dim3 grid((n_tiles + ORDER_THREADS - 1u) / ORDER_THREADS, group ? n_frames : 1u), block(ORDER_THREADS);
    uint32_t bits = 1;
    while ((1u << bits) < n_tiles) bits++;
    hipExtLaunchKernelGGL(k_order, grid, block, 0, st, start, done, 0, one, n_tiles, bits, group);
    TR_LAUNCH_CHECK();
    return 0;
}

template <int WAVES, bool SHARED>
static int launch_tile_waves(int fs, const TileArgs &a, uint32_t n_tiles, const TileArgs *group, uint32_t n_frames,
                             hipStream_t st, hipEvent_t start, hipEvent_t done, bool fused_single)
{
    // fused_single: ONE frame that starts from cleared targets and has no winner tap runs the fused launches' kernels
    // (compiled for exactly that: no accumulate paths, no tap, eight workgroups per CU, depth left on the chip when
    // a.store says so) with its arguments by value: no table (`group` null), one frame.
    const bool fused = group != nullptr || fused_single;
    if (fused_single && !group) n_frames = 1u;
    // (n_tiles here: the workgroups per frame -- the frame's tiles, or the work units a host that knows the lists'
    // lengths asks for, plan::work_units)
    const dim3 grid(n_tiles * (fused ? n_frames : 1u)), block(64 * WAVES);
    // (a fused launch whose frames leave their depth on the chip: `a` -- what the group's frames have in common -- says so)
    const bool transient = fused && fs != FS_DEPTH && a.store == TR_STORE_COLOR;
#define TR_TILE_CASE(F)                                                                                                  \
    case F:                                                                                                              \
        if (transient)                                                                                                   \
            hipExtLaunchKernelGGL((k_tile<F, WAVES, SHARED, (F == FS_DEPTH ? 1 : 2)>), grid, block, 0, st, start, done, 0, a, group, n_frames); \
        else if (fused)                                                                                                  \
            hipExtLaunchKernelGGL((k_tile<F, WAVES, SHARED, 1>), grid, block, 0, st, start, done, 0, a, group, n_frames); \
        else                                                                                                             \
            hipExtLaunchKernelGGL((k_tile<F, WAVES, SHARED, 0>), grid, block, 0, st, start, done, 0, a, nullptr, 0u); \
        break;
    switch (fs) {
    TR_TILE_CASE(FS_DEFAULT)
    TR_TILE_CASE(FS_PHONG)
    TR_TILE_CASE(FS_NORMAL_MAP)
    TR_TILE_CASE(FS_SPECULAR)
    TR_TILE_CASE(FS_DARBOUX)
    TR_TILE_CASE(FS_SHADOW2)
    TR_TILE_CASE(FS_OCCLUSION2)
    TR_TILE_CASE(FS_DEPTH)
    TR_TILE_CASE(FS_LIT)
    default: return (int)hipErrorInvalidValue;
    }
#undef TR_TILE_CASE
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_tile(int fs, const TileArgs &a, int tile_waves, int shared, uint32_t n_polygons, const TileArgs *group,
                uint32_t n_frames, hipStream_t st, hipEvent_t start, hipEvent_t done, uint32_t units_per_frame, bool fused_single)
{
    uint32_t n_tiles = a.frame.ntx * a.frame.nty;
    if (n_tiles == 0) return 0;
    // workgroups per frame: one per tile always suffices (units <= tiles); fewer when the caller knows the lists' lengths
    if (units_per_frame != 0u && units_per_frame < n_tiles) n_tiles = units_per_frame;
    if ((int)a.rec_pieces != rec_pieces_for_fs(fs)) return (int)hipErrorInvalidValue;
    if (group && (n_frames == 0 || (uint64_t)n_tiles * n_frames > 0x7FFFFFFFull)) return (int)hipErrorInvalidValue;
    if (fused_single && (group || a.fresh == 0u || a.winner)) return (int)hipErrorInvalidValue;  // (what those kernels are compiled for)
    // the shared keys pack polygon id and bin slot into 32 bits: beyond their fields, resolve by columns
    if (n_polygons > SHARED_MAX_POLYGONS) shared = 0;  // (a tile with more records than the slot field holds resolves by columns: k_tile)
    if (shared) {
        if (tile_waves == 16) return launch_tile_waves<16, true>(fs, a, n_tiles, group, n_frames, st, start, done, fused_single);
        if (tile_waves == 8) return launch_tile_waves<8, true>(fs, a, n_tiles, group, n_frames, st, start, done, fused_single);
        if (tile_waves == 4) return launch_tile_waves<4, true>(fs, a, n_tiles, group, n_frames, st, start, done, fused_single);
    } else {
        if (tile_waves == 16) return launch_tile_waves<16, false>(fs, a, n_tiles, group, n_frames, st, start, done, fused_single);
        if (tile_waves == 8) return launch_tile_waves<8, false>(fs, a, n_tiles, group, n_frames, st, start, done, fused_single);
        if (tile_waves == 4) return launch_tile_waves<4, false>(fs, a, n_tiles, group, n_frames, st, start, done, fused_single);
    }
    return (int)hipErrorInvalidValue;
}

int launch_materialize_depth(float *zbuf, uint32_t *zclean, const DevFrame &frame, hipStream_t st)
{
    const uint32_t n_tiles = frame.ntx * frame.nty;
    if (n_tiles == 0) return 0;
    hipLaunchKernelGGL(k_materialize_depth, dim3(n_tiles), dim3(256), 0, st, zbuf, zclean, frame);
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

int launch_read_back(const uint8_t *fb, uint8_t *host, const uint32_t *fb_clean, uint32_t *host_clean, const DevFrame &frame,
                     hipStream_t st)
{
    const uint32_t n_tiles = frame.ntx * frame.nty;
    if (n_tiles == 0) return 0;
    if (frame.width % 16u) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_read_back, dim3(n_tiles), dim3(256), 0, st, fb, host, fb_clean, host_clean, frame);
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_push_tiles(const uint8_t *fb, uint8_t *peer, const uint32_t *fb_clean, uint32_t *remote_clean, const DevFrame &frame,
                      const uint32_t *poisoned, unsigned long long *bytes, hipStream_t st)
{
    const uint32_t n_tiles = frame.ntx * frame.nty;
    if (n_tiles == 0) return 0;
    if (frame.width % 16u) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_push_tiles, dim3(n_tiles), dim3(256), 0, st, fb, peer, fb_clean, remote_clean, frame, poisoned, bytes);
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_selftest_shadow(const float *plain, const float *stale, const uint32_t *sclean, uint32_t W, uint32_t H,
                           const float *x, const float *y, uint32_t n, uint32_t *out_plain, uint32_t *out_flagged,
                           uint32_t *err_plain, uint32_t *err_flagged, hipStream_t st)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_selftest_shadow, dim3((n + 255u) / 256u), dim3(256), 0, st, plain, stale, sclean, W, H, x, y, n,
                       out_plain, out_flagged, err_plain, err_flagged);
    TR_LAUNCH_CHECK();
    return 0;
}

int specular_is_exact() { return TR_POWF_EXACT; }

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
