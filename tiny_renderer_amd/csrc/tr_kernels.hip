// tr_kernels.hip -- gfx950 kernels of the triangle-fill path.
//
// The reference (src/scene.rs:151-268) is one serial loop nest: pass x polygon x bbox-x x
// bbox-y, depth-testing and shading each covered pixel in polygon order.  Here a render pass is
//
//   k_setup   one thread per polygon: vertex closure, clamped bounding box, per-tile counts
//   k_scan    one workgroup: exclusive scan of the per-tile counts
//   k_fill    one thread per polygon: scatter polygon ids into per-tile bins
//   k_tile    one 256-thread workgroup per 128x32 screen tile: coverage + depth resolve in LDS,
//             then shading of the winners and a single streaming write of depth and colour
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
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.mesh.n_tri) return;

    RasterRec r;
    float vary[VARY_STRIDE];
#pragma unroll
    for (int i = 0; i < VARY_STRIDE; i++) vary[i] = 0.0f;
    uint32_t err = 0;
    const bool keep = vertex_stage<VS>(a.mesh, a.u, t, r, vary, err);
    if (keep)
        finish_raster_rec(r, a.frame);
    else
        mark_rejected(r);

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
        const int32_t tx0 = r.bx0 / TILE_W, tx1 = r.bx1 / TILE_W;
        const int32_t ty0 = r.by0 / TILE_H, ty1 = r.by1 / TILE_H;
        for (int32_t ty = ty0; ty <= ty1; ty++)
            for (int32_t tx = tx0; tx <= tx1; tx++) atomicAdd(&a.tile_count[tile_index(a.frame, tx, ty)], 1u);
    }
    if (err) atomicOr(a.err, err);
}

// -----------------------------------------------------------------------------------------
// k_scan: exclusive scan of tile_count -> tile_offset, one 1024-thread workgroup
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(ScanArgs a)
{
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (a.n_tiles + 1023u) / 1024u;
    const uint32_t begin = min(tid * per, a.n_tiles), end = min(begin + per, a.n_tiles);
    uint32_t sum = 0;
    for (uint32_t i = begin; i < end; i++) sum += a.tile_count[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t base = part[tid] - sum;
    for (uint32_t i = begin; i < end; i++) {
        a.tile_offset[i] = base;
        base += a.tile_count[i];
        a.tile_cursor[i] = 0u;
    }
    if (tid == 1023u) {
        a.tile_offset[a.n_tiles] = part[1023];
        if ((uint64_t)part[1023] > a.capacity) atomicOr(a.err, (uint32_t)DE_BIN_OVERFLOW);
    }
}

// -----------------------------------------------------------------------------------------
// k_fill: scatter polygon ids into the bins (order inside a bin is irrelevant, see header)
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill(FillArgs a)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_tri) return;
    const int4 box = *reinterpret_cast<const int4 *>(&a.rast[t].bx0);  // bx0,bx1,by0,by1
    if (box.x > box.y) return;
    const int32_t tx0 = box.x / TILE_W, tx1 = box.y / TILE_W;
    const int32_t ty0 = box.z / TILE_H, ty1 = box.w / TILE_H;
    for (int32_t ty = ty0; ty <= ty1; ty++)
        for (int32_t tx = tx0; tx <= tx1; tx++) {
            const int32_t tile = tile_index(a.frame, tx, ty);
            const uint64_t pos = (uint64_t)a.tile_offset[tile] + atomicAdd(&a.tile_cursor[tile], 1u);
            if (pos < a.capacity) a.bins[pos] = t;
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

    const uint32_t tile = blockIdx.x;
    const int32_t tx = (int32_t)(tile % a.frame.ntx);
    const int32_t ty = a.frame.ty_base + (int32_t)(tile / a.frame.ntx);
    const int32_t tile_x0 = tx * TILE_W, tile_y0 = ty * TILE_H;
    const int32_t W = (int32_t)a.frame.width, H = (int32_t)a.frame.height;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;

    uint32_t n = a.tile_count[tile];
    const uint32_t start = a.tile_offset[tile];
    if ((uint64_t)start + n > a.bin_capacity) n = (uint64_t)start < a.bin_capacity ? (uint32_t)(a.bin_capacity - start) : 0u;

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

    // ---- coverage + depth resolve: each wave walks the tile's bin over its own quadrant ---
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t tri = __builtin_amdgcn_readfirstlane(a.bins[start + k]);
        const RasterRec r = a.rast[tri];
        const int32_t bx0 = imax(r.bx0, qx0), bx1 = imin(r.bx1, qx0 + QUAD - 1);
        const int32_t by0 = imax(r.by0, qy0), by1 = imin(r.by1, qy0 + TILE_H - 1);
        if (bx0 > bx1 || by0 > by1) continue;
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

    // ---- shade the survivors and stream the tile out, two 32-pixel rows per wave step ------
    // Lanes are row-major here (lane = x within the row), so depth is stored straight from
    // registers as whole 128-byte lines and colour is packed to dwords with two lane
    // permutes.  The vertical flip of get_frame_buffer (scene.rs:92-97) is folded into the
    // colour address.
    const int32_t hx = (int32_t)(lane & 31u), hrow = (int32_t)(lane >> 5);
    const uint32_t half_base = lane & 32u;
    for (int32_t rp = 0; rp < TILE_H / 2; rp++) {
        const int32_t qy = rp * 2 + hrow;
        const int32_t px = qx0 + hx, py = qy0 + qy;
        const bool live = px < W && py >= a.frame.band_y0 && py < a.frame.band_y1;
        const uint64_t key = wkey[key_slot((uint32_t)hx, (uint32_t)qy)];
        const uint32_t low = (uint32_t)key;
        const bool won = live && low != PRIOR;

        uint32_t rgb = 0u;
        float zout = bits_f32(TR_F32_MIN_BITS);
        uint32_t tri = NO_WINNER;
        uint32_t err = 0u;
        if (won) {
            tri = DEPTH ? low - 1u : 0xFFFFFFFEu - low;
            const RasterRec *rr = a.rast + tri;
            const int4 v0 = *reinterpret_cast<const int4 *>(&rr->x0);    // x0 y0 x1 y1
            const int4 v1 = *reinterpret_cast<const int4 *>(&rr->x2);    // x2 y2 z0 z1
            RasterRec r;
            r.x0 = v0.x; r.y0 = v0.y; r.x1 = v0.z; r.y1 = v0.w;
            r.x2 = v1.x; r.y2 = v1.y;
            r.z0 = __int_as_float(v1.z); r.z1 = __int_as_float(v1.w); r.z2 = rr->z2;
            const Edge e = edge_setup(r);
            float cx, cy;
            edge_cross(e, px, py, cx, cy);
            const vec3 bar = barycentric(cx, cy, e.cz);
            zout = dot3(bar, make3(r.z0, r.z1, r.z2));
            if (!DEPTH) {
                const float4 *vp = reinterpret_cast<const float4 *>(a.vary + (size_t)tri * VARY_STRIDE);
                float vary[VARY_STRIDE];
                constexpr int NV = (FS == FS_DARBOUX) ? 6 : 3;
#pragma unroll
                for (int i = 0; i < NV; i++) {
                    const float4 q = vp[i];
                    vary[4 * i] = q.x; vary[4 * i + 1] = q.y; vary[4 * i + 2] = q.z; vary[4 * i + 3] = q.w;
                }
                rgb = fragment_stage<FS>(a.u, a.tex, vary, bar, (uint32_t)px, (uint32_t)py, zout, a.shadow,
                                         (uint32_t)W, (uint32_t)H, err);
            }
        } else if (live && !a.fresh && !DEPTH) {
            // untouched pixel of an accumulate render: its colour may share a dword with a
            // touched neighbour, so fetch it
            const uint8_t *old = a.fb + ((size_t)(H - 1 - py) * W + px) * 3;
            rgb = pack_rgb(old[0], old[1], old[2]);
        }
        if (err) atomicOr(a.err, err);

        // depth: only pixels that changed (or every live pixel of a fresh tile)
        if (live && (won || a.fresh)) depth[(size_t)py * W + px] = zout;

        if (!DEPTH) {
            if (a.winner && live && (won || a.fresh)) a.winner[(size_t)py * W + px] = tri;
            uint8_t *row = a.fb + ((size_t)(H - 1 - py) * W + qx0) * 3;
            if (a.aligned4) {
                // dword j of the 96-byte row = bytes 4j..4j+3 = pixel p0 = 4j/3 from byte (4j)%3 on,
                // topped up from pixel p0+1
                const uint32_t j = (uint32_t)hx;
                const uint32_t p0 = (4u * j) / 3u, o = (4u * j) % 3u;
                const uint32_t c0 = (uint32_t)__shfl((int)rgb, (int)(half_base + (p0 & 31u)), 64);
                const uint32_t c1 = (uint32_t)__shfl((int)rgb, (int)(half_base + ((p0 + 1u) & 31u)), 64);
                const uint32_t dw = (c0 >> (8u * o)) | (c1 << (24u - 8u * o));
                const bool row_live = py >= a.frame.band_y0 && py < a.frame.band_y1;
                if (j < 24u && row_live && (qx0 * 3 + (int32_t)(4u * j)) < W * 3)
                    *reinterpret_cast<uint32_t *>(row + 4u * j) = dw;
            } else if (live && (won || a.fresh)) {
                uint8_t *p = row + 3 * hx;
                p[0] = (uint8_t)(rgb & 0xFFu);
                p[1] = (uint8_t)((rgb >> 8) & 0xFFu);
                p[2] = (uint8_t)((rgb >> 16) & 0xFFu);
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

int launch_scan(const ScanArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, a);
    TR_LAUNCH_CHECK();
    return 0;
}

int launch_fill(const FillArgs &a, hipStream_t st)
{
    if (a.n_tri == 0) return 0;
    hipLaunchKernelGGL(k_fill, dim3((a.n_tri + 255u) / 256u), dim3(256), 0, st, a);
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
