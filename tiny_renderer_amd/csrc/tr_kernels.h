// tr_kernels.h -- launchers of the HIP kernels (tr_kernels.hip).  Each returns 0 or a
// hipError_t value; none synchronizes, allocates or copies, so a caller may capture them.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>

#include "tr_types.h"

namespace tr {

// `start` / `done` (may be null) are attached to the kernel's own dispatch packet: `done` is
// signalled by the kernel's completion without a separate event packet, and the pair brackets
// exactly the kernel's execution when both are timing events.
//
// `group` (a table of n_frames argument structs in DEVICE memory; null = an ordinary launch) makes a launch a
// FUSED one over a group of frames (tr_scene_render_frames): the workgroups of frame f take entry f of the
// table; `a` (and `tile_count` / `order`) then only describe what the frames have in common -- mesh size,
// tile grid, record layout, bin capacity -- to the launcher.  launch_order reads each frame's counters and
// work lists from the TILE kernel's table (TileArgs::tile_count / order / list_len).
// Vertex stage: every polygon's record into a.recs, its tiles' counters bumped.
// `hurry`: nothing else is on the GPU and the caller's tile kernel waits for the chain (shapes for latency, not for
// running beside a tile kernel).
int launch_setup(int vs_kind, const SetupArgs &a, const SetupArgs *group, uint32_t n_frames, bool hurry, hipStream_t st,
                 hipEvent_t start, hipEvent_t done);
// The frame's lit texel image (k_lit: the normal-map / specular closure once per texel); a.lit, a.texel_set etc. say where.
int launch_lit(int fs, const SetupArgs &a, const SetupArgs *group, uint32_t n_frames, hipStream_t st, hipEvent_t start, hipEvent_t done);
// Builds the tile kernel's work lists from the counters k_setup filled (same stream, after it) and gives every tile
// its range of the pool.  `one`: the pass's arguments (a per-frame launch); `group`: the fused launch's table.
int launch_order(const TileArgs &one, uint32_t n_tiles, const TileArgs *group, uint32_t n_frames, hipStream_t st, hipEvent_t start,
                 hipEvent_t done);
// Copies the records into the tiles' ranges, with each (polygon, tile) pair's coverage masks (same stream, after
// launch_order); counts the counters back down to zero.
int launch_bin(const SetupArgs &a, const SetupArgs *group, uint32_t n_frames, bool hurry, hipStream_t st, hipEvent_t start,
               hipEvent_t done);
// tile_waves: 4, 8 or 16 wavefronts per tile workgroup (see tr_types.h); shared != 0: the waves share the
// tile's bin and resolve through atomic keys instead of each owning a column of the tile (k_tile's
// SHARED parameter; falls back to columns when n_polygons or a.bin_cap exceed the key's fields).
// units_per_frame: workgroups per frame when the caller knows the pass's list lengths (tr_plan.h, group_grid_units;
// 0 = one per tile, which always suffices).
int launch_tile(int fs_kind, const TileArgs &a, int tile_waves, int shared, uint32_t n_polygons, const TileArgs *group,
                uint32_t n_frames, hipStream_t st, hipEvent_t start, hipEvent_t done, uint32_t units_per_frame = 0,
                bool fused_single = false);
int launch_materialize_depth(float *zbuf, uint32_t *zclean, const DevFrame &frame, hipStream_t st);
// The band's tiles of frame buffer `fb` into the page-locked host buffer `host` (device address of it), skipping the
// tiles that are zeros on both sides (fb_clean: the target's colour-clean flags; host_clean: the host buffer's own)
int launch_read_back(const uint8_t *fb, uint8_t *host, const uint32_t *fb_clean, uint32_t *host_clean, const DevFrame &frame,
                     hipStream_t st);
// The band's tiles of `fb` into a peer GPU's copy of the frame, skipping the tiles that are zeros on both sides
// (k_push_tiles; remote_clean: this rank's record of the peer's copy; poisoned: the exchange's error word).
int launch_push_tiles(const uint8_t *fb, uint8_t *peer, const uint32_t *fb_clean, uint32_t *remote_clean, const DevFrame &frame,
                      const uint32_t *poisoned, unsigned long long *bytes, hipStream_t st);
int launch_selftest(const float *x, const float *d, uint32_t n, uint32_t *out_u32, int32_t *out_i32,
                    uint32_t *out_u8, float *out_div, float *out_div_ref, hipStream_t st);
// Peer exchange flags (tr_exchange.cpp): system-scope store of a generation number; waits that poll
// until flag >= value (as a wrapping distance) and raise *error after timeout_ticks of the 100 MHz wall clock
int launch_flag_store(uint32_t *flag, uint32_t value, hipStream_t st);
// ... unless *unless != 0 (the exchange's error word: a wait earlier on the queue timed out)
int launch_flag_store_unless(uint32_t *flag, uint32_t value, const uint32_t *unless, hipStream_t st);
int launch_flags_store_all(uint32_t *const *flags, uint32_t n, uint32_t skip, uint32_t value, hipStream_t st);
int launch_flag_wait(uint32_t *flag, uint32_t value, uint32_t *error, uint64_t timeout_ticks, hipStream_t st);
int launch_flags_wait_all(uint32_t *const *flags, uint32_t n, uint32_t skip, uint32_t value, uint32_t *error,
                          uint64_t timeout_ticks, hipStream_t st);
// rcp2 (which = 0) / sqrt2 (1) of tr_pk.h against '/' and sqrtf for the f32 bit patterns [first, first + count)
int launch_selftest_unary(int which, uint32_t first, uint64_t count, unsigned long long *n_bad, uint32_t *bad_bits,
                          hipStream_t st);
// shadow_fetch (tr_shaders.h) on the device: through the fast-clear flags on `stale`, and plain on `plain`
int launch_selftest_shadow(const float *plain, const float *stale, const uint32_t *sclean, uint32_t W, uint32_t H,
                           const float *x, const float *y, uint32_t n, uint32_t *out_plain, uint32_t *out_flagged,
                           uint32_t *err_plain, uint32_t *err_flagged, hipStream_t st);
// 1 when the specular closure's powf reproduces the host libm's bit for bit (tr_powf.h)
int specular_is_exact();
int launch_depth_view(const float *src, uint8_t *dst, uint32_t W, uint32_t H, hipStream_t st);

}  // namespace tr
