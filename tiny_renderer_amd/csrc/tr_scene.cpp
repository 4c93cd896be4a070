// tr_scene.cpp -- host side of the C ABI: scene state, pass sequencing, buffer ownership.
//
// Mirrors the role of the reference's `Scene` (src/scene.rs:25-269) and `ShaderPipeline`
// registry (src/scene/shader.rs:97-112) for a device-resident frame: the model, textures,
// z / shadow / frame buffers live in HBM for the life of the scene; a frame is a handful of
// kernel launches on one HIP stream and nothing is copied unless a getter is called.
//
// There is no CPU rendering path in this library.  If HIP or a gfx950 device is not usable,
// tr_scene_create fails with TR_E_HIP.
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "tiny_renderer.h"
#include "tr_error.h"
#include "tr_kernels.h"
#include "tr_math.h"
#include "tr_plan.h"
#include "tr_powf.h"
#include "tr_prepare.h"
#include "tr_shaders.h"
#include "tr_texels.h"
#include "tr_types.h"

namespace tr {

namespace {
thread_local std::string g_last_error;
}

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

}  // namespace tr

using namespace tr;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return tr::fail(TR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

namespace {

// Page-locked host buffers handed out by tr_host_alloc: host address -> {bytes, device address of the mapping}
struct HostAlloc {
    size_t bytes;
    void *device;
    uint64_t serial;  // (an address can come back from a later allocation: what a scene remembers is about THIS one)
    // Who wrote the buffer last, and how often anybody has: a scene's record of the buffer's zero tiles (HostFlags)
    // holds only while that scene was the last writer and nobody has written since -- another scene reading back
    // into the same buffer, or tr_scene_host_buffer_written, moves `gen` on and every record of the buffer lapses
    uint64_t writer;
    uint64_t gen;
};
std::mutex g_host_mutex;
uint64_t g_host_serial = 0;
uint64_t g_scene_serial = 0;  // tr_scene::id
std::map<void *, HostAlloc> g_host_allocs;

struct PassDesc {
    int prepare_kind;  // 0 default_prepare, 1 shadow_pass_prepare_1, 2 shadow_pass_prepare_2
    int vs, fs;
};

struct PipelineDesc {
    const char *name;
    int n_passes;
    PassDesc pass[2];
};

// shader.rs:100-109 and the pass lists of shader.rs:282-963
const PipelineDesc kPipelines[P_COUNT] = {
    { "default", 1, { { 0, VS_DEFAULT, FS_DEFAULT }, {} } },
    { "phong", 1, { { 0, VS_PHONG, FS_PHONG }, {} } },
    { "normal_map", 1, { { 0, VS_PLAIN, FS_NORMAL_MAP }, {} } },
    { "specular", 1, { { 0, VS_PLAIN, FS_SPECULAR }, {} } },
    { "darboux", 1, { { 0, VS_DARBOUX, FS_DARBOUX }, {} } },
    { "shadow", 2, { { 1, VS_DEPTH, FS_DEPTH }, { 2, VS_PHONG, FS_SHADOW2 } } },
    { "occlusion", 2, { { 1, VS_DEPTH, FS_DEPTH }, { 2, VS_PLAIN, FS_OCCLUSION2 } } },
};

const char *kKernelNames[] = { "k_setup", "k_tile", "k_tile_depth", "k_clear", "k_order", "k_bin", "k_lit" };
enum KernelId { K_SETUP = 0, K_TILE, K_TILE_DEPTH, K_CLEAR, K_ORDER, K_BIN, K_LIT, K_COUNT };

struct EventPair {
    hipEvent_t a, b;
    int kernel;
    uint32_t frames;  // frames the launch covered (fused launches of tr_scene_render_frames: more than one)
};

}  // namespace

// How many passes the setup stream may run ahead of the tile kernels: setup of pass p waits for the
// tile kernel of pass p - LOOKAHEAD only.  With 2 the setup chain (k_setup, k_order_count,
// k_order_place: ~35 us beside a busy machine) had to fit inside one tile kernel (~34 us) and was
// the critical path of the frame loop; 3 gave it two (44.1 -> 42.4 us per frame); 5 lets four passes
// wait on the host for their setups to finish, so that their tile kernels need no wait packet (below).
// Costs LOOKAHEAD sets of bins (0.2 GB each at 4096^2) -- cheap next to 288 GB.
using tr::plan::LOOKAHEAD;   // (tr_plan.h: the decisions of this file live there, testable without a GPU)
constexpr uint64_t LIT_PIXELS_PER_TEXEL = 16;  // frame pixels per image texel from which the lit path (k_lit) is taken
constexpr int SETS = LOOKAHEAD + 1;
// Handing tile kernels to the main stream.  The tile kernel of pass p must run after that pass's setup
// (other stream).  A cross-stream wait packet sitting between two tile kernels in the main stream's
// queue costs 5.5 us (two back-to-back tile kernels are 3.8 us apart, 9.3 with the wait in between), and
// none is needed if the setup has ALREADY completed when the tile kernel is enqueued.  So a pass whose
// setup is queued stays "pending" on the host until one of these:
//   * more than BATCH passes are pending (the steady state of a running loop): the host waits for the
//     oldest one's setup event itself -- it has completed or is about to, the setup stream runs ahead --
//     and enqueues its tile kernel with no wait packet at all.  That host-side wait is also the
//     frames-in-flight limit: render() cannot run more than BATCH + LOOKAHEAD passes ahead of the GPU;
//   * its setup event reads complete at a later render(): out it goes, no wait (start-up);
//   * the main stream has run dry (its newest tile kernel has completed): the oldest pending pass goes
//     out behind a wait packet on ITS setup event -- the first frame after a sync;
//   * anything needs the main stream or waits for the scene -- a getter, a sync, a clear that must be
//     materialised, a read-back, destruction: everything pending goes out behind ONE wait for the newest
//     setup (the setup stream is in order).
// BATCH <= LOOKAHEAD - 1: the setup of pass p waits for the tile kernel of pass p - LOOKAHEAD, which must
// have been submitted by then.  Nothing observable changes (submit_pending runs before every use of the
// stream).  Round 1 handed over in fixed batches of four behind one wait packet: 35.2 us/frame where this
// gives 34.2 (same box), and at small frames 25 -> 20.5 us (800^2), 24.8 -> 22.4 (2048^2).
using tr::plan::BATCH;  // passes that may be pending (setup queued, tile kernel not yet)
constexpr int RING = 16;  // events: pass p's are waited for until pass p + LOOKAHEAD is set up
constexpr size_t ORDER_LISTS = 8;  // k_order's work lists per pass, n_tiles entries each (tr_kernels.hip)

// Frame groups (tr_scene_render_frames).  A lone frame cannot keep the GPU full: at 4096^2 a third of the tile
// kernel runs on a machine that is draining (2 168 busy tiles on 1 536 workgroup slots), and smaller frames never
// fill it at all; kernels of different launches do not overlap on this stack (profiles/r02_notes.md).  So the
// frames of a group are rendered by ONE launch per kernel -- vertex stage + binning, work lists, tiles -- each
// frame into a frame slot of its own (z, colour, fast-clear flags, shadow buffer).  Same replicated frame,
// k_tile per frame (us, group of 1 / 2 / 4 / 8): 4096^2 phong 33.9 / 29.0 / 27.0 / 27.0, darboux 65.9 / 59.5 /
// 58.9 / 56.3, 2048^2 phong 21.3 / 14.2 / 12.6 / 12.0, 800^2 26.3 / 12.8 / 7.4 / 4.5, 512^2 22.5 / 12.2 / 6.6 / 3.6.
using tr::plan::GROUP_MAX;   // frames per fused launch, at most
using tr::plan::GROUP_SETS;  // groups in flight: group g's setup reuses the bins of group g - GROUP_SETS
constexpr size_t LEN_WORDS = 8;  // a pass's words in GroupSet::h_lens: its eight list lengths

struct tr_scene {
    uint64_t id = 0;  // unique per scene of the process (who wrote a tr_host_alloc buffer last)
    uint32_t width = 0, height = 0;
    int pipeline = 0;
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // Scene::new defaults, scene.rs:66-69
    float light[3] = { 0.0f, 0.0f, -1.0f };
    float from[3] = { 0.0f, 0.0f, 1.0f };
    float at[3] = { 0.0f, 0.0f, 0.0f };
    float up[3] = { 0.0f, 1.0f, 0.0f };

    DevMesh mesh = {};
    DevTextures tex = {};
    DevFrame frame = {};       // the rows this scene owns (colour passes)
    DevFrame frame_full = {};  // the whole frame: depth passes fill the entire shadow buffer, whose
                               // lookups are in light space and can land anywhere (shader.rs:774-778)
    uint32_t n_tiles = 0, n_tiles_full = 0;

    // device allocations
    float *d_tri = nullptr;
    uint32_t *d_texel[4] = { nullptr, nullptr, nullptr, nullptr };
    uint32_t *d_packed = nullptr;  // the colour closure's images as one interleaved, tiled array (tr_texels.h)
    // The frame's lit texel image (k_lit, tr_kernels.hip): the normal-map / specular closure once per texel and frame
    // instead of once per fragment, where a frame has many more pixels than the images have texels.
    bool lit = false;
    uint32_t lit_words = 0, lit_bpr = 0, set_bpr = 0;  // words of one lit image; blocks per row of it / of the texel set
    uint32_t *d_lit[LOOKAHEAD] = {};                   // per pass in flight (per-frame path); a group set has its own
    // Per-tile polygon counters (followed by k_order's 16 words).  Colour passes (the scene's band)
    // and depth passes (always the whole frame) have different tile grids, hence a state each.
    // SETS sets: the tile kernel of pass q zeroes set (q + SETS - 1) % SETS, which no pass before
    // q + SETS - 1 touches; meanwhile the setup kernels of passes q + 1 .. q + LOOKAHEAD - 1 (running
    // ahead on the setup stream) fill their own sets.
    struct BinState {
        uint32_t *count[SETS] = {};
        uint64_t seq = 0;           // passes of this kind issued so far
    } bin_color, bin_depth;
    // Record bins and work lists, LOOKAHEAD-buffered by global pass number: pass p's setup fills
    // bins[p % LOOKAHEAD] while the tile kernels of the passes before it may still be reading theirs.
    WorkItem *d_order[LOOKAHEAD] = {};  // the tile kernel's work list (k_order)
    Piece *d_bins[LOOKAHEAD] = {};      // the passes' pools: pool_cap records of rec_pieces x 16 B each
    Piece *d_recs[LOOKAHEAD] = {};      // every polygon's record, once per pass (k_setup -> k_bin)
    // Pass pipelining: k_setup and k_order_* of pass p run on `setup_stream`, ordered after the tile
    // kernel of pass p - LOOKAHEAD (which freed its bins and zeroed its counters) and before the tile
    // kernel of pass p on the main stream.  They need only frame constants, so they overlap the tile
    // kernels of earlier passes: consecutive frames are in flight together, like any renderer's.
    hipStream_t setup_stream = nullptr;
    // A second one for the per-frame path: the chains of consecutive passes alternate between the two, so that the chain
    // of pass p + 1 (three dependent kernels, as long as the tile kernel it has to hide behind) starts beside the chain
    // of pass p instead of behind it.  Passes share nothing but the targets the tile kernels write (main stream, in order).
    hipStream_t setup_stream2 = nullptr;
    bool two_setup_streams = true;
    hipEvent_t ev_setup[RING] = {};
    hipEvent_t ev_tile[RING] = {};
    uint64_t pass_seq = 0;
    struct PendingTile {
        int fs, tile_waves, shared, kernel_id;
        bool fused_single = false;  // the pass starts from cleared targets without a winner tap: the fused launches' kernels (launch_tile)
        uint64_t p_seq;
        TileArgs args;
    };
    std::vector<PendingTile> pending;  // passes whose setup is queued and whose tile kernel is not yet
    uint64_t tiles_submitted = 0;      // tile kernels handed to the main stream so far
    uint64_t last_submitted_seq = 0;   // pass number of the newest of them (its ev_tile tells whether the stream is idle)
    uint32_t tile_waves = 0;     // tr_options.tile_waves: 4, 8, 16 or 0 = by tile count
    uint32_t tile_mode = 0;      // tr_options.tile_mode: 1 columns, 2 shared bin, 0 = automatic
    uint32_t pool_cap = 0;       // records in a pass's pool (all its (polygon, tile) pairs); grown should a pass exceed it
    uint32_t rec_pieces = 0;
    uint32_t *d_bin_need = nullptr;
    // First pass (global pass number) whose bins overflowed since the last sync, written by k_setup
    // with an atomic minimum; ~0 = none.  Frames older than the last one cannot be rendered again,
    // so sync compares it with `observed_seq`: passes below that number have been handed to a
    // consumer the library cannot call back (an asynchronous read-back, or a caller's stream).
    unsigned long long *d_overflow_seq = nullptr;
    uint64_t observed_seq = 0;
    // Passes below this number rendered frames into caller-provided buffers that the library can no longer
    // render again (older frames of a tr_scene_render_frames call, frames of an automatic group before its last):
    // a bin overflow among them is reported, like one in a frame that was handed on
    uint64_t unreplayable_seq = 0;
    // The current targets (aliases: the memory belongs to the frame slots below, or to the caller)
    float *d_z = nullptr, *d_shadow = nullptr;
    uint32_t *d_sclean = nullptr;  // the shadow buffer's fast-clear flags (n_tiles_full)
    uint8_t *d_fb = nullptr;      // where the next render writes
    // Frame slots: complete sets of render targets.  Slot 0 is what the scene is created with; the others
    // appear with the first tr_scene_render_frames, whose frame i goes to slot i % (frames per group).
    // `fb` is the library's own colour buffer of the slot (allocated when first needed: callers may
    // bring their own); slots of pipelines without a depth pass share slot 0's shadow buffer.
    struct FrameSlot {
        float *z = nullptr;
        uint32_t *zclean = nullptr;
        float *shadow = nullptr;
        uint32_t *sclean = nullptr;
        uint8_t *fb = nullptr;
        // Transient depth: the slot's frame was rendered from cleared targets with its depth left on the chip
        // (TileArgs::store): z memory and fast-clear flags say nothing; the frame's z is what `z_params` renders.
        // ensure_depth() repeats the colour pass for the depth alone when somebody wants the z buffer.
        bool z_deferred = false;
        tr_frame_params z_params = {};
    };
    std::vector<FrameSlot> slots;
    int cur_slot = 0;
    uint32_t frames_per_launch = 0;  // tr_options.frames_per_launch; 0 = by tile count
    uint32_t max_slots = 0;          // tr_options.max_frame_slots; 0 = automatic
    bool transient_depth = true;     // a cleared frame's colour pass leaves its depth on the chip (TR_OPT_STORE_DEPTH / TR_DEFER_Z=0: off)
    bool no_long_runs = false;       // the large groups' resources did not fit the device once: the usual groups from then on
    bool broken = false;             // a tile kernel could not be launched behind its chain: counters and ranges are stale
    // One group in flight: bins, counters, work lists and argument tables of its frames' passes
    struct GroupSet {
        Piece *bins = nullptr;      // [pass][frame] x pool_cap records
        Piece *recs = nullptr;      // [pass][frame] x polygons
        uint32_t *count = nullptr;  // [pass][frame] x (n_tiles_full + 16)
        WorkItem *order = nullptr;  // [pass][frame] x n_tiles_full
        uint8_t *d_tables = nullptr, *h_tables = nullptr;  // [pass] x frames SetupArgs, then [pass] x frames TileArgs
        hipEvent_t ev_setup = nullptr, ev_tile = nullptr;
        bool in_flight = false;
        uint32_t pool_cap = 0, frames = 0;  // what the set was allocated for
        uint32_t g = 0;                    // frames of the group it holds now
        int tile_waves[2] = { 4, 4 }, shared[2] = { 0, 0 };  // the tile kernels' layout, per pass (decided with the setup)
        bool chain_on_main = false;        // its setup was queued on the main stream itself (nothing was in flight)
        uint32_t *lit = nullptr;           // [frame] x lit_words: the frames' lit texel images (scenes with the lit path)
        // [pass][frame] x 8 words, page-locked and mapped: the list lengths of the passes this set last held
        // (k_bin_group writes them; group_units sizes later tile kernels' grids by them)
        volatile uint32_t *h_lens = nullptr;
        uint32_t *d_lens = nullptr;
    } grp[GROUP_SETS];
    uint64_t group_seq = 0;       // groups whose setup has been queued
    uint64_t group_submitted = 0; // groups whose tile kernels have been queued (<= group_seq)
    bool groups_unfenced = false; // group tile kernels were queued since the setup stream was last ordered behind them
    bool quiescent = false;       // the host has waited for the main stream and queued nothing since
    // Automatic frame groups: cleared frames rendered through the per-frame calls on the library's own stream
    // are held back until a group is full (or anything needs them) and then rendered by fused launches.
    struct DeferredFrame {
        tr_frame_params p;
        uint8_t *fb;  // the colour target that was current at its render()
    };
    std::vector<DeferredFrame> deferred;
    bool auto_group = true;
    uint32_t auto_streak = 0;  // groups that filled up inside a running loop since anything else needed the frames
    // The frames of the last tr_scene_render_frames call that still exist (the last `frames per group` of
    // them): what tr_scene_select_frame chooses from, and what is rendered again after a bin overflow.
    struct {
        std::vector<tr_frame_params> params;
        std::vector<void *> fbs;  // the caller's buffers, or empty
        std::vector<int> slot;
        uint64_t first_seq = 0;   // pass number of the first of them
    } tail;
    bool last_was_group = false;
    uint8_t *d_view = nullptr;  // scratch for get_z_buffer / get_shadow_buffer
    uint32_t *d_winner = nullptr;
    // Fast depth clear: one word per colour-pass tile, non-zero = "every z of the tile is f32::MIN,
    // memory not written".  Raised by the tile kernel for the empty tiles of a cleared frame (and by
    // a clear that has to be materialised), lowered by whoever writes the tile's z.
    uint32_t *d_zclean = nullptr;
    // Colour counterpart, one flag set per frame buffer the scene has rendered into (its own, and the
    // ones a caller alternates through tr_scene_set_frame_buffer_device): non-zero = the tile's colour
    // and winner words in THAT buffer hold the cleared value, so an empty tile of a cleared frame is
    // not stored again.  The buffer belongs to the scene while it is the target: whoever else writes
    // its rows must call tr_scene_set_frame_buffer_device again (which forgets the flags of a buffer
    // it has not seen, and keeps those of one it has).
    struct FbFlags {
        uint8_t *fb;
        uint32_t *clean;
    };
    std::vector<FbFlags> fb_flags;
    uint32_t *d_fbclean = nullptr;  // the current target's set
    // Sparse read-back (tr_scene_get_frame_buffer_async into tr_host_alloc memory): per host buffer, which tiles of it
    // hold zeros since this scene last wrote them there (device memory, updated by k_read_back)
    struct HostFlags {
        void *host;
        uint64_t serial;
        uint32_t *clean;
        uint64_t gen;  // HostAlloc::gen as this scene's last read-back into the buffer left it
    };
    std::vector<HostFlags> host_flags;
    uint64_t *d_stamps = nullptr;
    uint32_t *d_err = nullptr;
    // page-locked host word the kernels set beside d_err (TileArgs::alarm), and its device address
    volatile uint32_t *h_alarm = nullptr;
    uint32_t *d_alarm = nullptr;

    // Lazy clear (scene.rs:128-137): `clear` only records that the targets are logically
    // f32::MIN / 0; the next render writes every pixel of them anyway, and a getter that comes
    // first materialises the values.
    bool z_fb_cleared = false;    // z buffer, frame buffer (and winner tap) are logically cleared
    bool shadow_cleared = false;  // shadow buffer is logically cleared

    tr_uniforms uniforms = {};
    int host_status = TR_OK;  // sticky failure of the last render's host-side prepare

    // What the last render() started from, so that a frame whose bins overflowed can be
    // rendered again after the bins have grown.
    struct {
        float light[3], from[3], at[3], up[3];
        bool z_fb_cleared, shadow_cleared;
        bool valid;
    } last = {};

    bool profiling = false;
    std::vector<EventPair> events;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[K_COUNT] = {};
    uint64_t prof_n[K_COUNT] = {};
    uint64_t prof_frames[K_COUNT] = {};
    std::vector<float> frame_intervals_us;  // completion-to-completion time of consecutive colour-pass tile kernels
};

namespace {

template <typename T>
int dev_alloc(T **p, size_t count)
{
    const hipError_t e = hipMalloc(reinterpret_cast<void **>(p), (count ? count : 1) * sizeof(T));
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();  // (not sticky: the caller may go on with less)
        *p = nullptr;
        char buf[96];
        snprintf(buf, sizeof buf, "out of device memory (%zu bytes)", (count ? count : 1) * sizeof(T));
        return tr::fail(TR_E_NOMEM, buf);
    }
    HIP_TRY(e);
    return TR_OK;
}

template <typename T>
void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

// Points d_fbclean at the flag set of the current frame buffer (allocated zeroed = "content unknown"
// the first time a buffer is seen; at most a handful of buffers are remembered).
// The flag set of frame buffer `fb` (allocated zeroed = "content unknown" the first time a buffer is seen;
// a few dozen buffers are remembered: a caller's double-buffered groups of frames, the scene's own slots).
int fb_flags_for(tr_scene *s, uint8_t *fb, uint32_t **out, bool forget = false)
{
    for (const tr_scene::FbFlags &f : s->fb_flags)
        if (f.fb == fb) {
            // `forget`: a caller's buffer handed over again without TR_OPT_TRUST_FRAME_BUFFERS -- anybody may have
            // written it since: "content unknown" (on the main stream: after the tile kernels that used the flags)
            if (forget) HIP_TRY(hipMemsetAsync(f.clean, 0, (size_t)s->n_tiles * 4, s->stream));
            *out = f.clean;
            return TR_OK;
        }
    if (s->fb_flags.size() >= 4u * GROUP_MAX + 8u) {  // forget the oldest that is not the current target's
        (void)hipStreamSynchronize(s->stream);
        size_t k = 0;
        while (k + 1 < s->fb_flags.size() && s->fb_flags[k].clean == s->d_fbclean) k++;
        dev_free(s->fb_flags[k].clean);
        s->fb_flags.erase(s->fb_flags.begin() + (long)k);
    }
    tr_scene::FbFlags f = { fb, nullptr };
    int st = dev_alloc(&f.clean, (size_t)s->n_tiles);
    if (st != TR_OK) return st;
    HIP_TRY(hipMemsetAsync(f.clean, 0, (size_t)s->n_tiles * 4, s->stream));
    s->fb_flags.push_back(f);
    *out = f.clean;
    return TR_OK;
}

// Is `fb` one of the scene's own colour buffers (whose remembered flags always hold)?
bool own_frame_buffer(const tr_scene *s, const uint8_t *fb)
{
    for (const tr_scene::FrameSlot &fs : s->slots)
        if (fs.fb == fb) return true;
    return false;
}

int select_fb_flags(tr_scene *s, bool handed_over = false)
{
    uint32_t *clean = nullptr;
    const bool forget = handed_over && !(s->flags & TR_OPT_TRUST_FRAME_BUFFERS) && !own_frame_buffer(s, s->d_fb);
    int st = fb_flags_for(s, s->d_fb, &clean, forget);
    if (st != TR_OK) return st;
    // the winner tap is one buffer shared by all targets: its tiles were last written with another
    // target's frame, so a remembered "clean" says nothing about them
    if (s->d_winner && s->d_fbclean != clean) HIP_TRY(hipMemsetAsync(clean, 0, (size_t)s->n_tiles * 4, s->stream));
    s->d_fbclean = clean;
    return TR_OK;
}

// The library's own colour buffer of a slot, zero-filled when it is created (Scene::new, scene.rs:71).
int slot_own_fb(tr_scene *s, int k, uint8_t **out)
{
    tr_scene::FrameSlot &fs = s->slots[(size_t)k];
    if (!fs.fb) {
        const size_t n = (size_t)s->width * s->height * 3;
        int st = dev_alloc(&fs.fb, n);
        if (st != TR_OK) return st;
        HIP_TRY(hipMemsetAsync(fs.fb, 0, n, s->stream));
        uint32_t *clean = nullptr;
        st = fb_flags_for(s, fs.fb, &clean);
        if (st != TR_OK) return st;
        if (!s->d_winner) HIP_TRY(hipMemsetAsync(clean, 0xFF, (size_t)s->n_tiles * 4, s->stream));  // zeros = the cleared colour
    }
    *out = fs.fb;
    return TR_OK;
}

// Makes slot k the current set of targets; colour goes to `fb` (a caller's buffer) or, if null, to the
// slot's own buffer.
int use_slot(tr_scene *s, int k, uint8_t *fb, bool handed_over = false)
{
    const tr_scene::FrameSlot &fs = s->slots[(size_t)k];
    s->cur_slot = k;
    s->d_z = fs.z;
    s->d_zclean = fs.zclean;
    s->d_shadow = fs.shadow;
    s->d_sclean = fs.sclean;
    if (!fb) {
        int st = slot_own_fb(s, k, &fb);
        if (st != TR_OK) return st;
    }
    s->d_fb = fb;
    return select_fb_flags(s, handed_over);
}

// The z buffer of frame slot k, allocated the first time a pass will read or write it.  The slots of frame groups
// (ensure_slots) start without one: a cleared frame's colour pass leaves its depth on the chip (transient depth), so a
// slot's z memory is touched only by a depth-only repeat (ensure_depth), an accumulating render or a scene that stores
// its depth -- 64 MiB per slot at 4096^2 that a running loop never uses (32 slots: 2 of the scene's 5 GB).
int need_z(tr_scene *s, int k)
{
    tr_scene::FrameSlot &fs = s->slots[(size_t)k];
    if (!fs.z) {
        int st = dev_alloc(&fs.z, (size_t)s->width * s->height);
        if (st != TR_OK) return st;
    }
    if (k == s->cur_slot) s->d_z = fs.z;
    return TR_OK;
}

hipEvent_t take_event(tr_scene *s)
{
    if (!s->event_pool.empty()) {
        hipEvent_t e = s->event_pool.back();
        s->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct Timed {
    tr_scene *s;
    hipStream_t st;
    EventPair ep;
    bool on;
    Timed(tr_scene *sc, int kernel, hipStream_t stream = nullptr) : s(sc), st(stream ? stream : sc->stream), on(false)
    {
        ep.kernel = kernel;
        ep.frames = 1u;
        ep.a = ep.b = nullptr;
        if (s->profiling && s->events.size() < (1u << 20)) {
            ep.a = take_event(s);
            ep.b = take_event(s);
            if (ep.a && ep.b && hipEventRecord(ep.a, st) == hipSuccess) on = true;
        }
    }
    ~Timed()
    {
        if (on && hipEventRecord(ep.b, st) == hipSuccess) s->events.push_back(ep);
    }
};

int drain_events(tr_scene *s)
{
    hipEvent_t prev_frame_end = nullptr;
    for (const EventPair &ep : s->events) {
        float ms = 0.0f;
        if (hipEventSynchronize(ep.b) == hipSuccess && hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            s->prof_ms[ep.kernel] += ms;
            s->prof_n[ep.kernel] += 1;
            s->prof_frames[ep.kernel] += ep.frames;
        }
        if (ep.kernel == K_TILE) {
            // a frame ends with its colour pass: the spacing of those completions is the frame time
            // of the running pipeline (frames overlap; a kernel's own duration says less); a fused
            // launch completes its frames together: each gets an equal share of the spacing
            if (prev_frame_end && hipEventElapsedTime(&ms, prev_frame_end, ep.b) == hipSuccess)
                for (uint32_t k = 0; k < ep.frames && s->frame_intervals_us.size() < (1u << 20); k++)
                    s->frame_intervals_us.push_back(ms * 1000.0f / (float)ep.frames);
            prev_frame_end = ep.b;
        }
        s->event_pool.push_back(ep.a);
        s->event_pool.push_back(ep.b);
    }
    s->events.clear();
    return TR_OK;
}

int launch_status(int rc, const char *what)
{
    if (rc == 0) return TR_OK;
    return tr::fail(TR_E_HIP, std::string(what) + ": " + hipGetErrorString((hipError_t)rc));
}

// Enqueues one pending pass's tile kernel on the main stream.
int launch_pending_tile(tr_scene *s, const tr_scene::PendingTile &t)
{
    int status = TR_OK;
    if (!s->profiling) {
        int rc = launch_tile(t.fs, t.args, t.tile_waves, t.shared, s->mesh.n_tri, nullptr, 0, s->stream, nullptr, s->ev_tile[t.p_seq % RING], 0u,
                             t.fused_single);
        if (rc) status = launch_status(rc, "k_tile");
        if (rc) s->broken = true;  // (its chain has run: the set's counters were not zeroed, the pass's ranges never consumed)
    } else {
        EventPair ep = { take_event(s), take_event(s), t.kernel_id, 1u };
        int rc = launch_tile(t.fs, t.args, t.tile_waves, t.shared, s->mesh.n_tri, nullptr, 0, s->stream, ep.a, ep.b, 0u, t.fused_single);
        if (rc) status = launch_status(rc, "k_tile");
        if (rc) s->broken = true;
        s->events.push_back(ep);
        if (hipEventRecord(s->ev_tile[t.p_seq % RING], s->stream) != hipSuccess && status == TR_OK)
            status = tr::fail(TR_E_HIP, "hipEventRecord");
    }
    s->tiles_submitted += 1;
    s->last_submitted_seq = t.p_seq;
    return status;
}

// Puts ALL pending per-frame tile kernels on the main stream: one wait for the setup stream (it is in
// order, so the newest pass's event covers the older ones), then the kernels back to back.
int submit_pending_tiles(tr_scene *s)
{
    if (s->pending.empty()) return TR_OK;
    // (each setup stream is in order: its newest pass's event covers the older ones)
    HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_setup[s->pending.back().p_seq % RING], 0));
    if (s->two_setup_streams && s->pending.size() > 1u)
        HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_setup[s->pending[s->pending.size() - 2u].p_seq % RING], 0));
    int status = TR_OK;
    for (const tr_scene::PendingTile &t : s->pending) {
        int st = launch_pending_tile(s, t);
        if (st != TR_OK && status == TR_OK) status = st;
    }
    s->pending.clear();
    return status;
}

// The oldest pending pass alone, behind a wait for its own setup if that may still be running.
int submit_front(tr_scene *s, bool wait_for_setup)
{
    if (s->pending.empty()) return TR_OK;
    const tr_scene::PendingTile t = s->pending.front();
    if (wait_for_setup) HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_setup[t.p_seq % RING], 0));
    s->pending.erase(s->pending.begin());
    return launch_pending_tile(s, t);
}

// Every pending pass whose setup has completed, oldest first, without any cross-stream wait.
int submit_ready(tr_scene *s)
{
    int status = TR_OK;
    while (!s->pending.empty() && hipEventQuery(s->ev_setup[s->pending.front().p_seq % RING]) == hipSuccess) {
        int st = submit_front(s, false);
        if (st != TR_OK && status == TR_OK) status = st;
    }
    return status;
}

int flush_deferred(tr_scene *s, bool hold_back);
int finish_groups(tr_scene *s);

// Everything the scene has been asked to render goes to the device: frames `render` has held back to fuse
// them (below, "Automatic frame groups"), per-frame tile kernels waiting for their setup, groups' tile
// kernels.  Runs before every use of the main stream and before anything waits for the scene.
int submit_pending(tr_scene *s)
{
    int st = flush_deferred(s, false);
    int st2 = submit_pending_tiles(s);
    int st3 = finish_groups(s);
    return st != TR_OK ? st : st2 != TR_OK ? st2 : st3;
}

// Materialise a pending clear of the z / frame buffers (and winner tap).
int flush_clear_color(tr_scene *s)
{
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    if (!s->z_fb_cleared) return TR_OK;
    // only the rows this scene owns: in a band-sharded frame the rest belongs to other ranks
    const size_t W = s->width;
    const size_t n = W * (size_t)(s->frame.band_y1 - s->frame.band_y0);
    const size_t z_first = W * (size_t)s->frame.band_y0;
    const size_t fb_first = W * (size_t)(s->height - (uint32_t)s->frame.band_y1) * 3;
    Timed t(s, K_CLEAR);
    // z: raise every tile's fast-clear flag; colour (and the winner tap) are real memory
    HIP_TRY(hipMemsetAsync(s->d_zclean, 0xFF, (size_t)s->n_tiles * 4, s->stream));
    s->slots[(size_t)s->cur_slot].z_deferred = false;   // (the slot's z IS the cleared value now)
    HIP_TRY(hipMemsetAsync(s->d_fb + fb_first, 0, n * 3, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_fbclean, 0xFF, (size_t)s->n_tiles * 4, s->stream));
    if (s->d_winner) HIP_TRY(hipMemsetAsync(s->d_winner + z_first, 0xFF, n * 4, s->stream));
    s->z_fb_cleared = false;
    return TR_OK;
}

int flush_clear_shadow(tr_scene *s)
{
    if (!s->shadow_cleared) return TR_OK;
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    // a cleared shadow buffer = every tile's fast-clear flag up (lookups consult the flags; getters
    // materialise the values first)
    Timed t(s, K_CLEAR);
    HIP_TRY(hipMemsetAsync(s->d_sclean, 0xFF, (size_t)s->n_tiles_full * 4, s->stream));
    s->shadow_cleared = false;
    return TR_OK;
}

// Gives the shadow buffer plain-memory meaning for a getter.
int materialize_shadow(tr_scene *s)
{
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    int rc = launch_materialize_depth(s->d_shadow, s->d_sclean, s->frame_full, s->stream);
    if (rc) return launch_status(rc, "k_materialize_depth");
    return TR_OK;
}

// Gives the z buffer plain-memory meaning: tiles still behind their fast-clear flag get their f32::MIN.
int materialize_depth(tr_scene *s)
{
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    {
        int st = need_z(s, s->cur_slot);
        if (st != TR_OK) return st;
    }
    int rc = launch_materialize_depth(s->d_z, s->d_zclean, s->frame, s->stream);
    if (rc) return launch_status(rc, "k_materialize_depth");
    return TR_OK;
}

int render_frame(tr_scene *s);
int replay_tail(tr_scene *s);
int ensure_depth(tr_scene *s);

// Tiles the bins must cover: a band scene's colour passes touch its own rows only; the depth passes of
// shadow / occlusion fill the whole shadow buffer on every rank (shader.rs:774-778).

// How the waves of a tile divide the work when tr_options.tile_mode leaves it open (speed only).
int tile_mode_auto(int by_tile_count)
{
    static const int forced = getenv("TR_TILE_MODE") ? atoi(getenv("TR_TILE_MODE")) : 0;  // test hook: 1 columns, 2 shared
    if (forced == 1 || forced == 2) return forced == 2;
    return by_tile_count;
}

// A tile received more polygons than its bin holds (k_tile then works on the first bin_cap records
// only: a truncated frame).  Grow the bins to what the passes asked for, then:
//   * if an overflowed pass has already been handed to a consumer the library cannot call back --
//     an asynchronous read-back queued behind it, or a caller-provided stream, whose next
//     operation may have used the frame -- report TR_E_BIN_OVERFLOW: the caller renders (and
//     copies) again, now with bins that fit;
//   * otherwise only the last frame can still be observed: render it again from the state it
//     started in.  Exact when that frame started from cleared targets (the per-frame protocol of
//     app.rs:170-210); an accumulating render cannot be replayed and reports TR_E_BIN_OVERFLOW.
int recover_from_overflow(tr_scene *s, unsigned long long first_bad_seq)
{
    uint32_t need = 0;
    HIP_TRY(hipMemcpy(&need, s->d_bin_need, sizeof need, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(s->d_bin_need, 0, sizeof need));
    HIP_TRY(hipStreamSynchronize(nullptr));  // (the scene's streams do not wait for the null stream)
    const uint64_t cap = plan::grown_pool(s->pool_cap, need);   // (need = the pairs the hungriest pass wanted)
    {
        // per-frame sets + the frame groups' sets, if they exist
        uint64_t sets = LOOKAHEAD;
        for (const tr_scene::GroupSet &g : s->grp) sets += (uint64_t)g.frames * (uint64_t)kPipelines[s->pipeline].n_passes;
        if (cap < need || sets * cap * s->rec_pieces * 16ull > (128ull << 30))
            return tr::fail(TR_E_BIN_OVERFLOW, "the pools of polygon records would exceed 128 GiB");
    }
    HIP_TRY(hipStreamSynchronize(s->setup_stream));
    HIP_TRY(hipStreamSynchronize(s->setup_stream2));
    s->pool_cap = (uint32_t)cap;
    int st = TR_OK;
    for (int k = 0; k < LOOKAHEAD && st == TR_OK; k++) {
        dev_free(s->d_bins[k]);
        st = dev_alloc(&s->d_bins[k], (size_t)s->pool_cap * s->rec_pieces);
    }
    if (st != TR_OK) return st;
    // (the frame groups' bins follow bin_cap when their set is next used)
    // what to do about the frame(s) that were rendered with too small a pool (tr_plan.h, overflow_action)
    const PipelineDesc &pd = kPipelines[s->pipeline];
    plan::OverflowState os;
    os.first_bad_seq = first_bad_seq;
    os.observed_seq = s->observed_seq;
    os.unreplayable_seq = s->unreplayable_seq;
    os.last_was_group = s->last_was_group;
    os.last_valid = s->last.valid;
    os.last_started_cleared = s->last.z_fb_cleared && (pd.n_passes == 1 || s->last.shadow_cleared);
    switch (plan::overflow_action(os)) {
    case plan::OverflowAction::REPORT_CALLERS_BUFFER: {
        char buf[320];
        snprintf(buf, sizeof buf,
                 "triangle bins overflowed in a frame that went to a caller's buffer and cannot be rendered again (an older "
                 "frame of tr_scene_render_frames, or of a fused group of per-frame renders): that buffer holds a truncated "
                 "frame; the pools have been grown to %u records per pass: render it again", s->pool_cap);
        return tr::fail(TR_E_BIN_OVERFLOW, buf);
    }
    case plan::OverflowAction::REPORT_HANDED_ON: {
        char buf[320];
        snprintf(buf, sizeof buf,
                 "triangle bins overflowed in a frame that was already handed on (asynchronous read-back or caller's "
                 "stream): that frame is truncated; the pools have been grown to %u records per pass (a pass wanted %u, "
                 "pass %llu of %llu): render it again", s->pool_cap, need, first_bad_seq, (unsigned long long)s->pass_seq);
        return tr::fail(TR_E_BIN_OVERFLOW, buf);
    }
    case plan::OverflowAction::REPLAY_TAIL:
        return replay_tail(s);  // frames older than the tail no longer exist anywhere
    case plan::OverflowAction::REPORT_ACCUMULATING:
        return tr::fail(TR_E_BIN_OVERFLOW,
                        "triangle bins overflowed during an accumulating render; they have been grown: clear and render again");
    case plan::OverflowAction::REPLAY_LAST:
        break;
    }
    float keep[12];
    memcpy(keep, s->light, 12); memcpy(keep + 3, s->from, 12); memcpy(keep + 6, s->at, 12); memcpy(keep + 9, s->up, 12);
    memcpy(s->light, s->last.light, 12); memcpy(s->from, s->last.from, 12);
    memcpy(s->at, s->last.at, 12); memcpy(s->up, s->last.up, 12);
    const bool z_now = s->z_fb_cleared, sh_now = s->shadow_cleared;
    s->z_fb_cleared = true;
    s->shadow_cleared = s->last.shadow_cleared;
    st = render_frame(s);
    // a clear() issued after the overflowing render stays pending
    s->z_fb_cleared = s->z_fb_cleared || z_now;
    s->shadow_cleared = s->shadow_cleared || sh_now;
    memcpy(s->light, keep, 12); memcpy(s->from, keep + 3, 12); memcpy(s->at, keep + 6, 12); memcpy(s->up, keep + 9, 12);
    return st;
}

// Reads and resets the device error word and the overflow bookkeeping (stream idle).
int take_device_errors(tr_scene *s, uint32_t &err, unsigned long long &first_bad_seq)
{
    err = 0;
    first_bad_seq = ~0ull;
    if (*s->h_alarm == 0u) return TR_OK;  // nothing was raised since the last look: no copy
    *s->h_alarm = 0u;
    HIP_TRY(hipMemcpy(&err, s->d_err, sizeof err, hipMemcpyDeviceToHost));
    if (err) {
        HIP_TRY(hipMemset(s->d_err, 0, sizeof err));  // the word is per frame, not sticky
        HIP_TRY(hipStreamSynchronize(nullptr));         // (the scene's streams do not wait for the null stream)
    }
    if (err & DE_BIN_OVERFLOW) {
        HIP_TRY(hipMemcpy(&first_bad_seq, s->d_overflow_seq, sizeof first_bad_seq, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(s->d_overflow_seq, 0xFF, sizeof first_bad_seq));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return TR_OK;
}

// Waits for the stream and folds the device error word into a status.
int sync_and_status(tr_scene *s)
{
    HIP_TRY(hipSetDevice(s->device));
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    uint32_t err = 0;
    unsigned long long first_bad = ~0ull;
    {
        int st = take_device_errors(s, err, first_bad);
        if (st != TR_OK) return st;
    }
    const int host_status = s->host_status;
    if (err & DE_BIN_OVERFLOW) {
        int st = recover_from_overflow(s, first_bad);
        s->observed_seq = 0;  // everything issued so far has completed; later hand-offs count afresh
        s->unreplayable_seq = 0;
        if (st != TR_OK) return st;
        st = submit_pending(s);
        if (st != TR_OK) return st;
        HIP_TRY(hipStreamSynchronize(s->stream));
        st = take_device_errors(s, err, first_bad);
        if (st != TR_OK) return st;
        if (err & DE_BIN_OVERFLOW) return tr::fail(TR_E_BIN_OVERFLOW, "triangle bins overflowed twice");
    }
    s->observed_seq = 0;
    s->unreplayable_seq = 0;
    s->quiescent = true;  // (whatever a getter queues next -- a view kernel, a copy -- it also waits for)
    if (host_status != TR_OK) return host_status;
    if (err & (DE_W_ZERO | DE_TEX_OOB | DE_SHADOW_OOB | DE_SINGULAR)) {
        char buf[160];
        snprintf(buf, sizeof buf,
                 "device lookups left their range (bits 0x%x: 1 w==0, 2 texture, 4 shadow buffer, 8 singular basis)",
                 err);
        return tr::fail(TR_E_OOB_LOOKUP, buf);
    }
    return TR_OK;
}

void fill_dev_uniforms(const tr_scene *s, DevUniforms &d)
{
    const tr_uniforms &u = s->uniforms;
    memcpy(d.vpmv, u.vpmv, sizeof d.vpmv);
    memcpy(d.m, u.m, sizeof d.m);
    memcpy(d.it_m, u.it_m, sizeof d.it_m);
    memcpy(d.shadow_matrix, u.shadow_matrix, sizeof d.shadow_matrix);
    memcpy(d.i_vpmv, u.i_vpmv, sizeof d.i_vpmv);
    memcpy(d.camera_direction, u.camera_direction, sizeof d.camera_direction);
    memcpy(d.t_light, u.t_light_direction, sizeof d.t_light);
    memset(d.sm_ivpmv, 0, sizeof d.sm_ivpmv);
    memset(d.occl_steps, 0, sizeof d.occl_steps);
}

// The frame constants of one pass (shader.rs:183-279 prepares) for the light and camera the scene holds.
int pass_uniforms(tr_scene *s, const PassDesc &p, DevUniforms &du)
{
    int st = prepare_uniforms(p.prepare_kind, &s->uniforms, s->width, s->height, s->light, s->from, s->at, s->up);
    if (st != TR_OK) return tr::fail(st, "prepare: singular matrix (try_inverse().unwrap() would panic)");
    fill_dev_uniforms(s, du);
    if (p.fs == FS_SHADOW2 || p.fs == FS_OCCLUSION2) shadow_times_inverse(&s->uniforms, du.sm_ivpmv);
    if (p.fs == FS_OCCLUSION2) {
        st = occlusion_steps(&s->uniforms, du.occl_steps);
        if (st != TR_OK) return tr::fail(st, "occlusion: rotation_between(..).unwrap() would panic");
    }
    return TR_OK;
}

// The fragment variant the tile kernel runs for a pass whose closure is `fs`: with the lit path (k_lit) the normal-map
// and specular closures have run per texel, and the fragment stage fetches their result.
int tile_fs(const tr_scene *s, int fs) { return (s->lit && (fs == FS_NORMAL_MAP || fs == FS_SPECULAR)) ? (int)FS_LIT : fs; }

// ... and what the pass's chain needs for it: where the frame's lit image goes, and the tile kernel's view of it.
void lit_args(const tr_scene *s, uint32_t *lit, SetupArgs &sa, TileArgs &ta)
{
    sa.lit = lit;
    sa.texel_set = s->tex.packed;
    sa.tex_w = s->tex.w[0];
    sa.tex_h = s->tex.h[0];
    sa.set_bpr = s->set_bpr;
    sa.lit_bpr = s->lit_bpr;
    ta.tex.packed = lit;        // (one word per texel, 8x4 blocks: what fetch_texels<FS_LIT> reads)
    ta.tex.packed_bpr = s->lit_bpr;
}

// How a tile's work is divided is a launch-time choice (speed only; tr_options.tile_waves / tile_mode
// pin it).  Few tiles cannot fill the GPU with four waves each: more waves per tile shorten every
// wave's serial chain, and sharing the bin between them (instead of giving each a column of the
// tile) keeps them equally loaded where polygons cluster.  Many tiles fill the GPU anyway: four
// waves with private columns do the least total work.  Measured on diablo / phong (k_tile us, best
// of the six combinations per size, profiles/r02_notes.md): 512^2 24.8 (16 shared; 4 columns 129),
// 1024^2 22.2 (16 shared), 2048^2 21.9 (8 shared), 2560^2 23.2 (8 columns), 4096^2 33.1 (4 columns).
// A fused launch (tr_scene_render_frames) fills the GPU with the tiles of ALL its frames, so its waves per
// tile go by that total; whether its waves share a tile's bin still goes by the size of ONE frame, which is
// what decides how unevenly the polygons fall on a tile's columns (800^2 african_head, 16 frames per launch,
// k_tile per frame: 4 waves columns 4.6 us, 4 waves shared 3.3, 8 shared 3.6, 16 shared 4.4) ...
void tile_layout(const tr_scene *s, uint64_t tiles_in_launch, uint32_t tiles_per_frame, int &tile_waves, int &shared)
{
    tile_waves = s->tile_waves ? (int)s->tile_waves : tiles_in_launch <= 1024u ? 16 : tiles_in_launch <= 4608u ? 8 : 4;
    // ... or by how many polygons a tile gets: many small ones are visited once per tile instead of once per
    // column they touch.  k_tile per frame, columns -> shared, polygons per tile of the frame: 8192^2 x64 grid
    // specular (9.8) 371 -> 350 us, the same with phong 259 -> 230, 4096^2 x16 (9.8) 66.9 -> 60.0, x9 (5.5)
    // 50.4 -> 47.6, x4 (2.5) 38.9 -> 38.5, one model (0.6) 33.5 -> 35.4
    const bool dense = (uint64_t)s->mesh.n_tri >= 3ull * (tiles_per_frame ? tiles_per_frame : 1u);
    shared = s->tile_mode ? (s->tile_mode == 2 ? 1 : 0) : tile_mode_auto((tiles_per_frame <= 2048u || dense) ? 1 : 0);
    // the shared keys pack polygon id and bin slot into 32 bits: beyond their fields, resolve by columns.  Decided
    // HERE, once per pass: k_setup prepares the pairs' masks for the form the tile kernel will run (SetupArgs::cells)
    if (s->mesh.n_tri > (1u << 20)) shared = 0;  // (a TILE with more records than the slot field holds resolves by columns: k_tile)
}

// May a cleared frame's colour pass leave its depth on the chip?  Not with the winner tap or the tile stamps (single
// diagnostic buffers whose tests read the z buffer after every frame: nothing to gain).
bool defer_depth(const tr_scene *s)
{
    static const int env_on = getenv("TR_DEFER_Z") ? atoi(getenv("TR_DEFER_Z")) : 1;
    return env_on && s->transient_depth && !s->d_winner && !s->d_stamps;
}

// (TR_FUSED_SINGLE=0: a per-frame pass runs the per-frame kernels whatever its targets' state, as before round 4)
static bool fused_single_launches()
{
    static const bool on = !getenv("TR_FUSED_SINGLE") || atoi(getenv("TR_FUSED_SINGLE")) != 0;
    return on;
}

// depth_only: the repeat of a colour pass for its depth alone (ensure_depth): from cleared targets, nothing of the
// scene's clear / colour state is consumed or changed.
int run_pass(tr_scene *s, const PassDesc &p, bool depth_only = false)
{
    DevUniforms du;
    int st = pass_uniforms(s, p, du);
    if (st != TR_OK) return st;
    // a per-frame pass after frame groups: their tile kernels first (it may render onto their targets)
    if (s->group_submitted < s->group_seq || s->groups_unfenced) {
        st = finish_groups(s);
        if (st != TR_OK) return st;
    }

    const bool depth_pass = (p.fs == FS_DEPTH);
    // Which targets does this pass write, and are they logically cleared?
    uint32_t fresh;
    if (depth_pass) {
        fresh = s->shadow_cleared ? 1u : 0u;
        s->shadow_cleared = false;
    } else if (depth_only) {
        fresh = 1u;
    } else {
        fresh = s->z_fb_cleared ? 1u : 0u;
        s->z_fb_cleared = false;
        // a colour pass that reads the shadow buffer needs real values in it
        if (p.fs == FS_SHADOW2 || p.fs == FS_OCCLUSION2) {
            st = flush_clear_shadow(s);
            if (st != TR_OK) return st;
        }
    }

    const DevFrame &frame = depth_pass ? s->frame_full : s->frame;

    const uint32_t n_tiles_pass = frame.ntx * frame.nty;
    tr_scene::PendingTile pt;
    tile_layout(s, n_tiles_pass, n_tiles_pass, pt.tile_waves, pt.shared);

    SetupArgs sa = {};   // (every member defined: the kernels take the struct by value)
    sa.mesh = s->mesh;
    sa.frame = frame;
    sa.u = du;
    sa.cells = pt.shared ? 1u : 0u;
    tr_scene::BinState &bs = depth_pass ? s->bin_depth : s->bin_color;
    const int set_cur = (int)(bs.seq % SETS);
    const uint64_t p_seq = s->pass_seq;
    Piece *bins = s->d_bins[p_seq % LOOKAHEAD];
    sa.tile_count = bs.count[set_cur];
    sa.recs = s->d_recs[p_seq % LOOKAHEAD];
    sa.bins = bins;
    sa.pool_cap = s->pool_cap;
    sa.rec_pieces = s->rec_pieces;
    sa.err = s->d_err;
    sa.alarm = s->d_alarm;
    TileArgs ta = {};
    ta.bins = bins;
    ta.pool_cap = s->pool_cap;
    ta.rec_pieces = s->rec_pieces;
    ta.order = s->d_order[p_seq % LOOKAHEAD];
    ta.tile_count = bs.count[set_cur];
    ta.bin_need = s->d_bin_need;
    ta.overflow_seq = s->d_overflow_seq;
    ta.pass_seq = p_seq;
    ta.frame = frame;
    ta.u = du;
    ta.tex = s->tex;
    ta.zbuf = s->d_z;
    ta.shadow = s->d_shadow;
    ta.fb = s->d_fb;
    ta.winner = s->d_winner;
    ta.zclean = depth_pass ? s->d_sclean : s->d_zclean;
    ta.fbclean = depth_pass ? nullptr : s->d_fbclean;
    ta.sclean = (p.fs == FS_SHADOW2 || p.fs == FS_OCCLUSION2) ? s->d_sclean : nullptr;
    ta.err = s->d_err;
    ta.alarm = s->d_alarm;
    ta.fresh = fresh;
    ta.store = TR_STORE_DEPTH | TR_STORE_COLOR;
    if (!depth_pass) {
        tr_scene::FrameSlot &slot = s->slots[(size_t)s->cur_slot];
        if (depth_only) {
            ta.store = TR_STORE_DEPTH;
            ta.winner = nullptr;
            ta.stamps = nullptr;
            slot.z_deferred = false;
        } else if (fresh && !s->d_winner && defer_depth(s) && fused_single_launches()) {
            // a cleared frame on its own: the fused launches' kernel for one frame (launch_tile, fused_single), and like
            // a group's frames it leaves its depth on the chip; whoever wants the z buffer gets it from ensure_depth
            ta.store = TR_STORE_COLOR;
            slot.z_deferred = true;
            memcpy(slot.z_params.light, s->light, 12); memcpy(slot.z_params.look_from, s->from, 12);
            memcpy(slot.z_params.look_at, s->at, 12); memcpy(slot.z_params.up, s->up, 12);
        } else {
            // (an accumulating render has had the slot's depth made real before it came here: ensure_depth, tr_scene_render)
            slot.z_deferred = false;
        }
    }
    pt.fused_single = !depth_only && fresh != 0u && !s->d_winner && fused_single_launches();
    if (!depth_pass && ((ta.store & TR_STORE_DEPTH) || fresh == 0u)) {   // the pass reads or writes z memory
        st = need_z(s, s->cur_slot);
        if (st != TR_OK) return st;
        ta.zbuf = s->d_z;
    }
    ta.aligned16 = (s->width % 16u == 0u) ? 1u : 0u;
    ta.aligned4 = (s->width % 4u == 0u) ? 1u : 0u;
    if (!depth_only) ta.stamps = depth_pass ? nullptr : s->d_stamps;
    const bool lit_pass = !depth_only && tile_fs(s, p.fs) == (int)FS_LIT;
    if (lit_pass) lit_args(s, s->d_lit[p_seq % LOOKAHEAD], sa, ta);
    // setup on its own stream: after the tile kernel of pass p - LOOKAHEAD, before the tile kernel of pass p.
    // With NOTHING in flight (a frame rendered and read, rendered and read: the interactive loop) the chain has
    // nothing to overlap with, and the hop between the streams -- event, wait packet, dispatch: 13-17 us -- is
    // pure latency: then setup, work list and tiles go down the main stream in order.
    // "Nothing in flight" is what the host KNOWS after it has waited for the stream (a loop that merely runs
    // ahead of a fast GPU must keep its overlap: asking the stream instead cost small frames 7-16 %).
    const bool chain_on_main = s->own_stream && s->quiescent && s->pending.empty() && s->group_submitted == s->group_seq;
    s->quiescent = false;
    hipStream_t chain = chain_on_main ? s->stream : (s->two_setup_streams && (p_seq & 1u)) ? s->setup_stream2 : s->setup_stream;
    if (!chain_on_main && p_seq >= (uint64_t)LOOKAHEAD)
        HIP_TRY(hipStreamWaitEvent(chain, s->ev_tile[(p_seq - LOOKAHEAD) % RING], 0));
    // the chain: vertex stage + counting, work lists + pool ranges, records into the ranges
    if (lit_pass) {  // (first in the chain: it needs nothing but the frame's constants)
        EventPair el = { nullptr, nullptr, K_LIT, 1u };
        if (s->profiling) { el.a = take_event(s); el.b = take_event(s); }
        int rc = launch_lit(p.fs, sa, nullptr, 0, chain, el.a, el.b);
        if (rc) return launch_status(rc, "k_lit");
        if (s->profiling) s->events.push_back(el);
    }
    if (!s->profiling) {
        int rc = launch_setup(p.vs, sa, nullptr, 0, chain_on_main, chain, nullptr, nullptr);
        if (rc) return launch_status(rc, "k_setup");
        rc = launch_order(ta, n_tiles_pass, nullptr, 0, chain, nullptr, s->mesh.n_tri ? nullptr : s->ev_setup[p_seq % RING]);
        if (rc) return launch_status(rc, "k_order");
        rc = launch_bin(sa, nullptr, 0, chain_on_main, chain, nullptr, s->ev_setup[p_seq % RING]);
        if (rc) return launch_status(rc, "k_bin");
    } else {
        // profiling: timing events on the dispatches themselves, then the pipeline's event separately
        EventPair ep = { take_event(s), take_event(s), K_SETUP, 1u };
        int rc = launch_setup(p.vs, sa, nullptr, 0, chain_on_main, chain, ep.a, ep.b);
        if (rc) return launch_status(rc, "k_setup");
        s->events.push_back(ep);
        EventPair eo = { take_event(s), take_event(s), K_ORDER, 1u };
        rc = launch_order(ta, n_tiles_pass, nullptr, 0, chain, eo.a, eo.b);
        if (rc) return launch_status(rc, "k_order");
        s->events.push_back(eo);
        if (s->mesh.n_tri) {
            EventPair eb = { take_event(s), take_event(s), K_BIN, 1u };
            rc = launch_bin(sa, nullptr, 0, chain_on_main, chain, eb.a, eb.b);
            if (rc) return launch_status(rc, "k_bin");
            s->events.push_back(eb);
        }
        HIP_TRY(hipEventRecord(s->ev_setup[p_seq % RING], chain));
    }
    pt.fs = depth_only ? p.fs : tile_fs(s, p.fs);
    pt.kernel_id = depth_pass ? K_TILE_DEPTH : K_TILE;
    pt.p_seq = p_seq;
    pt.args = ta;
    s->pending.push_back(pt);
    bs.seq++;
    s->pass_seq++;
    // a caller's stream must hold the frame when render() returns (its next operation may consume
    // it: from here on the pass counts as handed on, see recover_from_overflow)
    if (!s->own_stream) {
        s->observed_seq = s->pass_seq;
        return submit_pending_tiles(s);
    }
    // the library's own stream: see "Handing tile kernels to the main stream" above
    if (chain_on_main) return submit_front(s, false);  // its setup is ahead of it in the same queue
    // (tr_plan.h, handover: the three cases of "Handing tile kernels to the main stream" above)
    switch (plan::handover(s->pending.size(), s->tiles_submitted == 0,
                           s->tiles_submitted != 0 && hipEventQuery(s->ev_tile[s->last_submitted_seq % RING]) == hipSuccess)) {
    case plan::Handover::HOST_WAITS: {
        // steady state: the HOST waits for the oldest pending pass's setup (it completed long ago, or will
        // within microseconds: the setup stream runs ahead, and the main stream still holds the tile kernels
        // of the passes before it), then its tile kernel goes out with no wait packet in the GPU's queue.
        // This is also what keeps the host from running ahead of the GPU without bound.
        int status = TR_OK;
        while ((int)s->pending.size() > BATCH) {
            HIP_TRY(hipEventSynchronize(s->ev_setup[s->pending.front().p_seq % RING]));
            int st2 = submit_front(s, false);
            if (st2 != TR_OK && status == TR_OK) status = st2;
        }
        return status;
    }
    case plan::Handover::FRONT_WITH_WAIT:
        // start-up (the first passes after a sync): a main stream that has run dry gets the next tile kernel
        // behind a wait packet (an idle GPU loses nothing to it) ...
        return submit_front(s, true);
    case plan::Handover::READY_ONLY:
        break;
    }
    // ... otherwise whatever has its setup behind it goes out without one
    return submit_ready(s);
}

int render_frame(tr_scene *s)
{
    s->host_status = TR_OK;
    const PipelineDesc &pd = kPipelines[s->pipeline];
    for (int i = 0; i < pd.n_passes; i++) {
        int st = run_pass(s, pd.pass[i]);
        if (st != TR_OK) {
            s->host_status = st;
            return st;
        }
    }
    return TR_OK;
}

// Makes the current slot's z buffer real.  A cleared frame's colour pass leaves its depth on the chip (transient depth:
// nothing reads the z buffer of such a frame -- the next cleared frame overwrites it unseen); the first consumer that
// does want it -- tr_scene_read_z_f32 / tr_scene_get_z_buffer, a render WITHOUT a clear, which depth-tests against it
// (scene.rs:151) -- repeats the frame's colour pass for the depth alone: same light and camera, same cull and
// projection, same resolve; colour, winner words, flags of the frame buffer and the scene's clear state untouched.
int ensure_depth(tr_scene *s)
{
    tr_scene::FrameSlot &slot = s->slots[(size_t)s->cur_slot];
    if (!slot.z_deferred) return TR_OK;
    int st = submit_pending(s);   // (the frame itself, should it still be held back)
    if (st != TR_OK) return st;
    if (!slot.z_deferred) return TR_OK;
    float keep[12];
    memcpy(keep, s->light, 12); memcpy(keep + 3, s->from, 12); memcpy(keep + 6, s->at, 12); memcpy(keep + 9, s->up, 12);
    const tr_frame_params &q = slot.z_params;
    memcpy(s->light, q.light, 12); memcpy(s->from, q.look_from, 12); memcpy(s->at, q.look_at, 12); memcpy(s->up, q.up, 12);
    const PipelineDesc &pd = kPipelines[s->pipeline];
    const int host_status = s->host_status;
    st = run_pass(s, pd.pass[pd.n_passes - 1], true);
    if (st == TR_OK) st = submit_pending_tiles(s);
    s->host_status = host_status;
    memcpy(s->light, keep, 12); memcpy(s->from, keep + 3, 12); memcpy(s->at, keep + 6, 12); memcpy(s->up, keep + 9, 12);
    return st;
}


// ---------------------------------------------------------------------------------------------
// Frame groups (tr_scene_render_frames)
// ---------------------------------------------------------------------------------------------

// Frames per fused launch: enough tiles to keep the machine full while one frame's light tiles drain and to
// make the gap between launches small against the launch -- 32 K tiles (4096^2: 4 frames, 8 no better; 3072^2
// 4 -> 8 frames 19.0 -> 18.1 us per frame, 2048^2 8 -> 16 10.4 -> 10.0, 1536^2 14 -> 28 7.2 -> 6.8), within
// GROUP_MAX; at least four frames.  The winner tap and the tile stamps are single buffers: one frame.
plan::SceneShape shape_of(const tr_scene *s)
{
    static const int forced = getenv("TR_GROUP") ? atoi(getenv("TR_GROUP")) : 0;  // experiment hook
    plan::SceneShape q;
    q.n_tiles = s->n_tiles;
    q.frames_per_launch = s->frames_per_launch;
    q.max_slots = s->max_slots;
    q.forced = forced > 0 ? (uint32_t)forced : 0u;
    q.winner_tap = s->d_winner != nullptr;
    q.tile_stamps = s->d_stamps != nullptr;
    q.no_long_runs = s->no_long_runs;
    q.n_passes = (uint32_t)kPipelines[s->pipeline].n_passes;
    q.pool_bytes_per_pass = (uint64_t)s->pool_cap * s->rec_pieces * 16ull;
    q.pixels = (uint64_t)s->width * s->height;
    return q;
}

uint32_t group_size(const tr_scene *s) { return plan::group_size(shape_of(s)); }

// Frames per launch that a LONG run of frames grows to (tr_scene_render_frames with sixteen groups or more; the
// per-frame protocol after sixteen full groups in a row): between two tile kernels of a stream lie 6-12 us (the end of
// a kernel that wrote 200 MB, the dispatch of the next), paid once per launch -- 32 frames per launch instead of 4 make
// 4096^2 30.1 -> 28.7 us per frame.  Only when the group size is automatic, within 8 GiB of frame slots and 16 GiB of
// record pools.  (Not for short runs: a slot's first frames are slower than its later ones -- 20 frames into 20 slots:
// tile kernel 31.2 us per frame, into 4 slots used five times each: 29.1.)
uint32_t long_run_group_size(const tr_scene *s) { return plan::long_run_group_size(shape_of(s)); }

// Frame slots 1 .. n - 1 (slot 0 exists since tr_scene_create).  Their z memory needs no initial value:
// a slot is only ever reached through a group's cleared frame, which writes every tile or raises its flag.
int ensure_slots(tr_scene *s, uint32_t n)
{
    const size_t npx = (size_t)s->width * s->height;
    while (s->slots.size() < n) {
        tr_scene::FrameSlot fs;
        // (no z buffer yet: need_z)
        int st = dev_alloc(&fs.zclean, (size_t)s->n_tiles);
        if (st == TR_OK && kPipelines[s->pipeline].n_passes == 2) st = dev_alloc(&fs.shadow, npx);
        if (st == TR_OK && kPipelines[s->pipeline].n_passes == 2) st = dev_alloc(&fs.sclean, (size_t)s->n_tiles_full);
        if (st != TR_OK) {
            dev_free(fs.z);
            dev_free(fs.zclean);
            dev_free(fs.shadow);
            dev_free(fs.sclean);
            return st;
        }
        if (!fs.shadow) {  // never written without a depth pass
            fs.shadow = s->slots[0].shadow;
            fs.sclean = s->slots[0].sclean;
        }
        s->slots.push_back(fs);
    }
    return TR_OK;
}

size_t group_bins_per_frame(const tr_scene *s) { return (size_t)s->pool_cap * s->rec_pieces; }
size_t group_recs_per_frame(const tr_scene *s) { return (size_t)(s->mesh.n_tri ? s->mesh.n_tri : 1u) * s->rec_pieces; }
size_t group_counts_per_frame(const tr_scene *s) { return (size_t)s->n_tiles_full + 16u; }

void free_group_set(tr_scene::GroupSet &gs);

int ensure_group_set(tr_scene *s, tr_scene::GroupSet &gs, uint32_t frames)
{
    const size_t np = (size_t)kPipelines[s->pipeline].n_passes;
    if (!gs.ev_setup) {
        HIP_TRY(hipEventCreateWithFlags(&gs.ev_setup, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&gs.ev_tile, hipEventDisableTiming));
    }
    int st = TR_OK;
    if ((gs.frames < frames || gs.pool_cap != s->pool_cap) && gs.in_flight) {
        HIP_TRY(hipEventSynchronize(gs.ev_tile));  // its memory is about to be replaced
        gs.in_flight = false;
    }
    if (gs.frames < frames) {
        free_group_set(gs);
        const size_t nc = np * frames * group_counts_per_frame(s);
        if ((st = dev_alloc(&gs.count, nc))) return st;
        // (from here on every tile kernel workgroup zeroes its tile's counter.)  hipMemset returns before the device has
        // executed it and the scene's streams do not wait for the null stream: without the synchronisation the
        // set's first k_setup could count on top of whatever the fresh allocation held (an intermittent "bin
        // overflow" the first time a fourth group was in flight)
        HIP_TRY(hipMemset(gs.count, 0, nc * 4));
        HIP_TRY(hipStreamSynchronize(nullptr));
        if ((st = dev_alloc(&gs.order, np * frames * (size_t)s->n_tiles_full * ORDER_LISTS))) return st;
        if ((st = dev_alloc(&gs.recs, np * frames * group_recs_per_frame(s)))) return st;
        dev_free(gs.lit);
        if (s->lit && (st = dev_alloc(&gs.lit, np * frames * (size_t)s->lit_words))) return st;
        const size_t tb = np * frames * (sizeof(SetupArgs) + sizeof(TileArgs));
        if ((st = dev_alloc(&gs.d_tables, tb))) return st;
        HIP_TRY(hipHostMalloc((void **)&gs.h_tables, tb, hipHostMallocDefault));
        {
            void *h = nullptr, *d = nullptr;
            HIP_TRY(hipHostMalloc(&h, np * frames * LEN_WORDS * sizeof(uint32_t), hipHostMallocMapped));
            memset(h, 0, np * frames * LEN_WORDS * sizeof(uint32_t));
            HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
            gs.h_lens = (volatile uint32_t *)h;
            gs.d_lens = (uint32_t *)d;
        }
        gs.frames = frames;
    }
    if (gs.pool_cap != s->pool_cap) {
        dev_free(gs.bins);
        gs.pool_cap = 0;
        if ((st = dev_alloc(&gs.bins, np * gs.frames * group_bins_per_frame(s)))) return st;
        gs.pool_cap = s->pool_cap;
    }
    return TR_OK;
}

int slot_own_fb(tr_scene *s, int k, uint8_t **out);

// Everything a long run's large groups need -- frame slots with their colour buffers, the sets of the groups in
// flight -- allocated once, the first time the scene sees frames in bulk (a tr_scene_render_frames call of more than
// one group, or a group that filled up under the per-frame protocol): a few milliseconds and, at 4096^2, 4 GiB that a
// later, longer run must not pay for in its own time (a warm-up of five frames, then the timed twenty or two thousand).
void free_group_set(tr_scene::GroupSet &gs)
{
    dev_free(gs.count);
    dev_free(gs.order);
    dev_free(gs.recs);
    dev_free(gs.lit);
    dev_free(gs.d_tables);
    if (gs.h_tables) (void)hipHostFree(gs.h_tables);
    gs.h_tables = nullptr;
    if (gs.h_lens) (void)hipHostFree((void *)gs.h_lens);
    gs.h_lens = nullptr;
    gs.d_lens = nullptr;
    dev_free(gs.bins);
    gs.pool_cap = 0;
    gs.frames = 0;
}

int prepare_long_runs(tr_scene *s, bool own_colour)
{
    const uint32_t g = long_run_group_size(s);
    if (g <= group_size(s) || s->d_winner) return TR_OK;
    const size_t slots_before = s->slots.size();
    int st = ensure_slots(s, g);
    for (uint32_t k = 0; k < g && st == TR_OK && own_colour; k++) {
        uint8_t *unused = nullptr;
        st = slot_own_fb(s, (int)k, &unused);
    }
    for (int k = 0; k < GROUP_SETS && st == TR_OK; k++) st = ensure_group_set(s, s->grp[k], g);
    if (st != TR_E_NOMEM) return st;
    // The device has no room for a long run's resources (a smaller GPU, several scenes or ranks on one): give back
    // what was taken just now -- nothing has used it -- and go on with the usual groups, for good.
    const std::string why = tr::g_last_error;
    (void)hipStreamSynchronize(s->stream);  // (the new colour buffers' zero fill)
    const uint32_t G = group_size(s);
    while (s->slots.size() > slots_before && s->slots.size() > (size_t)G) {
        tr_scene::FrameSlot &fs = s->slots.back();
        if (fs.fb)
            for (size_t k = 0; k < s->fb_flags.size(); k++)
                if (s->fb_flags[k].fb == fs.fb && s->fb_flags[k].clean != s->d_fbclean) {
                    dev_free(s->fb_flags[k].clean);
                    s->fb_flags.erase(s->fb_flags.begin() + (long)k);
                    break;
                }
        dev_free(fs.z);
        dev_free(fs.zclean);
        if (fs.shadow != s->slots[0].shadow) {
            dev_free(fs.shadow);
            dev_free(fs.sclean);
        }
        dev_free(fs.fb);
        s->slots.pop_back();
    }
    for (tr_scene::GroupSet &gs : s->grp)
        if (!gs.in_flight && gs.frames > G) free_group_set(gs);
    s->no_long_runs = true;
    tr::g_last_error = "large frame groups disabled for this scene: " + why;
    return TR_OK;
}

int submit_groups(tr_scene *s, bool all);

// (TR_WORK_UNITS=0: every tile kernel is launched with one workgroup per tile, as before round 4)
static bool use_work_units()
{
    static const bool on = !getenv("TR_WORK_UNITS") || atoi(getenv("TR_WORK_UNITS")) != 0;
    return on;
}

// Workgroups per frame for the tile kernels of pass `pi` of the group in `gs`, whose chain HAS COMPLETED (the caller has
// seen its event), or 0: one per tile.  The tile kernel needs one workgroup per WORK UNIT -- a tile with polygons, or
// EMPTY_CHUNK empty tiles (tr_kernels.hip) -- and k_bin_group, the chain's last kernel, has left every frame's list
// lengths in page-locked memory: the frame that needs most decides (a fused launch's frames share the grid).
static uint32_t group_units(const tr_scene *s, const tr_scene::GroupSet &gs, uint32_t pi, uint32_t n_tiles_pass)
{
    if (!gs.h_lens || s->mesh.n_tri == 0 || gs.g > (uint32_t)GROUP_MAX) return 0u;  // (no polygons: no k_bin)
    uint32_t lens[GROUP_MAX * LEN_WORDS];
    const volatile uint32_t *w = gs.h_lens + (size_t)pi * gs.frames * LEN_WORDS;
    for (uint32_t i = 0; i < gs.g * LEN_WORDS; i++) lens[i] = w[i];
    return plan::group_grid_units(lens, gs.g, n_tiles_pass);
}

// Queues the setup of g <= frames-per-group cleared frames, one launch per kernel and pass.  Frame j takes
// light and camera from p[j], its targets from slot slot_of[j] and its colour buffer from fbs[j] (fbs == null:
// the slot's own).  The tile kernels follow with submit_groups.
int run_group(tr_scene *s, const tr_frame_params *p, void *const *fbs, const int *slot_of, uint32_t g, bool forget_callers_buffers = false)
{
    const PipelineDesc &pd = kPipelines[s->pipeline];
    const uint32_t np = (uint32_t)pd.n_passes;
    tr_scene::GroupSet &gs = s->grp[s->group_seq % GROUP_SETS];
    // nothing in flight (the first group after a sync): its setup goes down the main stream, ahead of its tile
    // kernels, without the hop between the streams (see run_pass)
    const bool chain_on_main = s->own_stream && s->quiescent && s->pending.empty() && s->group_submitted == s->group_seq;
    s->quiescent = false;
    // the set's previous group must have left the GPU before its bins, counters and tables are written again;
    // this wait is also what keeps the host from running ahead of the GPU without bound
    if (gs.in_flight) {
        // (about to wait on the host: whatever tile kernels are still held back go to the device first -- a wait
        // packet in front of them costs less than a GPU that runs dry while the host sleeps)
        if (s->group_seq - s->group_submitted >= (uint64_t)GROUP_SETS || hipEventQuery(gs.ev_tile) != hipSuccess) {
            int sg = submit_groups(s, true);
            if (sg != TR_OK) return sg;
        }
        HIP_TRY(hipEventSynchronize(gs.ev_tile));
    }
    gs.in_flight = false;
    int st = ensure_group_set(s, gs, g > group_size(s) ? g : group_size(s));
    if (st != TR_OK) return st;
    const uint32_t G = gs.frames;
    if (g == 0 || g > G) return tr::fail(TR_E_INVALID, "group larger than its set");

    SetupArgs *h_setup = reinterpret_cast<SetupArgs *>(gs.h_tables);
    TileArgs *h_tile = reinterpret_cast<TileArgs *>(gs.h_tables + (size_t)np * G * sizeof(SetupArgs));
    const SetupArgs *d_setup = reinterpret_cast<const SetupArgs *>(gs.d_tables);
    const TileArgs *d_tile = reinterpret_cast<const TileArgs *>(gs.d_tables + (size_t)np * G * sizeof(SetupArgs));

    for (uint32_t pi = 0; pi < np; pi++) {
        const DevFrame &fr = pd.pass[pi].fs == FS_DEPTH ? s->frame_full : s->frame;
        tile_layout(s, (uint64_t)fr.ntx * fr.nty * g, fr.ntx * fr.nty, gs.tile_waves[pi], gs.shared[pi]);
    }
    float keep[12];
    memcpy(keep, s->light, 12); memcpy(keep + 3, s->from, 12); memcpy(keep + 6, s->at, 12); memcpy(keep + 9, s->up, 12);
    for (uint32_t j = 0; j < g && st == TR_OK; j++) {
        memcpy(s->light, p[j].light, 12); memcpy(s->from, p[j].look_from, 12);
        memcpy(s->at, p[j].look_at, 12); memcpy(s->up, p[j].up, 12);
        if (!defer_depth(s)) st = need_z(s, slot_of[j]);   // (the frames store their depth)
        if (st != TR_OK) break;
        tr_scene::FrameSlot &slot = s->slots[(size_t)slot_of[j]];
        slot.z_deferred = defer_depth(s);   // (what the slot's z is then: this frame)
        slot.z_params = p[j];
        uint8_t *fb = fbs ? (uint8_t *)fbs[j] : nullptr;
        const bool callers = fb != nullptr;
        if (!fb) st = slot_own_fb(s, slot_of[j], &fb);
        uint32_t *fbclean = nullptr;
        if (st == TR_OK) st = fb_flags_for(s, fb, &fbclean, callers && forget_callers_buffers);
        for (uint32_t pi = 0; pi < np && st == TR_OK; pi++) {
            const PassDesc &pass = pd.pass[pi];
            const bool depth_pass = (pass.fs == FS_DEPTH);
            const size_t e = (size_t)pi * G + j;  // this frame-pass's share of the set
            SetupArgs &sa = h_setup[e];
            TileArgs &ta = h_tile[e];
            memset(&sa, 0, sizeof sa);
            memset(&ta, 0, sizeof ta);
            st = pass_uniforms(s, pass, sa.u);
            if (st != TR_OK) break;
            const DevFrame &frame = depth_pass ? s->frame_full : s->frame;
            sa.mesh = s->mesh;
            sa.frame = frame;
            sa.tile_count = gs.count + e * group_counts_per_frame(s);
            sa.recs = gs.recs + e * group_recs_per_frame(s);
            sa.bins = gs.bins + e * group_bins_per_frame(s);
            sa.pool_cap = s->pool_cap;
            sa.rec_pieces = s->rec_pieces;
            sa.err = s->d_err;
            sa.alarm = s->d_alarm;
            sa.cells = gs.shared[pi] ? 1u : 0u;
            sa.len_src = d_tile[e].list_len;
            sa.len_host = gs.d_lens + e * LEN_WORDS;
            ta.bins = sa.bins;
            ta.pool_cap = s->pool_cap;
            ta.rec_pieces = s->rec_pieces;
            ta.order = gs.order + e * (size_t)s->n_tiles_full * ORDER_LISTS;
            ta.tile_count = sa.tile_count;
            ta.bin_need = s->d_bin_need;
            ta.overflow_seq = s->d_overflow_seq;
            ta.pass_seq = s->pass_seq + (uint64_t)j * np + pi;  // frame by frame, as the per-frame path numbers them
            ta.frame = frame;
            ta.u = sa.u;
            ta.tex = s->tex;
            ta.zbuf = slot.z;
            ta.shadow = slot.shadow;
            ta.fb = fb;
            ta.winner = nullptr;
            ta.zclean = depth_pass ? slot.sclean : slot.zclean;
            ta.fbclean = depth_pass ? nullptr : fbclean;
            ta.sclean = (pass.fs == FS_SHADOW2 || pass.fs == FS_OCCLUSION2) ? slot.sclean : nullptr;
            ta.err = s->d_err;
            ta.alarm = s->d_alarm;
            ta.fresh = 1u;  // every frame of a group starts from cleared targets
            ta.store = (!depth_pass && defer_depth(s)) ? TR_STORE_COLOR : (TR_STORE_DEPTH | TR_STORE_COLOR);
            ta.aligned16 = (s->width % 16u == 0u) ? 1u : 0u;
            ta.aligned4 = (s->width % 4u == 0u) ? 1u : 0u;
            ta.stamps = depth_pass ? nullptr : s->d_stamps;
            if (tile_fs(s, pass.fs) == (int)FS_LIT) lit_args(s, gs.lit + e * (size_t)s->lit_words, sa, ta);
        }
    }
    memcpy(s->light, keep, 12); memcpy(s->from, keep + 3, 12); memcpy(s->at, keep + 6, 12); memcpy(s->up, keep + 9, 12);
    if (st != TR_OK) return st;

    // tables to the device, then per pass: vertex stage + binning of all frames, their work lists
    hipStream_t chain = chain_on_main ? s->stream : s->setup_stream;
    HIP_TRY(hipMemcpyAsync(gs.d_tables, gs.h_tables, (size_t)np * G * (sizeof(SetupArgs) + sizeof(TileArgs)),
                           hipMemcpyHostToDevice, chain));
    for (uint32_t pi = 0; pi < np; pi++) {
        const PassDesc &pass = pd.pass[pi];
        const SetupArgs &sa0 = h_setup[(size_t)pi * G];
        const uint32_t n_tiles_pass = sa0.frame.ntx * sa0.frame.nty;
        EventPair ep = { nullptr, nullptr, K_SETUP, g }, eo = { nullptr, nullptr, K_ORDER, g }, eb = { nullptr, nullptr, K_BIN, g };
        if (s->profiling) {
            ep.a = take_event(s); ep.b = take_event(s);
            eo.a = take_event(s); eo.b = take_event(s);
            eb.a = take_event(s); eb.b = take_event(s);
        }
        int rc = 0;
        if (tile_fs(s, pass.fs) == (int)FS_LIT) {
            EventPair el = { nullptr, nullptr, K_LIT, g };
            if (s->profiling) { el.a = take_event(s); el.b = take_event(s); }
            rc = launch_lit(pass.fs, sa0, d_setup + (size_t)pi * G, g, chain, el.a, el.b);
            if (rc) return launch_status(rc, "k_lit");
            if (s->profiling) s->events.push_back(el);
        }
        rc = launch_setup(pass.vs, sa0, d_setup + (size_t)pi * G, g, chain_on_main, chain, ep.a, ep.b);
        if (rc) return launch_status(rc, "k_setup");
        rc = launch_order(h_tile[(size_t)pi * G], n_tiles_pass, d_tile + (size_t)pi * G, g, chain, eo.a, eo.b);
        if (rc) return launch_status(rc, "k_order");
        rc = launch_bin(sa0, d_setup + (size_t)pi * G, g, chain_on_main, chain, eb.a, eb.b);
        if (rc) return launch_status(rc, "k_bin");
        if (s->profiling) {
            s->events.push_back(ep);
            s->events.push_back(eo);
            if (s->mesh.n_tri) s->events.push_back(eb);
            else { s->event_pool.push_back(eb.a); s->event_pool.push_back(eb.b); }
        }
    }
    HIP_TRY(hipEventRecord(gs.ev_setup, chain));
    gs.chain_on_main = chain_on_main;
    gs.g = g;
    gs.in_flight = true;
    s->pass_seq += (uint64_t)g * np;
    s->group_seq++;
    return TR_OK;
}

// The tile kernels of the oldest group whose setup is queued, pass by pass (a colour pass reads what its
// frame's depth pass wrote), behind a wait for that setup unless the caller knows it has completed.
int submit_group_tiles(tr_scene *s, bool wait_for_setup)
{
    if (s->group_submitted == s->group_seq) return TR_OK;
    const PipelineDesc &pd = kPipelines[s->pipeline];
    const uint32_t np = (uint32_t)pd.n_passes;
    tr_scene::GroupSet &gs = s->grp[s->group_submitted % GROUP_SETS];
    const uint32_t G = gs.frames, g = gs.g;
    const TileArgs *h_tile = reinterpret_cast<const TileArgs *>(gs.h_tables + (size_t)np * G * sizeof(SetupArgs));
    const TileArgs *d_tile = reinterpret_cast<const TileArgs *>(gs.d_tables + (size_t)np * G * sizeof(SetupArgs));
    s->group_submitted++;
    if (wait_for_setup) HIP_TRY(hipStreamWaitEvent(s->stream, gs.ev_setup, 0));
    // (has the chain completed?  "no wait needed" alone does not say so: a chain on the main stream itself needs none either)
    const bool chain_done = !wait_for_setup && !gs.chain_on_main && hipEventQuery(gs.ev_setup) == hipSuccess;
    for (uint32_t pi = 0; pi < np; pi++) {
        const PassDesc &pass = pd.pass[pi];
        const TileArgs &ta0 = h_tile[(size_t)pi * G];
        const int tile_waves = gs.tile_waves[pi], shared = gs.shared[pi];
        EventPair ep = { nullptr, nullptr, pass.fs == FS_DEPTH ? K_TILE_DEPTH : K_TILE, g };
        if (s->profiling) {
            ep.a = take_event(s);
            ep.b = take_event(s);
        }
        // (the group's "tiles done" event rides on its last tile kernel's own completion signal: a separate record
        // is one more packet between this group's tile kernel and the next one's)
        const bool last = pi + 1u == np;
        // Workgroups per frame: exactly the pass's work units when its chain has completed (the host has seen the event:
        // the lists' lengths are in page-locked memory), else one per tile
        const uint32_t units = (chain_done && use_work_units()) ? group_units(s, gs, pi, ta0.frame.ntx * ta0.frame.nty) : 0u;
        int rc = launch_tile(tile_fs(s, pass.fs), ta0, tile_waves, shared, s->mesh.n_tri, d_tile + (size_t)pi * G, g, s->stream, ep.a,
                             (!s->profiling && last) ? gs.ev_tile : ep.b, units);
        if (rc) {
            s->broken = true;  // (the group's chains have run: counters not zeroed, ranges never consumed)
            return launch_status(rc, "k_tile");
        }
        if (s->profiling) s->events.push_back(ep);
    }
    if (s->profiling) HIP_TRY(hipEventRecord(gs.ev_tile, s->stream));
    s->groups_unfenced = true;
    if (!s->own_stream) s->observed_seq = s->pass_seq;  // a caller's stream: handed on (see recover_from_overflow)
    return TR_OK;
}

// Hands groups to the main stream, oldest first.  As with single frames ("Handing tile kernels to the main
// stream" above) a cross-stream wait packet between two tile kernels costs microseconds and is not needed
// when the setup has already completed, so inside a long call a group's tile kernels go out one or two
// groups after its setup was queued -- by then the setup stream, which runs in the gaps and tails of the tile
// kernels before, has finished it.  `all`: the end of a call -- everything goes out, behind a wait if need be.
int submit_groups(tr_scene *s, bool all)
{
    int status = TR_OK;
    while (s->group_submitted < s->group_seq) {
        tr_scene::GroupSet &gs = s->grp[s->group_submitted % GROUP_SETS];
        bool ready = gs.chain_on_main || hipEventQuery(gs.ev_setup) == hipSuccess;  // (in order on the main stream: as good as done)
        if (!ready) {
            // a main stream that has run dry (the first group of a call, typically) gets the group at once, behind
            // a wait: an idle GPU loses nothing to the packet, and a call of a few frames is mostly start-up
            // (20 frames at 4096^2: 40.3 -> 33 us per frame)
            const bool idle = s->group_submitted == 0 ||
                              hipEventQuery(s->grp[(s->group_submitted - 1) % GROUP_SETS].ev_tile) == hipSuccess;
            // otherwise two groups are held back at most: the next setup needs the set of group_seq - GROUP_SETS,
            // whose tile kernels must be on the stream by then
            if (!all && !idle && s->group_seq - s->group_submitted <= 2) break;
            // It has to go now (the end of a call, or the host is about to sleep until a set is free).  While the main
            // stream still has a tile kernel to run, the chain -- which runs beside that kernel and is far shorter --
            // completes first: the HOST waits for it instead of a wait packet, and then knows the lists' lengths, i.e.
            // how many workgroups the tile kernel really needs (group_units: a third of the tiles at 4096^2).
            if (!idle && s->own_stream && use_work_units()) {
                HIP_TRY(hipEventSynchronize(gs.ev_setup));
                ready = true;
            }
        }
        int st = submit_group_tiles(s, !ready);
        if (st != TR_OK && status == TR_OK) status = st;
    }
    return status;
}

// All groups' tile kernels to the main stream, and -- once per batch of groups -- the setup stream ordered
// behind them: per-frame passes set up later reuse the per-frame bins and counters, which tile kernels queued
// so far may still read.
int finish_groups(tr_scene *s)
{
    int st = submit_groups(s, true);
    if (st != TR_OK) return st;
    if (s->groups_unfenced) {
        const tr_scene::GroupSet &gs = s->grp[(s->group_submitted + GROUP_SETS - 1) % GROUP_SETS];
        HIP_TRY(hipStreamWaitEvent(s->setup_stream, gs.ev_tile, 0));
        HIP_TRY(hipStreamWaitEvent(s->setup_stream2, gs.ev_tile, 0));
        s->groups_unfenced = false;
    }
    return TR_OK;
}

// ---------------------------------------------------------------------------------------------
// Automatic frame groups
// ---------------------------------------------------------------------------------------------
// The reference's caller renders frame after frame through clear -> set_* -> render (app.rs:170-213).  On the
// library's own stream nobody can observe a frame before the next getter / sync / flush, so `render` of a
// CLEARED frame only records the frame (light, camera, colour target) and returns; when a group's worth of
// frames has been recorded -- or anything needs them -- they are rendered together by the fused launches of
// run_group: the last one into the scene's current targets, the ones before it (which the per-frame protocol
// overwrites unobserved) into other frame slots.  A frame that must be seen alone (a getter right after it)
// goes through the ordinary per-frame path, so an interactive loop is what it was; a loop that keeps
// rendering gets the group rate without calling tr_scene_render_frames (4096^2 phong: 36 -> 30 us per frame).
// Not on a caller's stream (its next operation may consume the frame), not for accumulating renders, not with
// the winner tap or tile stamps (single buffers).  TR_OPT_NO_AUTO_GROUP / TR_AUTO_GROUP=0 turn it off.
bool frame_is_groupable(const tr_scene *s)
{
    static const int env_on = getenv("TR_AUTO_GROUP") ? atoi(getenv("TR_AUTO_GROUP")) : 1;
    return env_on && s->auto_group && s->own_stream && !s->d_winner && !s->d_stamps && s->z_fb_cleared &&
           (kPipelines[s->pipeline].n_passes == 1 || s->shadow_cleared) && group_size(s) > 1u;
}

// Renders the frames `render` has held back.  hold_back: a group has just filled up inside a running loop --
// its tile kernels may wait on the host until their setup has completed (submit_groups); otherwise everything
// is handed to the device now.
int flush_deferred(tr_scene *s, bool hold_back)
{
    if (!hold_back) s->auto_streak = 0;
    if (s->deferred.empty()) return TR_OK;
    std::vector<tr_scene::DeferredFrame> fr;
    fr.swap(s->deferred);
    const uint32_t g = (uint32_t)fr.size();
    int st = TR_OK;
    if (g == 1u) {
        // alone: the ordinary path, from the state the frame was recorded in
        float keep[12];
        memcpy(keep, s->light, 12); memcpy(keep + 3, s->from, 12); memcpy(keep + 6, s->at, 12); memcpy(keep + 9, s->up, 12);
        const bool z_now = s->z_fb_cleared, sh_now = s->shadow_cleared;
        uint8_t *fb_now = s->d_fb;
        const tr_frame_params &q = fr[0].p;
        memcpy(s->light, q.light, 12); memcpy(s->from, q.look_from, 12); memcpy(s->at, q.look_at, 12); memcpy(s->up, q.up, 12);
        s->z_fb_cleared = s->shadow_cleared = true;
        if (fr[0].fb != fb_now) st = use_slot(s, s->cur_slot, fr[0].fb == s->slots[(size_t)s->cur_slot].fb ? nullptr : fr[0].fb);
        if (st == TR_OK) st = render_frame(s);
        if (fr[0].fb != fb_now) {
            int st2 = use_slot(s, s->cur_slot, fb_now == s->slots[(size_t)s->cur_slot].fb ? nullptr : fb_now);
            if (st == TR_OK) st = st2;
        }
        // a clear() issued after that render stays pending
        s->z_fb_cleared = s->z_fb_cleared || z_now;
        s->shadow_cleared = s->shadow_cleared || sh_now;
        memcpy(s->light, keep, 12); memcpy(s->from, keep + 3, 12); memcpy(s->at, keep + 6, 12); memcpy(s->up, keep + 9, 12);
        return st;
    }
    // per-frame tile kernels issued before these frames go first (same targets)
    st = submit_pending_tiles(s);
    if (st != TR_OK) return st;
    const uint32_t G = group_size(s);
    if (hold_back && (st = prepare_long_runs(s, true)) != TR_OK) return st;  // (a loop that fills groups: see there)
    if ((st = ensure_slots(s, G < g ? g : G)) != TR_OK) return st;
    tr_frame_params params[GROUP_MAX];
    void *fbs[GROUP_MAX];
    int slot_of[GROUP_MAX];
    std::vector<const void *> wanted(g);
    for (uint32_t j = 0; j < g; j++) {
        params[j] = fr[j].p;
        wanted[j] = fr[j].fb;
    }
    // which slot and which colour target every frame gets (tr_plan.h, plan_deferred): the last one the current targets,
    // the others -- which the per-frame protocol overwrites unobserved -- other slots
    const std::vector<plan::DeferredTarget> targets = plan::plan_deferred(
        wanted, s->cur_slot, s->slots[(size_t)s->cur_slot].fb, [s](const void *fb) { return own_frame_buffer(s, (const uint8_t *)fb); });
    for (uint32_t j = 0; j < g; j++) {
        slot_of[j] = targets[j].slot;
        fbs[j] = const_cast<void *>(targets[j].fb);
        // (such a frame can be seen in its buffer but not rendered again: only the last frame is replayed)
        if (targets[j].unreplayable)
            s->unreplayable_seq = s->pass_seq + (uint64_t)(j + 1u) * (uint64_t)kPipelines[s->pipeline].n_passes;
    }
    st = run_group(s, params, fbs, slot_of, g);
    if (st == TR_OK) st = hold_back ? submit_groups(s, false) : finish_groups(s);
    return st;
}

// n cleared frames, frame i into slot i % G; afterwards the last one is the scene's current frame.
int render_frames(tr_scene *s, uint32_t n, const tr_frame_params *p, void *const *fbs)
{
    int st = submit_pending(s);  // per-frame renders issued before go first
    if (st != TR_OK) return st;
    const uint32_t G = group_size(s);
    const uint32_t np = (uint32_t)kPipelines[s->pipeline].n_passes;
    // The groups of the call, its frame slots, what every group set must hold (tr_plan.h, plan_call): a long call's
    // groups grow, a short call's double from the usual one.
    static const uint32_t growth = (uint32_t)(getenv("TR_GROUP_GROWTH") ? atoi(getenv("TR_GROUP_GROWTH")) : 4);      // experiment hooks
    static const uint32_t short_factor = (uint32_t)(getenv("TR_SHORT_GROUPS") ? atoi(getenv("TR_SHORT_GROUPS")) : 3);  // (plan_call clamps them)
    const bool automatic = !s->d_winner && !s->d_stamps && !s->frames_per_launch && !getenv("TR_GROUP");
    const plan::CallPlan cp = plan::plan_call(n, G, long_run_group_size(s), automatic, (int)growth < 2 ? 2u : growth,
                                              (int)short_factor < 1 ? 1u : short_factor);
    const std::vector<uint32_t> &sizes = cp.sizes;
    const uint32_t S = cp.slots;
    if (n > G && (st = prepare_long_runs(s, fbs == nullptr)) != TR_OK) return st;
    if ((st = ensure_slots(s, S)) != TR_OK) return st;
    // all the sets of groups in flight now (allocations of a few hundred MiB each: not in the middle of a call)
    for (int k = 0; k < GROUP_SETS && !s->d_winner; k++)
        if ((st = ensure_group_set(s, s->grp[k], cp.set_frames)) != TR_OK) return st;
    s->host_status = TR_OK;
    const uint64_t first_seq = s->pass_seq;
    if (s->d_winner) {
        // the winner tap is a single buffer: frame by frame through the ordinary path, slots all the same
        for (uint32_t i = 0; i < n; i++) {
            if ((st = use_slot(s, (int)(i % G), fbs ? (uint8_t *)fbs[i] : nullptr, fbs != nullptr)) != TR_OK) return st;
            memcpy(s->light, p[i].light, 12); memcpy(s->from, p[i].look_from, 12);
            memcpy(s->at, p[i].look_at, 12); memcpy(s->up, p[i].up, 12);
            s->z_fb_cleared = s->shadow_cleared = true;
            if ((st = render_frame(s)) != TR_OK) return st;
        }
    } else {
        int slot_of[GROUP_MAX];
        uint32_t i0 = 0;
        for (size_t k = 0; k < sizes.size(); i0 += sizes[k], k++) {
            const uint32_t g = sizes[k];
            for (uint32_t j = 0; j < g; j++) slot_of[j] = (int)((i0 + j) % S);
            st = run_group(s, p + i0, fbs ? fbs + i0 : nullptr, slot_of, g, fbs && !(s->flags & TR_OPT_TRUST_FRAME_BUFFERS));
            if (st == TR_OK) st = submit_groups(s, false);
            if (st != TR_OK) {
                // (a singular camera in frame i0 .. i0 + g - 1, or the device refused a launch): the groups before are
                // on their way, this one and the rest are not rendered; nothing of the call can be selected
                (void)finish_groups(s);
                s->host_status = st;
                s->tail.params.clear();
                s->tail.fbs.clear();
                s->tail.slot.clear();
                s->last_was_group = true;
                s->last.valid = false;
                return st;
            }
        }
        if ((st = finish_groups(s)) != TR_OK) return st;
        // the scene now stands where the per-frame calls would have left it
        const tr_frame_params &l = p[n - 1];
        memcpy(s->light, l.light, 12); memcpy(s->from, l.look_from, 12); memcpy(s->at, l.look_at, 12); memcpy(s->up, l.up, 12);
        s->z_fb_cleared = s->shadow_cleared = false;
        if ((st = use_slot(s, (int)((n - 1) % S), fbs ? (uint8_t *)fbs[n - 1] : nullptr)) != TR_OK) return st;
    }
    // what is left of the call: its last min(n, G) frames
    const uint32_t kept = cp.kept;
    s->tail.params.assign(p + (n - kept), p + n);
    s->tail.fbs.clear();
    if (fbs) s->tail.fbs.assign(fbs + (n - kept), fbs + n);
    s->tail.slot.resize(kept);
    for (uint32_t k = 0; k < kept; k++) s->tail.slot[k] = (int)((n - kept + k) % (s->d_winner ? G : S));
    s->tail.first_seq = first_seq + (uint64_t)(n - kept) * np;
    if (fbs && n > kept) s->unreplayable_seq = s->tail.first_seq;  // older frames' buffers: theirs for good
    s->last_was_group = true;
    s->last.valid = false;
    return TR_OK;
}

// After a bin overflow: the frames of the last call that still exist, again (the bins have grown).
int replay_tail(tr_scene *s)
{
    const uint32_t kept = (uint32_t)s->tail.params.size();
    if (kept == 0) return TR_OK;
    const int cur = s->cur_slot;
    uint8_t *cur_fb = s->d_fb;
    int st = TR_OK;
    if (s->d_winner) {
        for (uint32_t k = 0; k < kept && st == TR_OK; k++) {
            st = use_slot(s, s->tail.slot[k], s->tail.fbs.empty() ? nullptr : (uint8_t *)s->tail.fbs[k]);
            const tr_frame_params &q = s->tail.params[k];
            memcpy(s->light, q.light, 12); memcpy(s->from, q.look_from, 12); memcpy(s->at, q.look_at, 12); memcpy(s->up, q.up, 12);
            s->z_fb_cleared = s->shadow_cleared = true;
            if (st == TR_OK) st = render_frame(s);
        }
    } else {
        st = run_group(s, s->tail.params.data(), s->tail.fbs.empty() ? nullptr : s->tail.fbs.data(), s->tail.slot.data(), kept);
        if (st == TR_OK) st = finish_groups(s);
    }
    if (st != TR_OK) return st;
    return use_slot(s, cur, cur_fb == s->slots[(size_t)cur].fb ? nullptr : cur_fb);  // the selection the caller had
}

int find_pipeline(const char *name)
{
    if (!name) return -1;
    if (!strcmp(name, "true_normal")) name = "normal_map";  // README.md:18 spelling
    for (int i = 0; i < P_COUNT; i++)
        if (!strcmp(name, kPipelines[i].name)) return i;
    return -1;
}

void destroy(tr_scene *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)submit_pending(s);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (const EventPair &ep : s->events) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    for (hipEvent_t e : s->event_pool) (void)hipEventDestroy(e);
    dev_free(s->d_tri);
    for (int k = 0; k < 4; k++) dev_free(s->d_texel[k]);
    dev_free(s->d_packed);
    for (int k = 0; k < LOOKAHEAD; k++) dev_free(s->d_lit[k]);
    if (s->setup_stream) (void)hipStreamSynchronize(s->setup_stream);
    if (s->setup_stream2) (void)hipStreamSynchronize(s->setup_stream2);
    for (int k = 0; k < RING; k++) {
        if (s->ev_setup[k]) (void)hipEventDestroy(s->ev_setup[k]);
        if (s->ev_tile[k]) (void)hipEventDestroy(s->ev_tile[k]);
    }
    if (s->setup_stream) (void)hipStreamDestroy(s->setup_stream);
    if (s->setup_stream2) (void)hipStreamDestroy(s->setup_stream2);
    for (tr_scene::BinState *b : { &s->bin_color, &s->bin_depth }) {
        for (int k = 0; k < SETS; k++) {
            dev_free(b->count[k]);
        }
    }
    for (int k = 0; k < LOOKAHEAD; k++) {
        dev_free(s->d_order[k]);
        dev_free(s->d_bins[k]);
        dev_free(s->d_recs[k]);
    }
    dev_free(s->d_bin_need);
    dev_free(s->d_overflow_seq);
    for (size_t k = s->slots.size(); k-- > 0;) {  // slot 0 last: the others may share its shadow buffer
        tr_scene::FrameSlot &fs = s->slots[k];
        dev_free(fs.z);
        dev_free(fs.zclean);
        if (k == 0 || fs.shadow != s->slots[0].shadow) {
            dev_free(fs.shadow);
            dev_free(fs.sclean);
        }
        dev_free(fs.fb);
    }
    for (tr_scene::GroupSet &g : s->grp) {
        dev_free(g.bins);
        dev_free(g.recs);
        dev_free(g.lit);
        dev_free(g.count);
        dev_free(g.order);
        dev_free(g.d_tables);
        if (g.h_tables) (void)hipHostFree(g.h_tables);
        if (g.ev_setup) (void)hipEventDestroy(g.ev_setup);
        if (g.ev_tile) (void)hipEventDestroy(g.ev_tile);
    }
    dev_free(s->d_view);
    dev_free(s->d_winner);
    for (tr_scene::FbFlags &f : s->fb_flags) dev_free(f.clean);
    for (tr_scene::HostFlags &f : s->host_flags) dev_free(f.clean);
    dev_free(s->d_stamps);
    dev_free(s->d_err);
    if (s->h_alarm) (void)hipHostFree((void *)s->h_alarm);
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int create(uint32_t width, uint32_t height, const tr_mesh *mesh, const tr_image_rgb8 tex[4],
           const char *pipeline_name, const tr_options *opts, tr_scene *s)
{
    const int pipe = find_pipeline(pipeline_name);
    if (pipe < 0) return tr::fail(TR_E_UNKNOWN_PIPELINE, "Provided pipeline name is not supported!");
    if (width == 0 || height == 0 || width > 32768u || height > 32768u)
        return tr::fail(TR_E_INVALID, "frame size must be within 1..32768");
    if (!mesh || !tex) return tr::fail(TR_E_INVALID, "null mesh or textures");
    if (mesh->n_tri >= 0xFFFFFFF0u) return tr::fail(TR_E_INVALID, "too many polygons");
    for (uint32_t t = 0; t < mesh->n_tri; t++) {
        const uint32_t *ix = mesh->idx + 9u * (size_t)t;
        for (int k = 0; k < 3; k++)
            if (ix[3 * k] >= mesh->n_pos || ix[3 * k + 1] >= mesh->n_tex || ix[3 * k + 2] >= mesh->n_nrm)
                return tr::fail(TR_E_BAD_POLYGON, "polygon index outside positions / tex_coords / normals");
    }
    for (int k = 0; k < 4; k++)
        if (!tex[k].rgb || tex[k].w == 0 || tex[k].h == 0 || tex[k].w > 65535u || tex[k].h > 65535u)
            return tr::fail(TR_E_INVALID, "texture must be non-empty and at most 65535 on a side");

    tr_options o;
    memset(&o, 0, sizeof o);
    o.device = -1;
    if (opts) memcpy(&o, opts, opts->struct_size < sizeof o ? opts->struct_size : sizeof o);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return tr::fail(TR_E_HIP, "no HIP device available (this library has no CPU rendering path)");
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return tr::fail(TR_E_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(dev));
    s->device = dev;
    s->width = width;
    s->height = height;
    s->pipeline = pipe;
    s->flags = o.flags;
    if (o.tile_waves != 0 && o.tile_waves != 4 && o.tile_waves != 8 && o.tile_waves != 16)
        return tr::fail(TR_E_INVALID, "tile_waves must be 0, 4, 8 or 16");
    s->tile_waves = o.tile_waves;
    if (o.tile_mode > 2) return tr::fail(TR_E_INVALID, "tile_mode must be 0 (automatic), 1 (columns) or 2 (shared bin)");
    s->tile_mode = o.tile_mode;
    if (o.frames_per_launch > (uint32_t)GROUP_MAX) return tr::fail(TR_E_INVALID, "frames_per_launch must be 0 (automatic) or 1..32");
    s->frames_per_launch = o.frames_per_launch;
    if (o.max_frame_slots > (uint32_t)GROUP_MAX) return tr::fail(TR_E_INVALID, "max_frame_slots must be 0 (automatic) or 1..32");
    if (o.max_frame_slots && o.frames_per_launch > o.max_frame_slots)
        return tr::fail(TR_E_INVALID, "frames_per_launch exceeds max_frame_slots (every frame of a launch needs a slot)");
    s->max_slots = o.max_frame_slots;
    s->auto_group = (o.flags & TR_OPT_NO_AUTO_GROUP) == 0;
    s->transient_depth = (o.flags & TR_OPT_STORE_DEPTH) == 0;

    if (o.stream) {
        s->stream = (hipStream_t)o.stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        s->own_stream = true;
    }

    // band (output rows, row 0 = top) -> internal rows (row 0 = bottom)
    uint32_t r0 = o.band_row0, r1 = o.band_row1;
    if (r0 == 0 && r1 == 0) r1 = height;
    if (r0 >= r1 || r1 > height) return tr::fail(TR_E_INVALID, "band rows must satisfy row0 < row1 <= height");
    s->frame.width = width;
    s->frame.height = height;
    s->frame.band_y0 = (int32_t)(height - r1);
    s->frame.band_y1 = (int32_t)(height - r0);
    s->frame.ntx = (width + TILE_W - 1) / TILE_W;
    s->frame.ty_base = s->frame.band_y0 / TILE_H;
    s->frame.nty = (uint32_t)((s->frame.band_y1 - 1) / TILE_H - s->frame.ty_base + 1);
    s->n_tiles = s->frame.ntx * s->frame.nty;
    s->frame_full = s->frame;
    s->frame_full.band_y0 = 0;
    s->frame_full.band_y1 = (int32_t)height;
    s->frame_full.ty_base = 0;
    s->frame_full.nty = (height + TILE_H - 1) / TILE_H;
    s->n_tiles_full = s->frame_full.ntx * s->frame_full.nty;

    const size_t npx = (size_t)width * height;
    int st;
    // model: one flat row per polygon
    {
        std::vector<float> rows((size_t)mesh->n_tri * TRI_FLOATS);
        for (uint32_t t = 0; t < mesh->n_tri; t++)
            gather_polygon(mesh->pos, mesh->tex, mesh->nrm, mesh->idx + 9u * (size_t)t, &rows[(size_t)t * TRI_FLOATS]);
        if ((st = dev_alloc(&s->d_tri, rows.size()))) return st;
        HIP_TRY(hipMemcpy(s->d_tri, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    }
    s->mesh.tri = s->d_tri;
    s->mesh.n_tri = mesh->n_tri;

    // textures: rgb8 -> rgba8 so a texel is one aligned dword fetch
    std::vector<uint32_t> rgba[4];
    bool same_size = true;
    for (int k = 0; k < 4; k++) {
        const size_t n = (size_t)tex[k].w * tex[k].h;
        rgba[k].resize(n);
        for (size_t i = 0; i < n; i++)
            rgba[k][i] = (uint32_t)tex[k].rgb[3 * i] | ((uint32_t)tex[k].rgb[3 * i + 1] << 8) |
                         ((uint32_t)tex[k].rgb[3 * i + 2] << 16);
        if ((st = dev_alloc(&s->d_texel[k], n))) return st;
        HIP_TRY(hipMemcpy(s->d_texel[k], rgba[k].data(), n * 4, hipMemcpyHostToDevice));
        s->tex.texel[k] = s->d_texel[k];
        s->tex.w[k] = tex[k].w;
        s->tex.h[k] = tex[k].h;
        same_size = same_size && tex[k].w == tex[0].w && tex[k].h == tex[0].h;
    }
    // ... and, when they all have one size, the images the colour closure reads as ONE array: interleaved texel by
    // texel, tiled into 128-byte blocks (tr_texels.h; fetch_texels, tr_shaders.h)
    static const bool plain_texels = getenv("TR_PLAIN_TEXELS") && atoi(getenv("TR_PLAIN_TEXELS"));  // test hook
    if (same_size && !plain_texels) {
        const int fs = kPipelines[pipe].pass[kPipelines[pipe].n_passes - 1].fs;
        const uint32_t *const image[4] = { rgba[0].data(), rgba[1].data(), rgba[2].data(), rgba[3].data() };
        uint32_t bpr = 0;
        const std::vector<uint32_t> packed = pack_texels(fs, image, tex[0].w, tex[0].h, bpr);
        if ((st = dev_alloc(&s->d_packed, packed.size()))) return st;
        HIP_TRY(hipMemcpy(s->d_packed, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
        s->tex.packed = s->d_packed;
        s->tex.packed_bpr = bpr;
        // The lit path (k_lit): for the closures that are functions of the texel alone, when a frame has many more
        // pixels than the images have texels.  1 Mi texels cost 11 us (specular) / 8 us (normal map) per frame beside
        // the tile kernels; measured per frame, per-fragment -> lit: specular 4096^2 42.8 -> 34.5 us, x64 grid at 8192^2
        // 313 -> 223; normal map 4096^2 33.6 -> 32.7; at 2048^2 both lose (14.9 -> 16.3, 11.8 -> 13.6): from sixteen
        // pixels per texel.  (TR_LIT=0/1 overrides: tests run small frames through it.)
        if (fs == FS_NORMAL_MAP || fs == FS_SPECULAR) {
            const char *force = getenv("TR_LIT");
            const uint64_t texels = (uint64_t)tex[0].w * tex[0].h, pixels = (uint64_t)width * height;
            s->lit = force ? atoi(force) != 0 : pixels >= LIT_PIXELS_PER_TEXEL * texels;
            if (s->lit) {
                s->set_bpr = bpr;
                s->lit_bpr = (tex[0].w + 7u) / 8u;
                s->lit_words = s->lit_bpr * ((tex[0].h + 3u) / 4u) * 32u;
                for (int k = 0; k < LOOKAHEAD; k++)
                    if ((st = dev_alloc(&s->d_lit[k], (size_t)s->lit_words))) return st;
            }
        }
    }

    // bins
    for (tr_scene::BinState *b : { &s->bin_color, &s->bin_depth }) {
        const size_t nt = (b == &s->bin_color) ? s->n_tiles : s->n_tiles_full;
        for (int k = 0; k < SETS; k++) {
            // nt counters, then k_order's 8 bucket sizes and 8 cursors
            if ((st = dev_alloc(&b->count[k], nt + 16))) return st;
            HIP_TRY(hipMemset(b->count[k], 0, (nt + 16) * 4));
        }
    }
    for (int k = 0; k < LOOKAHEAD; k++)
        if ((st = dev_alloc(&s->d_order[k], (size_t)s->n_tiles_full * ORDER_LISTS))) return st;
    if ((st = dev_alloc(&s->d_bin_need, 1))) return st;
    // records in a pass's pool = all its (polygon, tile) pairs.  Automatic: twice an estimate from the frame and
    // the polygon count -- a model that fills half the frame with half of its polygons facing the viewer has
    // polygons with boxes of side s = sqrt(2 W H / n), each meeting (1 + s/128)(1 + s/16) tiles of 128x16: 5.0 pairs
    // per polygon for the reference's model at 4096^2 (measured 2.4), 1.32 for its 8x8 grid at 8192^2 (measured
    // 1.29) -- at least 65 536 records (6 MiB) and at most 16 Mi; a pass that wants more makes the pools grow (and is
    // rendered again)
    uint64_t cap = o.bin_capacity;
    if (!cap) {
        const double n = (double)(mesh->n_tri ? mesh->n_tri : 1u);
        const double side = sqrt(2.0 * (double)width * (double)height / n);
        const double pairs = 0.5 * (1.0 + side / (double)TILE_W) * (1.0 + side / (double)TILE_H) * n;
        cap = (uint64_t)(2.0 * pairs);
        cap = cap < 65536u ? 65536u : cap > (16u << 20) ? (16u << 20) : cap;
    }
    if (cap < 64) cap = 64;
    if (cap > 0x7FFFFFFFull) cap = 0x7FFFFFFFull;
    s->pool_cap = (uint32_t)cap;
    s->rec_pieces = (pipe == P_DARBOUX) ? REC_PIECES_LARGE : REC_PIECES_SMALL;
    for (int k = 0; k < LOOKAHEAD; k++) {
        if ((st = dev_alloc(&s->d_bins[k], (size_t)s->pool_cap * s->rec_pieces))) return st;
        if ((st = dev_alloc(&s->d_recs[k], (size_t)(mesh->n_tri ? mesh->n_tri : 1u) * s->rec_pieces))) return st;
    }
    HIP_TRY(hipStreamCreateWithFlags(&s->setup_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&s->setup_stream2, hipStreamNonBlocking));
    s->two_setup_streams = !(getenv("TR_SETUP_STREAMS") && atoi(getenv("TR_SETUP_STREAMS")) == 1);  // experiment hook
    for (int k = 0; k < RING; k++) {
        HIP_TRY(hipEventCreateWithFlags(&s->ev_setup[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_tile[k], hipEventDisableTiming));
    }
    HIP_TRY(hipMemset(s->d_bin_need, 0, 4));
    if ((st = dev_alloc(&s->d_overflow_seq, 1))) return st;
    HIP_TRY(hipMemset(s->d_overflow_seq, 0xFF, sizeof(unsigned long long)));

    // render targets; Buffer::new / Scene::new zero-fill them (shader.rs:46-47, scene.rs:71)
    s->slots.resize(1);
    {
        tr_scene::FrameSlot &fs = s->slots[0];
        if ((st = dev_alloc(&fs.z, npx))) return st;
        if ((st = dev_alloc(&fs.shadow, npx))) return st;
        HIP_TRY(hipMemset(fs.z, 0, npx * 4));
        HIP_TRY(hipMemset(fs.shadow, 0, npx * 4));
        if ((st = dev_alloc(&fs.zclean, (size_t)s->n_tiles))) return st;
        HIP_TRY(hipMemset(fs.zclean, 0, (size_t)s->n_tiles * 4));
        if ((st = dev_alloc(&fs.sclean, (size_t)s->n_tiles_full))) return st;
        HIP_TRY(hipMemset(fs.sclean, 0, (size_t)s->n_tiles_full * 4));
    }
    if (o.flags & TR_OPT_WINNER_TAP) {
        if ((st = dev_alloc(&s->d_winner, npx))) return st;
        HIP_TRY(hipMemset(s->d_winner, 0xFF, npx * 4));
    }
    if ((st = use_slot(s, 0, (uint8_t *)o.frame_buffer_device))) return st;
    if (s->d_fb == s->slots[0].fb)  // zero-filled just now, winner tap at "no fragment": every tile is clean
        HIP_TRY(hipMemsetAsync(s->d_fbclean, 0xFF, (size_t)s->n_tiles * 4, s->stream));
    if (o.flags & TR_OPT_TILE_STAMPS) {
        if ((st = dev_alloc(&s->d_stamps, (size_t)s->n_tiles_full * 8))) return st;
        HIP_TRY(hipMemset(s->d_stamps, 0, (size_t)s->n_tiles_full * 64));
    }
    if ((st = dev_alloc(&s->d_err, 1))) return st;
    HIP_TRY(hipMemset(s->d_err, 0, 4));
    {
        void *h = nullptr, *d = nullptr;
        HIP_TRY(hipHostMalloc(&h, 64, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
        s->h_alarm = (volatile uint32_t *)h;
        s->d_alarm = (uint32_t *)d;
        *s->h_alarm = 0u;
    }
    HIP_TRY(hipDeviceSynchronize());
    return TR_OK;
}

// Getters run in this order: (1) wait for the frame and take its status -- a bin overflow found
// here grows the bins and renders the frame again; (2) only then enqueue what derives from the
// frame (a pending clear, the fast-clear flags' f32::MIN, the u8 depth view); (3) wait, copy.
// Deriving first would copy out a view of the truncated frame.  The frame's status is returned
// after the copy (TR_E_OOB_LOOKUP / TR_E_BIN_OVERFLOW frames are still delivered); TR_E_HIP ends
// the call.
bool fatal(int st) { return st == TR_E_HIP || st == TR_E_NOMEM || st == TR_E_INVALID; }

int finish_read_back(tr_scene *s, int frame_status, void *dst, const void *src, size_t bytes)
{
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return frame_status;
}

int depth_view(tr_scene *s, int frame_status, const float *src, uint8_t *rgb)
{
    const size_t npx = (size_t)s->width * s->height;
    if (!s->d_view) {
        int st = dev_alloc(&s->d_view, npx * 3);
        if (st) return st;
    }
    int rc = launch_depth_view(src, s->d_view, s->width, s->height, s->stream);
    if (rc) return launch_status(rc, "k_depth_view");
    return finish_read_back(s, frame_status, rgb, s->d_view, npx * 3);
}

// Runs at the first use: does tr_powf (tr_powf.h: glibc's algorithm with the tables the BUILD
// host's libm held) return what the powf of the C library this process RUNS with returns?  The
// library may be built on one machine and loaded on another; the specular closure's
// "bit for bit the host's powf" only holds when the two agree.
int powf_matches_this_host()
{
    static int cached = -1;
    if (cached >= 0) return cached;
    if (!tr::specular_is_exact()) return cached = 0;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    int ok = 1;
    for (int i = 0; i < 20000 && ok; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        // the closure's domain: base max(r.z, 0) in [0, 1], exponent 0..255 (util.rs:82)
        const float base = (i % 97 == 0) ? 0.0f : (i % 89 == 0) ? 1.0f : (float)((x >> 11) & 0xFFFFFFu) / 16777216.0f;
        const float e = (float)((x >> 40) & 0xFFu);
        const float a = tr::tr_powf(base, e), b = powf(base, e);
        if (memcmp(&a, &b, 4) != 0) ok = 0;
    }
    return cached = ok;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int tr_abi_version(void) { return TR_ABI_VERSION; }

int tr_specular_exact(void) { return powf_matches_this_host(); }

const char *tr_last_error(void) { return tr::g_last_error.c_str(); }

int tr_selftest_device_math(int device, const float *x, const float *d, uint32_t n, uint32_t *out_u32,
                            int32_t *out_i32, uint32_t *out_u8, float *out_div, float *out_div_ref)
{
    if (!x || !d || !out_u32 || !out_i32 || !out_u8 || !out_div || !out_div_ref)
        return tr::fail(TR_E_INVALID, "null argument");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    void *buf[7] = {};
    int st = TR_OK;
    for (int i = 0; i < 7 && st == TR_OK; i++)
        if (hipMalloc(&buf[i], (size_t)(n ? n : 1) * 4) != hipSuccess) st = tr::fail(TR_E_HIP, "hipMalloc failed");
    if (st == TR_OK && (hipMemcpy(buf[0], x, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(buf[1], d, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess))
        st = tr::fail(TR_E_HIP, "upload failed");
    if (st == TR_OK) {
        int rc = launch_selftest((const float *)buf[0], (const float *)buf[1], n, (uint32_t *)buf[2], (int32_t *)buf[3],
                                 (uint32_t *)buf[4], (float *)buf[5], (float *)buf[6], nullptr);
        if (rc || hipDeviceSynchronize() != hipSuccess) st = tr::fail(TR_E_HIP, "self-test kernel failed");
    }
    void *outs[5] = { out_u32, out_i32, out_u8, out_div, out_div_ref };
    for (int i = 0; i < 5 && st == TR_OK; i++)
        if (hipMemcpy(outs[i], buf[2 + i], (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
            st = tr::fail(TR_E_HIP, "download failed");
    for (int i = 0; i < 7; i++)
        if (buf[i]) (void)hipFree(buf[i]);
    return st;
}

int tr_selftest_shadow_fetch(int device, uint32_t width, uint32_t height, const float *plain, const float *stale,
                             const uint32_t *sclean, uint32_t n, const float *x, const float *y, uint32_t *out_plain,
                             uint32_t *out_flagged, uint32_t *err_plain, uint32_t *err_flagged)
{
    if (!plain || !stale || !sclean || !x || !y || !out_plain || !out_flagged || !err_plain || !err_flagged || width == 0 ||
        height == 0)
        return tr::fail(TR_E_INVALID, "null argument");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    const size_t npx = (size_t)width * height;
    const size_t n_tiles = (size_t)((width + TILE_W - 1) / TILE_W) * ((height + TILE_H - 1) / TILE_H);
    const size_t bytes[9] = { npx * 4, npx * 4, n_tiles * 4, (size_t)n * 4, (size_t)n * 4, (size_t)n * 4, (size_t)n * 4,
                              (size_t)n * 4, (size_t)n * 4 };
    const void *src[5] = { plain, stale, sclean, x, y };
    void *buf[9] = {};
    int st = TR_OK;
    for (int i = 0; i < 9 && st == TR_OK; i++)
        if (hipMalloc(&buf[i], bytes[i] ? bytes[i] : 4) != hipSuccess) st = tr::fail(TR_E_HIP, "hipMalloc failed");
    for (int i = 0; i < 5 && st == TR_OK; i++)
        if (hipMemcpy(buf[i], src[i], bytes[i], hipMemcpyHostToDevice) != hipSuccess) st = tr::fail(TR_E_HIP, "upload failed");
    if (st == TR_OK) {
        int rc = launch_selftest_shadow((const float *)buf[0], (const float *)buf[1], (const uint32_t *)buf[2], width, height,
                                        (const float *)buf[3], (const float *)buf[4], n, (uint32_t *)buf[5], (uint32_t *)buf[6],
                                        (uint32_t *)buf[7], (uint32_t *)buf[8], nullptr);
        if (rc || hipDeviceSynchronize() != hipSuccess) st = tr::fail(TR_E_HIP, "self-test kernel failed");
    }
    void *outs[4] = { out_plain, out_flagged, err_plain, err_flagged };
    for (int i = 0; i < 4 && st == TR_OK; i++)
        if (hipMemcpy(outs[i], buf[5 + i], (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) st = tr::fail(TR_E_HIP, "download failed");
    for (int i = 0; i < 9; i++)
        if (buf[i]) (void)hipFree(buf[i]);
    return st;
}

int tr_selftest_device_unary(int device, int which, int exp_lo, int exp_hi, uint64_t *n_tested, uint64_t *n_bad,
                             uint32_t bad_bits[16])
{
    if (!n_tested || !n_bad || !bad_bits || which < 0 || which > 2 || exp_lo < -127 || exp_hi > 128 || exp_lo > exp_hi)
        return tr::fail(TR_E_INVALID, "bad argument");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    unsigned long long *d_n = nullptr;
    uint32_t *d_bits = nullptr;
    HIP_TRY(hipMalloc((void **)&d_n, 8));
    HIP_TRY(hipMalloc((void **)&d_bits, 64));
    HIP_TRY(hipMemset(d_n, 0, 8));
    HIP_TRY(hipMemset(d_bits, 0, 64));
    // every f32 with exponent in [exp_lo, exp_hi]: one contiguous range of bit patterns (positive
    // values; the reciprocal kernel checks -x beside x)
    const uint32_t first = (uint32_t)(exp_lo + 127) << 23;
    const uint64_t count = (uint64_t)(exp_hi - exp_lo + 1) << 23;
    int rc = launch_selftest_unary(which, first, count, d_n, d_bits, nullptr);
    int st = TR_OK;
    if (rc || hipDeviceSynchronize() != hipSuccess) st = tr::fail(TR_E_HIP, "self-test kernel failed");
    unsigned long long nb = 0;
    if (st == TR_OK && (hipMemcpy(&nb, d_n, 8, hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(bad_bits, d_bits, 64, hipMemcpyDeviceToHost) != hipSuccess))
        st = tr::fail(TR_E_HIP, "download failed");
    (void)hipFree(d_n);
    (void)hipFree(d_bits);
    *n_tested = count;
    *n_bad = nb;
    return st;
}

int tr_pipeline_count(void) { return P_COUNT; }

const char *tr_pipeline_name(int i) { return (i >= 0 && i < P_COUNT) ? kPipelines[i].name : nullptr; }

int tr_prepare_uniforms(int kind, tr_uniforms *u, uint32_t width, uint32_t height, const float light[3],
                        const float look_from[3], const float look_at[3], const float up[3])
{
    if (!u || !light || !look_from || !look_at || !up) return tr::fail(TR_E_INVALID, "null argument");
    return prepare_uniforms(kind, u, width, height, light, look_from, look_at, up);
}

int tr_scene_create(uint32_t width, uint32_t height, const tr_mesh *mesh, const tr_image_rgb8 tex[4],
                    const char *pipeline_name, const tr_options *opts, tr_scene **out)
{
    if (!out) return tr::fail(TR_E_INVALID, "null out pointer");
    *out = nullptr;
    tr_scene *s = new tr_scene();
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        s->id = ++g_scene_serial;
    }
    int st = create(width, height, mesh, tex, pipeline_name, opts, s);
    if (st != TR_OK) {
        std::string keep = tr::g_last_error;
        destroy(s);
        tr::g_last_error = keep;
        return st;
    }
    *out = s;
    return TR_OK;
}

void tr_scene_destroy(tr_scene *s) { destroy(s); }

int tr_scene_clear(tr_scene *s)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    s->z_fb_cleared = true;
    s->shadow_cleared = true;
    return TR_OK;
}

int tr_scene_set_light_direction(tr_scene *s, const float v[3])
{
    if (!s || !v) return tr::fail(TR_E_INVALID, "null argument");
    memcpy(s->light, v, sizeof s->light);
    return TR_OK;
}

int tr_scene_set_camera(tr_scene *s, const float look_from[3], const float look_at[3], const float up[3])
{
    if (!s || !look_from || !look_at || !up) return tr::fail(TR_E_INVALID, "null argument");
    memcpy(s->from, look_from, sizeof s->from);
    memcpy(s->at, look_at, sizeof s->at);
    memcpy(s->up, up, sizeof s->up);
    return TR_OK;
}

int tr_scene_render(tr_scene *s)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    if (s->broken) return tr::fail(TR_E_HIP, "the scene is unusable: a tile kernel could not be launched behind its chain");
    HIP_TRY(hipSetDevice(s->device));
    memcpy(s->last.light, s->light, 12);
    memcpy(s->last.from, s->from, 12);
    memcpy(s->last.at, s->at, 12);
    memcpy(s->last.up, s->up, 12);
    s->last.z_fb_cleared = s->z_fb_cleared;
    s->last.shadow_cleared = s->shadow_cleared;
    s->last.valid = true;
    s->last_was_group = false;
    if (!frame_is_groupable(s)) {
        // frames held back before it go first (it may render onto the last of them)
        int st = flush_deferred(s, false);
        if (st != TR_OK) return st;
        // a render without a clear depth-tests against the frame so far: its depth must be in memory
        if (!s->z_fb_cleared && (st = ensure_depth(s)) != TR_OK) return st;
        return render_frame(s);
    }
    // a cleared frame on the library's own stream: recorded now, rendered with its neighbours ("Automatic frame
    // groups").  Its constants are checked here, so that a camera the reference would panic on is this call's status.
    s->host_status = TR_OK;
    const PipelineDesc &pd = kPipelines[s->pipeline];
    for (int i = 0; i < pd.n_passes; i++) {
        DevUniforms du;
        int st = pass_uniforms(s, pd.pass[i], du);
        if (st != TR_OK) {
            s->host_status = st;
            return st;
        }
    }
    tr_scene::DeferredFrame f;
    memcpy(f.p.light, s->light, 12); memcpy(f.p.look_from, s->from, 12);
    memcpy(f.p.look_at, s->at, 12); memcpy(f.p.up, s->up, 12);
    f.fb = s->d_fb;
    s->deferred.push_back(f);
    s->z_fb_cleared = s->shadow_cleared = false;  // the frame has consumed the clear
    // (a loop that has filled sixteen groups in a row without anybody looking at a frame is a long run: its groups grow)
    const uint32_t target = s->auto_streak >= 16u ? long_run_group_size(s) : group_size(s);
    if (s->deferred.size() >= (size_t)target) {
        s->auto_streak++;
        return flush_deferred(s, true);
    }
    return TR_OK;
}

int tr_scene_render_frames(tr_scene *s, uint32_t n_frames, const tr_frame_params *frames, void *const *frame_buffers_device)
{
    if (!s || (n_frames && !frames)) return tr::fail(TR_E_INVALID, "null argument");
    if (s->broken) return tr::fail(TR_E_HIP, "the scene is unusable: a tile kernel could not be launched behind its chain");
    if (n_frames == 0) return TR_OK;
    if (frame_buffers_device)
        for (uint32_t i = 0; i < n_frames; i++)
            if (!frame_buffers_device[i]) return tr::fail(TR_E_INVALID, "null frame buffer in the list");
    HIP_TRY(hipSetDevice(s->device));
    return render_frames(s, n_frames, frames, frame_buffers_device);
}

int tr_scene_frames_per_launch(tr_scene *s) { return s ? (int)group_size(s) : tr::fail(TR_E_INVALID, "null scene"); }

int tr_scene_frames_kept(tr_scene *s)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    return s->last_was_group ? (int)s->tail.params.size() : 0;
}

int tr_scene_select_frame(tr_scene *s, uint32_t back)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    if (!s->last_was_group || back >= s->tail.params.size())
        return tr::fail(TR_E_INVALID, "tr_scene_select_frame: no such frame (tr_scene_frames_kept tells how many the last tr_scene_render_frames left)");
    HIP_TRY(hipSetDevice(s->device));
    const size_t k = s->tail.params.size() - 1u - back;
    const tr_frame_params &q = s->tail.params[k];
    memcpy(s->light, q.light, 12); memcpy(s->from, q.look_from, 12); memcpy(s->at, q.look_at, 12); memcpy(s->up, q.up, 12);
    return use_slot(s, s->tail.slot[k], s->tail.fbs.empty() ? nullptr : (uint8_t *)s->tail.fbs[k]);
}

int tr_scene_set_auto_group(tr_scene *s, int on)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    HIP_TRY(hipSetDevice(s->device));
    int st = submit_pending(s);  // frames held back so far are rendered as they were issued
    s->auto_group = on != 0;
    return st;
}

int tr_scene_flush(tr_scene *s)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    HIP_TRY(hipSetDevice(s->device));
    return submit_pending(s);
}

int tr_scene_sync(tr_scene *s)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    return sync_and_status(s);
}

void *tr_scene_frame_buffer_device(tr_scene *s) { return s ? s->d_fb : nullptr; }

int tr_scene_set_frame_buffer_device(tr_scene *s, void *frame_buffer_device)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    HIP_TRY(hipSetDevice(s->device));
    return use_slot(s, s->cur_slot, (uint8_t *)frame_buffer_device, true);
}

int tr_band_rows(uint32_t height, uint32_t n_ranks, uint32_t rank, uint32_t *row0, uint32_t *row1)
{
    if (!row0 || !row1 || n_ranks == 0 || rank >= n_ranks || height < n_ranks)
        return tr::fail(TR_E_INVALID, "tr_band_rows: need rank < n_ranks <= height");
    *row0 = (uint32_t)(((uint64_t)rank * height) / n_ranks);
    *row1 = (uint32_t)(((uint64_t)(rank + 1u) * height) / n_ranks);
    return TR_OK;
}

int tr_scene_set_stream(tr_scene *s, void *hip_stream)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    HIP_TRY(hipSetDevice(s->device));
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->own_stream) (void)hipStreamDestroy(s->stream);
    s->own_stream = false;
    if (hip_stream) {
        s->stream = (hipStream_t)hip_stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        s->own_stream = true;
    }
    return TR_OK;
}

int tr_scene_get_frame_buffer(tr_scene *s, uint8_t *rgb)
{
    if (!s || !rgb) return tr::fail(TR_E_INVALID, "null argument");
    int fst = sync_and_status(s);
    if (fatal(fst)) return fst;
    int st = flush_clear_color(s);
    if (st != TR_OK) return st;
    return finish_read_back(s, fst, rgb, s->d_fb, (size_t)s->width * s->height * 3);
}

int tr_scene_get_frame_buffer_async(tr_scene *s, uint8_t *rgb)
{
    if (!s || !rgb) return tr::fail(TR_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    int st = flush_clear_color(s);
    if (st != TR_OK) return st;
    st = submit_pending(s);
    if (st != TR_OK) return st;
    const size_t bytes = (size_t)s->width * s->height * 3;
    // same stream as the tile kernels: after the frame, before the next one overwrites it.  Into memory from
    // tr_host_alloc only the tiles that are not zeros on both sides travel (k_read_back); any other buffer gets
    // the whole frame from the copy engine.
    // The sparse form needs a scene that renders the WHOLE frame: k_read_back writes the tiles of the scene's band only,
    // and "the buffer ends up holding the complete frame" must hold for a band scene too (whose other rows, in a
    // caller's all-gather buffer, come from other ranks): such a scene takes the copy engine.
    const bool whole_frame = s->frame.band_y0 == 0 && s->frame.band_y1 == (int32_t)s->height;
    void *mapped = nullptr;
    uint64_t serial = 0, gen = 0;
    bool mine = false;  // this scene was the buffer's last writer
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        auto it = g_host_allocs.find(rgb);
        if (it != g_host_allocs.end()) {
            if (it->second.bytes >= bytes && whole_frame && s->width % 16u == 0u && s->d_fbclean) {
                mapped = it->second.device;
                serial = it->second.serial;
                mine = it->second.writer == s->id;
                gen = it->second.gen;
            }
            // (whichever way the frame travels: from now on this scene is the last writer, and every other scene's
            // record of the buffer has lapsed)
            it->second.writer = s->id;
            it->second.gen += 1u;
            if (mapped) gen = it->second.gen - 1u;
        }
    }
    if (mapped) {
        uint32_t *host_clean = nullptr;
        for (tr_scene::HostFlags &f : s->host_flags)
            if (f.host == rgb) {
                // the address of a buffer that has been freed (another buffer now), or somebody else -- another scene's
                // read-back, the caller (tr_scene_host_buffer_written) -- has written the buffer since: content unknown
                if (f.serial != serial || !mine || f.gen != gen) {
                    HIP_TRY(hipMemsetAsync(f.clean, 0, (size_t)s->n_tiles * 4, s->stream));
                    f.serial = serial;
                }
                f.gen = gen + 1u;
                host_clean = f.clean;
            }
        if (!host_clean) {
            if (s->host_flags.size() >= 64u) {  // (a caller that cycles through many buffers: forget the oldest)
                HIP_TRY(hipStreamSynchronize(s->stream));
                dev_free(s->host_flags.front().clean);
                s->host_flags.erase(s->host_flags.begin());
            }
            if ((st = dev_alloc(&host_clean, (size_t)s->n_tiles))) return st;
            HIP_TRY(hipMemsetAsync(host_clean, 0, (size_t)s->n_tiles * 4, s->stream));  // content of the buffer unknown
            s->host_flags.push_back({ rgb, serial, host_clean, gen + 1u });
        }
        int rc = launch_read_back(s->d_fb, (uint8_t *)mapped, s->d_fbclean, host_clean, s->frame, s->stream);
        if (rc) return launch_status(rc, "k_read_back");
    } else {
        HIP_TRY(hipMemcpyAsync(rgb, s->d_fb, bytes, hipMemcpyDeviceToHost, s->stream));
    }
    s->quiescent = false;
    // every pass issued so far is now in a consumer's hands: a bin overflow among them can no longer
    // be repaired by rendering again, and tr_scene_sync will say so (TR_E_BIN_OVERFLOW)
    s->observed_seq = s->pass_seq;
    return TR_OK;
}

int tr_scene_band_tiles(tr_scene *s, const void *frame_buffer_device, tr_band_tiles *out)
{
    if (!s || !out) return tr::fail(TR_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    int st = flush_clear_color(s);
    if (st != TR_OK) return st;
    st = submit_pending(s);
    if (st != TR_OK) return st;
    const uint8_t *fb = frame_buffer_device ? (const uint8_t *)frame_buffer_device : s->d_fb;
    const uint32_t *clean = nullptr;
    for (const tr_scene::FbFlags &f : s->fb_flags)
        if (f.fb == fb) clean = f.clean;
    if (!clean) return tr::fail(TR_E_INVALID, "tr_scene_band_tiles: the scene has not rendered into this frame buffer");
    out->frame_buffer_device = fb;
    out->clean_device = clean;
    out->width = s->frame.width;
    out->height = s->frame.height;
    out->tiles_x = s->frame.ntx;
    out->tiles_y = s->frame.nty;
    out->first_tile_row = s->frame.ty_base;
    out->band_y0 = s->frame.band_y0;
    out->band_y1 = s->frame.band_y1;
    // the flags now count as read by a consumer: a pass that overflowed its pool can no longer be repaired unseen
    s->observed_seq = s->pass_seq;
    s->quiescent = false;
    return TR_OK;
}

void *tr_host_alloc(size_t bytes)
{
    void *p = nullptr, *d = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocMapped) != hipSuccess) return nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) d = nullptr;
    if (d) {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        g_host_allocs[p] = { bytes, d, ++g_host_serial, 0u, 0u };
    }
    return p;
}

void tr_host_free(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        g_host_allocs.erase(p);
    }
    (void)hipHostFree(p);
}

int tr_scene_host_buffer_written(tr_scene *s, void *p)
{
    if (!s || !p) return tr::fail(TR_E_INVALID, "null argument");
    // nobody's record of the buffer holds any more (this scene's, and any other scene's that reads back into it)
    std::lock_guard<std::mutex> lock(g_host_mutex);
    auto it = g_host_allocs.find(p);
    if (it != g_host_allocs.end()) {
        it->second.writer = 0u;
        it->second.gen += 1u;
    }
    return TR_OK;
}

int tr_scene_get_z_buffer(tr_scene *s, uint8_t *rgb)
{
    if (!s || !rgb) return tr::fail(TR_E_INVALID, "null argument");
    int fst = sync_and_status(s);
    if (fatal(fst)) return fst;
    int st = s->z_fb_cleared ? TR_OK : ensure_depth(s);
    if (st == TR_OK) st = flush_clear_color(s);
    if (st == TR_OK) st = materialize_depth(s);
    if (st != TR_OK) return st;
    return depth_view(s, fst, s->d_z, rgb);
}

int tr_scene_get_shadow_buffer(tr_scene *s, uint8_t *rgb)
{
    if (!s || !rgb) return tr::fail(TR_E_INVALID, "null argument");
    int fst = sync_and_status(s);
    if (fatal(fst)) return fst;
    int st = flush_clear_shadow(s);
    if (st == TR_OK) st = materialize_shadow(s);
    if (st != TR_OK) return st;
    return depth_view(s, fst, s->d_shadow, rgb);
}

int tr_scene_read_z_f32(tr_scene *s, float *out)
{
    if (!s || !out) return tr::fail(TR_E_INVALID, "null argument");
    int fst = sync_and_status(s);
    if (fatal(fst)) return fst;
    int st = s->z_fb_cleared ? TR_OK : ensure_depth(s);
    if (st == TR_OK) st = flush_clear_color(s);
    if (st == TR_OK) st = materialize_depth(s);
    if (st != TR_OK) return st;
    return finish_read_back(s, fst, out, s->d_z, (size_t)s->width * s->height * 4);
}

int tr_scene_read_shadow_f32(tr_scene *s, float *out)
{
    if (!s || !out) return tr::fail(TR_E_INVALID, "null argument");
    int fst = sync_and_status(s);
    if (fatal(fst)) return fst;
    int st = flush_clear_shadow(s);
    if (st == TR_OK) st = materialize_shadow(s);
    if (st != TR_OK) return st;
    return finish_read_back(s, fst, out, s->d_shadow, (size_t)s->width * s->height * 4);
}

int tr_scene_read_winner_u32(tr_scene *s, uint32_t *out)
{
    if (!s || !out) return tr::fail(TR_E_INVALID, "null argument");
    if (!s->d_winner) return tr::fail(TR_E_INVALID, "scene was created without TR_OPT_WINNER_TAP");
    int fst = sync_and_status(s);
    if (fatal(fst)) return fst;
    int st = flush_clear_color(s);
    if (st != TR_OK) return st;
    return finish_read_back(s, fst, out, s->d_winner, (size_t)s->width * s->height * 4);
}

int tr_scene_debug_tile_stamps(tr_scene *s, uint64_t *out, uint32_t cap_tiles)
{
    if (!s || !out) return tr::fail(TR_E_INVALID, "null argument");
    if (!s->d_stamps) return tr::fail(TR_E_INVALID, "scene was created without TR_OPT_TILE_STAMPS");
    if (cap_tiles < s->n_tiles) return tr::fail(TR_E_INVALID, "buffer too small");
    HIP_TRY(hipSetDevice(s->device));
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(out, s->d_stamps, (size_t)s->n_tiles * 64, hipMemcpyDeviceToHost));
    return (int)s->n_tiles;
}

int tr_scene_profile_enable(tr_scene *s, int on)
{
    if (!s) return tr::fail(TR_E_INVALID, "null scene");
    HIP_TRY(hipSetDevice(s->device));
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    drain_events(s);
    s->profiling = on != 0;
    if (on) {
        memset(s->prof_ms, 0, sizeof s->prof_ms);
        memset(s->prof_n, 0, sizeof s->prof_n);
        memset(s->prof_frames, 0, sizeof s->prof_frames);
        s->frame_intervals_us.clear();
    }
    return TR_OK;
}

int tr_scene_profile_read(tr_scene *s, tr_kernel_time *out, int cap)
{
    if (!s || !out || cap <= 0) return tr::fail(TR_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(s->device));
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    drain_events(s);
    int n = 0;
    for (int k = 0; k < K_COUNT && n < cap; k++) {
        if (s->prof_n[k] == 0) continue;
        memset(&out[n], 0, sizeof out[n]);
        strncpy(out[n].name, kKernelNames[k], sizeof out[n].name - 1);
        out[n].launches = s->prof_n[k];
        out[n].total_ms = s->prof_ms[k];
        out[n].frames = s->prof_frames[k];
        n++;
    }
    return n;
}

int tr_scene_profile_frame_intervals(tr_scene *s, float *out_us, int cap)
{
    if (!s || !out_us || cap <= 0) return tr::fail(TR_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(s->device));
    {
        int sp = submit_pending(s);
        if (sp != TR_OK) return sp;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    drain_events(s);
    const int n = (int)s->frame_intervals_us.size() < cap ? (int)s->frame_intervals_us.size() : cap;
    memcpy(out_us, s->frame_intervals_us.data(), (size_t)n * sizeof(float));
    return n;
}

}  // extern "C"
