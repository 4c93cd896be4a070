// tr_plan.h -- the host's DECISIONS, free of HIP: how many frames a fused launch holds, in which groups a call of n frames
// goes out and into which frame slots, which targets the frames of an automatic group get, what to do about a pool that
// overflowed, when a pass's tile kernel is handed to the main stream.  tr_scene.cpp (the HIP layer) asks these functions
// and executes the answers; tests/test_planner.py drives them on the CPU (tests/emul binds them) with hypothesis:
// every frame rendered exactly once, kept frames in distinct slots, a replay never behind an observer's back, never
// more than BATCH + LOOKAHEAD passes between the host and the GPU.  Nothing here has a counterpart upstream (the
// reference renders one frame at a time on one thread, app.rs:166-247).
#pragma once

#include <stdint.h>

#include <vector>

namespace tr {
namespace plan {

constexpr int GROUP_MAX = 32;   // frames per fused launch, at most
constexpr int GROUP_SETS = 4;   // groups in flight: group g's setup reuses the sets of group g - GROUP_SETS
constexpr int LOOKAHEAD = 5;    // per-frame path: setup of pass p is ordered (on the device) after the tile kernel of pass p - LOOKAHEAD
constexpr int BATCH = LOOKAHEAD - 1;  // passes whose setup is queued and whose tile kernel the host still holds

// ---- the tile kernel's grid ---------------------------------------------------------------------------------------
// A pass's tiles are sorted into eight lists (k_order): seven by weight, the last the tiles without polygons.  The tile
// kernel takes one workgroup per WORK UNIT: a tile with polygons, or EMPTY_CHUNK entries of the empty list (a workgroup
// per empty tile was four waves and 20 KiB of LDS dispatched to read one flag).  A launch needs work_units(lengths)
// workgroups per frame; any larger grid is correct (the surplus exits at once) and one per tile always suffices, so a
// host that does not know the lengths asks for that -- and one that does (the chain has completed: k_bin_group reports
// them) asks for exactly the units of the frame that needs most.
constexpr uint32_t ORDER_LISTS_ = 8u;     // (= ORDER_BUCKETS, tr_kernels.hip)
constexpr uint32_t EMPTY_CHUNK = 32u;     // empty tiles per workgroup
inline uint32_t work_units(const uint32_t lengths[ORDER_LISTS_])
{
    uint32_t busy = 0u;
    for (uint32_t b = 0; b + 1u < ORDER_LISTS_; b++) busy += lengths[b];
    return busy + (lengths[ORDER_LISTS_ - 1u] + EMPTY_CHUNK - 1u) / EMPTY_CHUNK;
}
// The grid of a fused launch of `frames` frames whose list lengths are `lengths[f][8]`: the largest frame's units, or 0
// (= one workgroup per tile) when a frame's lengths do not add up to the pass's tiles -- not what a completed k_order
// leaves -- or when nothing would be saved.
inline uint32_t group_grid_units(const uint32_t *lengths, uint32_t frames, uint32_t n_tiles)
{
    uint32_t units = 0u;
    for (uint32_t f = 0; f < frames; f++) {
        uint32_t sum = 0u;
        for (uint32_t b = 0; b < ORDER_LISTS_; b++) sum += lengths[f * ORDER_LISTS_ + b];
        if (sum != n_tiles) return 0u;
        const uint32_t u = work_units(lengths + f * ORDER_LISTS_);
        units = u > units ? u : units;
    }
    return units >= n_tiles ? 0u : units;
}

// ---- frames per fused launch ---------------------------------------------------------------------------------------
struct SceneShape {
    uint32_t n_tiles;             // 128 x 16 tiles of the scene's band
    uint32_t frames_per_launch;   // tr_options.frames_per_launch (0 = automatic)
    uint32_t max_slots;           // tr_options.max_frame_slots (0 = automatic)
    uint32_t forced;              // TR_GROUP experiment hook (0 = none)
    bool winner_tap, tile_stamps; // single diagnostic buffers
    bool no_long_runs;            // the large groups' resources did not fit the device once
    uint32_t n_passes;
    uint64_t pool_bytes_per_pass; // pool_cap * record size
    uint64_t pixels;              // width * height
};

// Enough tiles to keep the machine full while one frame's light tiles drain and to make the gap between launches small
// against the launch -- 32 K tiles -- within GROUP_MAX, at least four frames; the sets of the groups in flight stay
// below 48 GiB of record pools; never more frames than slots.
inline uint32_t group_size(const SceneShape &s)
{
    if (s.winner_tap) return 1u;
    if (s.forced >= 1u && s.forced <= (uint32_t)GROUP_MAX) return s.max_slots && s.forced > s.max_slots ? s.max_slots : s.forced;
    if (s.frames_per_launch) return s.frames_per_launch;
    uint32_t g = s.n_tiles ? 32768u / s.n_tiles : (uint32_t)GROUP_MAX;
    g = g < 4u ? 4u : g > (uint32_t)GROUP_MAX ? (uint32_t)GROUP_MAX : g;
    const uint64_t per_frame = s.pool_bytes_per_pass * s.n_passes;
    while (g > 1u && (uint64_t)GROUP_SETS * g * per_frame > (48ull << 30)) g--;
    if (s.max_slots && g > s.max_slots) g = s.max_slots;
    return g;
}

// What a LONG run of frames grows to: one kernel-to-kernel gap per launch is the only lever on the 6-12 us between two
// tile kernels, so 32 frames per launch -- within 8 GiB of frame slots and 16 GiB of record pools, never below the usual
// group, only when the group size is automatic.
inline uint32_t long_run_group_size(const SceneShape &s)
{
    const uint32_t G = group_size(s);
    if (s.winner_tap || s.tile_stamps || s.frames_per_launch || s.forced || s.no_long_runs) return G;
    const uint64_t slot_bytes = s.pixels * (4ull * s.n_passes + 3ull);
    const uint64_t set_bytes = s.pool_bytes_per_pass * s.n_passes;
    uint32_t g = (uint32_t)GROUP_MAX;
    while (g > G && (g * slot_bytes > (8ull << 30) || (uint64_t)GROUP_SETS * g * set_bytes > (16ull << 30))) g /= 2u;
    if (s.max_slots && g > s.max_slots) g = s.max_slots;
    return g < G ? G : g;
}

// ---- the groups of one tr_scene_render_frames call -------------------------------------------------------------------
struct CallPlan {
    std::vector<uint32_t> sizes;  // frames per group, in issue order: they sum to n
    uint32_t slots = 0;           // frame i renders into slot i % slots
    uint32_t set_frames = 0;      // frames every group set must hold
    uint32_t kept = 0;            // the call's last `kept` frames exist afterwards (distinct slots)
};

// A LONG call (sixteen groups or more): the first group is the usual one (its tile kernel starts as early as it can),
// every later one up to `growth` times the one before -- its chain still hides behind the tile kernel in front -- up to
// the long-run size.  A SHORT call is mostly start-up and gaps: the usual group first, then up to twice the one before,
// up to `short_factor` times the usual size, the remainder in equal parts rather than a full group and a sliver.
inline CallPlan plan_call(uint32_t n, uint32_t G, uint32_t long_run, bool automatic, uint32_t growth, uint32_t short_factor)
{
    CallPlan p;
    if (n == 0u || G == 0u) return p;
    if (growth < 2u) growth = 2u;
    if (short_factor < 1u) short_factor = 1u;
    if (long_run < G) long_run = G;
    const uint32_t Gmax = n >= 16u * G ? long_run : G;
    uint32_t largest = 0;
    if (automatic && n > G && Gmax == G && short_factor > 1u) {
        uint32_t cap = short_factor * G < (uint32_t)GROUP_MAX ? short_factor * G : (uint32_t)GROUP_MAX;
        if (cap > long_run) cap = long_run;  // (slots and sets exist for that many)
        for (uint32_t left = n, g = 0; left; left -= g) {
            g = p.sizes.empty() ? G : (2u * g < cap ? 2u * g : cap);
            const uint32_t parts = (left + g - 1u) / g;
            g = (left + parts - 1u) / parts;
            p.sizes.push_back(g);
            largest = g > largest ? g : largest;
        }
    } else {
        for (uint32_t left = n, g = 0; left; left -= g) {
            g = p.sizes.empty() ? G : (growth * g < Gmax ? growth * g : Gmax);
            g = g < left ? g : left;
            p.sizes.push_back(g);
            largest = g > largest ? g : largest;
        }
    }
    // A long call's slots and sets are made for the largest group the policy can reach, not for this call's own (calls
    // of 100 and of 2 000 frames allocate the same); never fewer slots than the frames the call leaves behind.
    const uint32_t floor_ = n < G ? n : G;
    p.slots = Gmax > G ? Gmax : (largest > floor_ ? largest : floor_);
    p.set_frames = largest > Gmax ? largest : Gmax;
    p.kept = n < G ? n : G;
    return p;
}

// ---- the frames `render` has held back (automatic frame groups) -------------------------------------------------------
struct DeferredTarget {
    int slot;            // frame slot
    const void *fb;      // colour target: a caller's buffer, or null = the slot's own
    bool unreplayable;   // the frame can be seen in a caller's buffer but cannot be rendered again (only the last frame is)
};

// g frames recorded by the per-frame calls, frame j's colour target `fbs[j]` as it was current at its render():
// the last one goes to the scene's current targets, the ones before it -- which the per-frame protocol overwrites
// unobserved -- into other slots; a frame's colour goes where the caller pointed render() unless a later frame of the
// group goes there too (then nobody can see it: the slot's own buffer).  `cur_own_fb`: the current slot's own buffer;
// `is_own(fb)`: one of the scene's buffers.
template <typename IsOwn>
inline std::vector<DeferredTarget> plan_deferred(const std::vector<const void *> &fbs, int cur_slot, const void *cur_own_fb, IsOwn is_own)
{
    const size_t g = fbs.size();
    std::vector<DeferredTarget> out(g);
    int next_slot = 0;
    for (size_t j = 0; j < g; j++) {
        if (j + 1 == g) {
            out[j] = { cur_slot, fbs[j] == cur_own_fb ? nullptr : fbs[j], false };
            continue;
        }
        if (next_slot == cur_slot) next_slot++;
        bool overwritten = false;
        for (size_t k = j + 1; k < g; k++) overwritten = overwritten || fbs[k] == fbs[j];
        const void *fb = overwritten ? nullptr : fbs[j];
        out[j] = { next_slot++, fb, fb != nullptr && !is_own(fb) };
    }
    return out;
}

// ---- a pool that overflowed ----------------------------------------------------------------------------------------------
enum class OverflowAction {
    REPORT_CALLERS_BUFFER,  // the truncated frame sits in a caller's buffer the library cannot render into again
    REPORT_HANDED_ON,       // ... was handed to a consumer the library cannot call back (async read-back, caller's stream)
    REPLAY_TAIL,            // render the frames the last tr_scene_render_frames call left behind again
    REPLAY_LAST,            // render the last frame again from the cleared state it started in
    REPORT_ACCUMULATING     // the last render accumulated onto older content: it cannot be replayed
};

struct OverflowState {
    uint64_t first_bad_seq;     // first pass whose pool overflowed
    uint64_t observed_seq;      // passes below this were handed to a consumer
    uint64_t unreplayable_seq;  // passes below this rendered into callers' buffers that cannot be rendered again
    bool last_was_group;
    bool last_valid, last_started_cleared;  // the last per-frame render, and whether every target it wrote was cleared
};

inline OverflowAction overflow_action(const OverflowState &o)
{
    if (o.first_bad_seq < o.unreplayable_seq && o.first_bad_seq >= o.observed_seq) return OverflowAction::REPORT_CALLERS_BUFFER;
    if (o.first_bad_seq < o.observed_seq) return OverflowAction::REPORT_HANDED_ON;
    if (o.last_was_group) return OverflowAction::REPLAY_TAIL;
    if (o.last_valid && o.last_started_cleared) return OverflowAction::REPLAY_LAST;
    return OverflowAction::REPORT_ACCUMULATING;
}

// How far the pools grow: doubling until the hungriest pass fits, below 2^31 records.
inline uint64_t grown_pool(uint64_t cap, uint64_t need)
{
    if (cap == 0) cap = 1;
    while (cap < need && cap < 0x7FFFFFFFull) cap *= 2;
    return cap > 0x7FFFFFFFull ? 0x7FFFFFFFull : cap;
}

// ---- handing tile kernels to the main stream (per-frame path, the library's own stream) --------------------------------
// The tile kernel of pass p must run after that pass's setup (other stream).  A cross-stream wait packet between two tile
// kernels costs 5.5 us and is not needed when the setup has ALREADY completed when the tile kernel is enqueued: a pass
// stays pending on the host, and after every render one of these happens:
enum class Handover {
    HOST_WAITS,      // more than BATCH pending (the steady state): the HOST waits for the oldest one's setup, then submits
                     // it with no wait packet -- repeated until BATCH are left; also the frames-in-flight limit
    FRONT_WITH_WAIT, // the main stream has run dry: the oldest pending pass goes out behind a wait packet on its setup
    READY_ONLY       // whatever has its setup behind it goes out, without a wait packet
};
inline Handover handover(size_t pending, bool nothing_submitted_yet, bool newest_tile_done)
{
    if (pending > (size_t)BATCH) return Handover::HOST_WAITS;
    if (nothing_submitted_yet || newest_tile_done) return Handover::FRONT_WITH_WAIT;
    return Handover::READY_ONLY;
}

}  // namespace plan
}  // namespace tr
