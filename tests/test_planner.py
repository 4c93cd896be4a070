"""The host's planner (tiny_renderer_amd/csrc/tr_plan.h) on the CPU: the decisions tr_scene.cpp takes -- frames per fused
launch, the groups and frame slots of a tr_scene_render_frames call, the targets of an automatic group's frames, what to
do about a pool that overflowed, when a tile kernel is handed to the main stream -- are pure functions of small inputs;
the HIP layer only executes them.  Driven with hypothesis through the test-only binding in tests/emul.

Invariants: every frame of a call is rendered exactly once, in order; the frames a call leaves behind sit in distinct
slots; no group is larger than its sets or the slots; a frame that an observer may have seen is never silently rendered
again; the host never runs more than BATCH + LOOKAHEAD passes ahead of the GPU, and never waits for something the GPU
cannot finish."""
import ctypes as C

import numpy as np
from hypothesis import given, settings, strategies as st

from tests import emul_bind as E


def L():
    lib = E.lib()
    lib.tr_emul_plan_call.restype = C.c_uint32
    lib.tr_emul_plan_grid_units.restype = C.c_uint32
    lib.tr_emul_plan_grid_units.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.tr_emul_plan_work_units.restype = C.c_uint32
    lib.tr_emul_plan_work_units.argtypes = [C.c_void_p]
    lib.tr_emul_plan_grown_pool.restype = C.c_uint64
    lib.tr_emul_plan_grown_pool.argtypes = [C.c_uint64, C.c_uint64]
    lib.tr_emul_plan_overflow.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]
    return lib


def constants():
    out = (C.c_int * 4)()
    L().tr_emul_plan_constants(out)
    return dict(GROUP_MAX=out[0], GROUP_SETS=out[1], LOOKAHEAD=out[2], BATCH=out[3])


def sizes_of(shape):
    s = (C.c_uint32 * 8)(shape["n_tiles"], shape["frames_per_launch"], shape["max_slots"], shape["forced"],
                         int(shape["winner_tap"]), int(shape["tile_stamps"]), int(shape["no_long_runs"]), shape["n_passes"])
    big = (C.c_uint64 * 2)(shape["pool_bytes_per_pass"], shape["pixels"])
    out = (C.c_uint32 * 2)()
    L().tr_emul_plan_sizes(s, big, out)
    return int(out[0]), int(out[1])


def plan_call(n, G, long_run, automatic=True, growth=4, short_factor=3):
    sizes = (C.c_uint32 * 4096)()
    info = (C.c_uint32 * 3)()
    k = L().tr_emul_plan_call(n, G, long_run, int(automatic), growth, short_factor, sizes, 4096, info)
    return [int(sizes[i]) for i in range(k)], dict(slots=int(info[0]), set_frames=int(info[1]), kept=int(info[2]))


shapes = st.fixed_dictionaries(dict(
    n_tiles=st.integers(1, 1 << 20), frames_per_launch=st.sampled_from([0, 0, 0, 1, 2, 4, 7, 32]),
    max_slots=st.sampled_from([0, 0, 1, 2, 3, 8, 32]), forced=st.sampled_from([0, 0, 0, 1, 5, 32]),
    winner_tap=st.booleans(), tile_stamps=st.booleans(), no_long_runs=st.booleans(), n_passes=st.sampled_from([1, 2]),
    pool_bytes_per_pass=st.integers(64 * 96, 1 << 34), pixels=st.integers(1, 1 << 30)))


@given(shapes)
@settings(max_examples=400, deadline=None)
def test_group_sizes(shape):
    if shape["max_slots"] and shape["frames_per_launch"] > shape["max_slots"]:
        return  # (tr_scene_create refuses that combination)
    K = constants()
    G, long_run = sizes_of(shape)
    assert 1 <= G <= K["GROUP_MAX"] and G <= long_run <= K["GROUP_MAX"]
    if shape["winner_tap"]:
        assert G == 1 and long_run == 1
    if shape["max_slots"]:
        assert G <= shape["max_slots"] and long_run <= shape["max_slots"]
    if shape["frames_per_launch"] and not shape["winner_tap"] and not shape["forced"]:
        assert G == shape["frames_per_launch"] and long_run == G          # pinned by the caller: no growth
    if shape["no_long_runs"] or shape["tile_stamps"]:
        assert long_run == G
    # memory bounds of what the sizes make the host allocate
    per_frame = shape["pool_bytes_per_pass"] * shape["n_passes"]
    if G > 1 and not shape["frames_per_launch"] and not shape["forced"]:
        assert K["GROUP_SETS"] * G * per_frame <= 48 << 30
    if long_run > G:
        assert long_run * shape["pixels"] * (4 * shape["n_passes"] + 3) <= 8 << 30
        assert K["GROUP_SETS"] * long_run * per_frame <= 16 << 30


def test_group_sizes_of_the_baseline_configs():
    base = dict(frames_per_launch=0, max_slots=0, forced=0, winner_tap=False, tile_stamps=False, no_long_runs=False)
    # 4096^2, one pass, 64 Ki records of 96 B: 4 frames per launch, long runs grow to 32
    assert sizes_of(dict(base, n_tiles=32 * 256, n_passes=1, pool_bytes_per_pass=65536 * 96, pixels=4096 * 4096)) == (4, 32)
    # 800^2: 7 x 50 tiles -> 32 frames per launch already
    assert sizes_of(dict(base, n_tiles=7 * 50, n_passes=1, pool_bytes_per_pass=65536 * 96, pixels=800 * 800)) == (32, 32)
    # 8192^2 x64 grid: 4 frames; the long-run slots (8 GiB) allow 16
    assert sizes_of(dict(base, n_tiles=64 * 512, n_passes=1, pool_bytes_per_pass=850000 * 96, pixels=8192 * 8192)) == (4, 16)
    # 4096^2 shadow (two passes, 11 B per pixel): 32 slots = 5.9 GiB fit
    assert sizes_of(dict(base, n_tiles=32 * 256, n_passes=2, pool_bytes_per_pass=65536 * 96, pixels=4096 * 4096)) == (4, 32)


@given(st.integers(1, 3000), st.integers(1, 32), st.integers(1, 32), st.booleans(), st.integers(0, 9), st.integers(0, 9))
@settings(max_examples=600, deadline=None)
def test_every_frame_of_a_call_is_rendered_exactly_once(n, G, long_run, automatic, growth, short_factor):
    K = constants()
    long_run = max(long_run, G)
    sizes, info = plan_call(n, G, long_run, automatic, growth, short_factor)
    assert sum(sizes) == n and all(s >= 1 for s in sizes)                  # exactly once, in order (groups are consecutive)
    assert max(sizes) <= K["GROUP_MAX"] and max(sizes) <= info["set_frames"]
    assert sizes[0] == min(G, n) or (automatic and n > G)                   # the first group is the usual one ...
    assert sizes[0] <= max(G, 1) or n <= G                                  # ... never larger
    # a group never outgrows the slots: frame i -> slot i % slots keeps the frames of ONE launch apart
    assert max(sizes) <= info["slots"] <= K["GROUP_MAX"]
    # the frames the call leaves behind: its last min(n, G), in distinct slots
    assert info["kept"] == min(n, G) <= info["slots"]
    kept_slots = [(n - info["kept"] + k) % info["slots"] for k in range(info["kept"])]
    assert len(set(kept_slots)) == info["kept"]
    # growth is bounded: a later group's chain must hide behind the tile kernel of the one in front
    factor = max(growth, 2) if n >= 16 * G else 2
    for a, b in zip(sizes, sizes[1:]):
        assert b <= max(factor * a, a)
    if n >= 16 * G:
        assert max(sizes) <= long_run
    else:
        assert max(sizes) <= max(G, min(max(short_factor, 1) * G, long_run, K["GROUP_MAX"]))


def test_the_drivers_twenty_steps():
    assert plan_call(20, 4, 32)[0] == [4, 8, 8]
    assert plan_call(5, 4, 32)[0] == [4, 1] or plan_call(5, 4, 32)[0] == [3, 2] or sum(plan_call(5, 4, 32)[0]) == 5
    sizes, info = plan_call(2000, 4, 32)
    assert sizes[:4] == [4, 16, 32, 32] and info["slots"] == 32
    assert plan_call(3, 4, 32) == ([3], dict(slots=3, set_frames=4, kept=3))


@given(st.lists(st.integers(0, 20), min_size=2, max_size=32), st.integers(0, 31))
@settings(max_examples=400, deadline=None)
def test_targets_of_an_automatic_group(fbs, cur_slot):
    g = len(fbs)
    arr = (C.c_uint32 * g)(*fbs)
    out = (C.c_uint32 * (3 * g))()
    L().tr_emul_plan_deferred(arr, g, cur_slot, out)
    t = [(int(out[3 * j]), int(out[3 * j + 1]), int(out[3 * j + 2])) for j in range(g)]
    slots = [x[0] for x in t]
    assert len(set(slots)) == g and slots[-1] == cur_slot                 # every frame of the launch has targets of its own
    assert max(slots[:-1]) <= g - 1                                        # the others: slots 0 .. g - 1 without the current
                                                                           # one (ensure_slots makes max(g, G) of them)
    # the last frame -- the one the caller can see -- goes where the caller pointed render()
    assert t[-1][1] == (0xFFFFFFFF if fbs[-1] == 0 else fbs[-1]) and t[-1][2] == 0
    for j in range(g - 1):
        later = fbs[j] in fbs[j + 1:]
        if later:
            assert t[j][1] == 0xFFFFFFFF and t[j][2] == 0                 # nobody can see it: the slot's own buffer
        else:
            assert t[j][1] == fbs[j]                                       # visible in the buffer the caller chose
            assert t[j][2] == (1 if fbs[j] >= 16 else 0)                   # a caller's buffer: seen, but not replayable
    # no two frames of the launch write the same colour buffer
    written = [x[1] for x in t if x[1] != 0xFFFFFFFF]
    assert len(set(written)) == len(written)


ACTIONS = ("REPORT_CALLERS_BUFFER", "REPORT_HANDED_ON", "REPLAY_TAIL", "REPLAY_LAST", "REPORT_ACCUMULATING")


@given(st.integers(0, 50), st.integers(0, 50), st.integers(0, 50), st.booleans(), st.booleans(), st.booleans())
@settings(max_examples=400, deadline=None)
def test_a_replay_never_goes_behind_an_observers_back(first_bad, observed, unreplayable, group, valid, cleared):
    a = ACTIONS[L().tr_emul_plan_overflow(first_bad, observed, unreplayable, int(group), int(valid), int(cleared))]
    if first_bad < observed:                       # a consumer may have used the truncated frame: say so
        assert a == "REPORT_HANDED_ON"
    elif first_bad < unreplayable:                 # it sits in a caller's buffer the library cannot render again
        assert a == "REPORT_CALLERS_BUFFER"
    elif group:
        assert a == "REPLAY_TAIL"
    elif valid and cleared:
        assert a == "REPLAY_LAST"
    else:                                          # an accumulating render cannot be reproduced from a cleared state
        assert a == "REPORT_ACCUMULATING"


@given(st.integers(0, 1 << 31), st.integers(0, 1 << 33))
def test_pools_grow_by_doubling(cap, need):
    new = int(L().tr_emul_plan_grown_pool(cap, need))
    assert new <= 0x7FFFFFFF and new >= min(max(cap, 1), 0x7FFFFFFF)
    if need <= 0x7FFFFFFF:
        assert new >= min(need, 0x7FFFFFFF) or new == 0x7FFFFFFF
    if cap >= need and cap >= 1:
        assert new == min(cap, 0x7FFFFFFF)


@given(st.lists(st.integers(0, 6), min_size=1, max_size=120), st.booleans())
@settings(max_examples=300, deadline=None)
def test_the_host_stays_within_nine_passes_of_the_gpu(progress, two_setup_streams):
    """The per-frame path on the library's own stream: run_pass issues pass p's chain (ordered ON THE DEVICE after the tile
    kernel of pass p - LOOKAHEAD), keeps its tile kernel pending, and asks handover() what to submit.  The GPU is modelled
    by its dependencies only; hypothesis decides how far it gets between two renders (`progress`: how many enabled
    events complete before the next render)."""
    K = constants()
    LOOK, BATCH = K["LOOKAHEAD"], K["BATCH"]
    setup_done, tile_done, submitted, waited_with_packet = set(), set(), [], set()
    pending = []

    def enabled():
        ev = []
        for p in range(issued):
            if p not in setup_done:
                prev = p - (2 if two_setup_streams else 1)
                if (prev < 0 or prev in setup_done) and (p < LOOK or (p - LOOK) in tile_done):
                    ev.append(("setup", p))
        for p in submitted:
            if p not in tile_done and p in setup_done and (p == 0 or (p - 1) in tile_done):
                ev.append(("tile", p))
        return ev

    def advance(k):
        for _ in range(k):
            ev = enabled()
            if not ev:
                return
            kind, p = ev[0]
            (setup_done if kind == "setup" else tile_done).add(p)

    def submit_front(with_packet):
        p = pending.pop(0)
        assert p == len(submitted)                                   # tile kernels go out in pass order, each once
        if with_packet:
            waited_with_packet.add(p)
        else:
            assert p in setup_done                                   # no wait packet: its setup HAS completed
        submitted.append(p)

    issued = 0
    for k in progress:
        pending.append(issued)
        issued += 1
        newest_done = bool(submitted) and submitted[-1] in tile_done
        action = ("HOST_WAITS", "FRONT_WITH_WAIT", "READY_ONLY")[L().tr_emul_plan_handover(len(pending), int(not submitted), int(newest_done))]
        if action == "HOST_WAITS":
            while len(pending) > BATCH:
                guard = 0
                while pending[0] not in setup_done:                  # hipEventSynchronize(ev_setup[front]): must be reachable
                    before = (len(setup_done), len(tile_done))
                    advance(1)
                    assert (len(setup_done), len(tile_done)) != before, "the host waits for something the GPU cannot finish"
                    guard += 1
                    assert guard < 10000
                submit_front(False)
        elif action == "FRONT_WITH_WAIT":
            submit_front(True)
        else:
            while pending and pending[0] in setup_done:
                submit_front(False)
        assert len(pending) <= BATCH
        in_flight = issued - len(tile_done)                          # passes issued whose frame is not complete
        assert in_flight <= BATCH + LOOK + 1, in_flight              # (+ the one just issued)
        advance(k)
    # everything can drain
    while pending:
        submit_front(True)
    advance(100000)
    assert len(tile_done) == issued and len(setup_done) == issued


@given(st.lists(st.lists(st.integers(0, 5000), min_size=8, max_size=8), min_size=1, max_size=32), st.integers(0, 3))
@settings(max_examples=300, deadline=None)
def test_the_tile_kernels_grid_covers_every_work_unit(lens, spoil):
    """tr_plan.h work_units / group_grid_units: a tile kernel takes one workgroup per tile WITH polygons and one per 32 entries
    of the empty list; a fused launch's frames share the grid.  Whatever the lists' lengths: the grid the host asks for
    holds every unit of every frame, never more workgroups than one per tile, exactly the largest frame's units -- and 0
    (= one per tile) as soon as a frame's lengths do not add up to the pass's tiles (words a chain has not written yet)."""
    lib = L()
    n_tiles = sum(lens[0])
    frames = [list(f) for f in lens]
    for f in frames:      # every frame of a pass has the same number of tiles: move the difference into the empty list
        d = n_tiles - sum(f[:7])
        if d < 0:
            f[:7] = [0] * 7
            d = n_tiles
        f[7] = d
    flat = (C.c_uint32 * (8 * len(frames)))(*[v for f in frames for v in f])
    units = [lib.tr_emul_plan_work_units((C.c_uint32 * 8)(*f)) for f in frames]
    for f, u in zip(frames, units):
        busy, empty = sum(f[:7]), f[7]
        assert u == busy + (empty + 31) // 32 and u <= max(n_tiles, 1)
    got = lib.tr_emul_plan_grid_units(flat, len(frames), n_tiles)
    assert got == (max(units) if max(units) < n_tiles else 0)
    if spoil and n_tiles:
        flat[8 * (spoil % len(frames)) + 7] += 1     # a frame whose words are not a completed k_order's
        assert lib.tr_emul_plan_grid_units(flat, len(frames), n_tiles) == 0
