#!/usr/bin/env python3
"""bench.py -- headline benchmark of the triangle-fill path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one frame of the reference's per-frame protocol (app.rs:170,208-213):
clear -> set_light_direction -> set_camera -> render, with the finished frame left in HBM
(`get_frame_buffer`'s vertical flip is folded into the render's store address; no host
readback inside the timed region; the readback-inclusive rate is reported beside it).
At N=1 the K steps are submitted through tr_scene_render_frames (the library's throughput path: the
same K frames, each into render targets of its own, but `config.frames_per_launch` of them rendered
by one launch of each kernel); `--submit frame` issues the four calls per frame instead, and the
default run reports that rate too (`per_frame_protocol`).

Workload at N=1: diablo.obj, -s phong, 4096x4096 -- the configuration the metric
"Mpixels/s shaded (z-test + Phong) at 4096x4096" is quoted on.  `value` =
N_shaded * K / t / 1e6 where N_shaded is the number of fragments the reference's serial loop
shades (z-accepts), counted by the CPU oracle on the same frame -- the GPU
shades only the survivors but is credited with the reference's count, never more.

N>1: one process per GPU.  `python bench.py --gpus N` without a launcher starts the N rank
processes itself (torch.distributed.run, before anything in this process touches a GPU) and
relays rank 0's line; under torchrun it joins the group it finds and refuses to run if the
group's size is not N.  The frame is sharded by screen rows (tr_band_rows) over the ranks and the
final framebuffer is all-gathered over RCCL/xGMI every frame (north_star); frames are
double-buffered so that the gather of frame f runs on a second stream under the render of
frame f+1.  Total work is fixed, so "scaling" is "strong".

The reference's assets are used when present ($TR_ASSETS, assets/_ref copied by
__graft_entry__.build(), or /root/reference/assets); otherwise a procedural sphere with the same
polygon count stands in and `config.workload` says so.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def find_assets(name):
    for root in (os.environ.get("TR_ASSETS"), os.path.join(REPO, "assets", "_ref"), "/root/reference/assets"):
        if root and os.path.isfile(os.path.join(root, name, "model.obj")):
            return os.path.join(root, name)
    return None


def camera(angle):
    a = np.float32(angle)
    return ([float(np.sin(a)), 0.0, float(np.cos(a))], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0])


def light(angle):
    a = np.float32(angle)
    return [float(np.sin(a)), 0.0, float(np.cos(a))]


TEXEL_BYTES = {"default": 3, "phong": 3, "shadow": 3, "normal_map": 6, "darboux": 6, "specular": 7, "occlusion": 0}


def algorithmic_bytes(W, H, pipe, stats):
    """bytes_alg = W*H*C + F_cov*4 + F_acc*(7+S) + T_kept*132 (+ shadow terms),
    split by the kernel that owns the bytes: {"k_tile": colour pass, "k_tile_depth": depth pass}."""
    two_pass = pipe in ("shadow", "occlusion")
    color = stats[1] if two_pass else stats[0]
    S = TEXEL_BYTES[pipe]
    out = {"k_tile": int(W * H * 7 + color["frag_covered"] * 4 + color["frag_accept"] * (7 + S)
                         + color["tri_kept"] * 132)}
    if two_pass:
        out["k_tile"] += int(color["frag_accept"] * 4)          # one shadow-buffer gather per accept
        out["k_tile_depth"] = int(W * H * 4 + stats[0]["frag_covered"] * 4 + stats[0]["shadow_upd"] * 4
                                  + stats[0]["tri_kept"] * 132)  # clear + read + update of the shadow buffer
    return out


def source_fingerprint():
    """sha256 over the sources the device code is built from.  profiles/pmc_traffic.json records the
    fingerprint of the build that was profiled (scripts/summarise_profiles.py); a counter value
    measured on other code is not reported (the GPU box has no .git to ask for a commit)."""
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "tiny_renderer_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N ranks, one per GPU, from this
    process -- which has not imported torch, let alone touched a GPU -- and pass rank 0's JSON on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def pmc_traffic(workload, kernel, frames_in_launch, world=1):
    """HBM bytes of one launch of `kernel` for `workload` from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    separate passes: profiles/pmc_traffic.json).  Only reported when the counters were collected on THIS source
    (fingerprint match) and on one GPU.  Returns (bytes or None, note, entry)."""
    traffic, note, entry = None, "not profiled", {}
    tf = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if not os.path.exists(tf):
        return traffic, note, entry
    try:
        db = json.load(open(tf))
        entry = db.get("workloads", {}).get(workload, {})
        if world != 1:
            note = "profiled on one GPU only"
        elif not entry:
            note = "workload not profiled"
        elif db.get("source_fingerprint") != source_fingerprint():
            note = "profiles/pmc_traffic.json was collected on other source (%s)" % db.get("source_fingerprint")
        elif kernel in entry.get("per_kernel_hbm_bytes_per_launch", {}) or entry.get("kernel") == kernel:
            traffic = entry.get("per_kernel_hbm_bytes_per_launch", {}).get(kernel, entry.get("hbm_bytes_per_launch"))
            note = "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, %s" % entry.get("source", "profiles/")
            profiled_frames = entry.get("frames_per_launch", 1)
            if abs(profiled_frames - frames_in_launch) > 0.05 * frames_in_launch:
                # (a long run's launches hold more frames than the profiled loop's: the bytes are per
                # frame -- every frame has targets of its own -- and are scaled to this run's launch)
                traffic = int(round(traffic / profiled_frames * frames_in_launch))
                note += "; measured per launch of %s frames, scaled to %.2f" % (profiled_frames, frames_in_launch)
    except Exception as e:  # a malformed file must not take the bench down
        traffic, note = None, "unreadable pmc_traffic.json: %s" % e
    return traffic, note, entry


# BASELINE.json `configs`, as bench flags: (label, model, pipeline, size, grid, steps)
BASELINE_CONFIGS = [
    ("configs[0]", "african_head", "default", 800, 1, 400),
    ("configs[1]", "diablo", "phong", 2048, 1, 400),
    ("configs[2]", "diablo", "darboux", 4096, 1, 200),
    ("configs[3]", "diablo", "shadow", 4096, 1, 200),
    ("configs[4]", "diablo", "specular", 8192, 8, 64),
]


def measure_config(T, O, torch, label, model, pipe, size, grid, steps, device_index):
    """One BASELINE config on one GPU (SURVEY.md 8d "Report"): `steps` frames of the reference's first-frame state
    through tr_scene_render_frames, timed like the headline (frames resident in HBM, device idle on both sides), the
    last frame checked against the CPU oracle's, the tile kernel's roofline from the library's own profile of the
    same frames.  configs[4] is the whole 8192^2 frame on ONE GPU here; its 8-way sharded form is the N > 1 run's
    `scale_config`."""
    adir = find_assets(model)
    if adir:
        mesh, texs = T.load_assets(adir)
        wl_model = "%s.obj" % model
    else:
        mesh, texs = T.synthetic_scene()
        wl_model = "synthetic-sphere-5022"
    if grid > 1:
        mesh = T.instanced_grid(mesh, grid)
        wl_model += " x%d grid" % (grid * grid)
    W = H = size
    workload = "%s, -s %s, %dx%d" % (wl_model, pipe, W, H)
    cam, lt = camera(0.0), light(0.0)
    scene = T.Scene(W, H, mesh, texs, pipe, device=device_index)
    params = np.zeros((steps, 12), np.float32)
    params[:, 0:3] = lt
    params[:, 3:6], params[:, 6:9], params[:, 9:12] = cam

    def idle():
        scene.flush()
        try:
            scene.sync()
        except T.TinyRendererError:
            pass
        torch.cuda.synchronize()

    for attempt in range(3):   # warm-up (again if it is what made the pools grow)
        scene.render_frames(params[:max(steps // 4, 8)])
        idle()
        try:
            scene.sync()
            break
        except T.TinyRendererError as e:
            if e.code != -9:
                raise
    idle()
    t0 = time.perf_counter()
    scene.render_frames(params)
    idle()
    elapsed = time.perf_counter() - t0
    status = scene.sync()
    scene.profile_enable(True)
    scene.render_frames(params)
    idle()
    prof = scene.profile_read()
    scene.profile_enable(False)
    gpu_frame = scene.get_frame_buffer()
    scene.close()

    cpu = O.Scene(W, H, mesh, texs, pipe)
    cpu.clear()
    cpu.set_light_direction(lt)
    cpu.set_camera(*cam)
    assert cpu.render() == 0
    stats = cpu.stats()
    color = stats[1] if pipe in ("shadow", "occlusion") else stats[0]
    n_shaded = color["frag_accept"]
    diff = np.abs(gpu_frame.astype(np.int16) - cpu.get_frame_buffer().astype(np.int16))
    tol = 1 if pipe == "specular" and not T.load_library().tr_specular_exact() else 0
    cpu.close()

    bytes_by_kernel = algorithmic_bytes(W, H, pipe, stats)
    dom = max(bytes_by_kernel, key=lambda k: prof.get(k, {}).get("total_ms", 0.0))
    tile = prof.get(dom)
    roofline = None
    if tile and tile["launches"]:
        avg_s = tile["total_ms"] / tile["launches"] / 1e3
        frames_in_launch = tile["frames"] / tile["launches"]
        ach = bytes_by_kernel[dom] * frames_in_launch / avg_s / 1e9
        traffic, note, _ = pmc_traffic(workload, dom, frames_in_launch)
        roofline = {"kernel": dom, "frac": round(ach / HBM_PEAK_GBPS, 4), "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "physical_frac": round(traffic / avg_s / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
                    "traffic": traffic, "traffic_source": note,
                    "avg_launch_us_per_frame": round(avg_s * 1e6 / frames_in_launch, 2),
                    "algorithmic_bytes_per_frame": sum(bytes_by_kernel.values())}
    return {"config": label, "workload": workload,
            "metric": "Mpixels/s shaded (z-test + %s) at %dx%d" % (pipe, W, H),
            "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 5),
            "value": round(n_shaded * steps / elapsed / 1e6, 2), "unit": "Mpixels/s",
            "n_shaded_per_frame": n_shaded, "frames_per_s": round(steps / elapsed, 1),
            "kernel_us_per_frame": {k: round(v["total_ms"] / max(v["frames"], 1) * 1e3, 2) for k, v in prof.items()},
            "roofline": roofline, "parity_ok": bool(diff.max() <= tol), "max_abs_rgb_diff": int(diff.max()),
            "device_status": status}


XGMI_LINK_GBPS = 76.8   # one xGMI link, one direction (7 links x ~153 GB/s bidirectional per GPU): the estimate's peak


def measure_scale_config(T, torch, dist, args, exchange_kind, rank, world, device_index):
    """BASELINE.json configs[4] -- "diablo x64 instanced grid, -s specular, 8192x8192, screen-tile shard across the GPUs
    with the framebuffer all-gather" -- sharded over the ranks of THIS run: every rank renders its row
    band of a group of frames per kernel launch and the bands are exchanged frame by frame on a second stream
    (ShardedScene.render_frames).  Collective: every rank calls it; rank 0 returns the report (per-rank device times,
    bytes exchanged, the assembled frame against the oracle's), the others None."""
    from tiny_renderer_amd.sharded import ShardedScene
    adir = find_assets(args.model)
    if adir:
        mesh, texs = T.load_assets(adir)
        wl_model = "%s.obj" % args.model
    else:
        mesh, texs = T.synthetic_scene()
        wl_model = "synthetic-sphere-5022"
    g = args.scale_grid
    if g > 1:
        mesh = T.instanced_grid(mesh, g)
        wl_model += " x%d grid" % (g * g)
    W = H = args.scale_size
    pipe = "specular"
    workload = "%s, -s %s, %dx%d" % (wl_model, pipe, W, H)
    cam, lt = camera(0.0), light(0.0)
    steps = max(args.scale_steps, 1)
    params = np.zeros((steps, 12), np.float32)
    params[:, 0:3] = lt
    params[:, 3:6], params[:, 6:9], params[:, 9:12] = cam
    s = ShardedScene(W, H, mesh, texs, pipe, device=device_index, exchange=exchange_kind)

    def idle():
        s.scene.flush()
        torch.cuda.synchronize()

    s.render_frames(params[:max(steps // 4, 4)])   # warm-up (a pool that has to grow is repaired by sync, on all ranks)
    idle()
    s.sync()
    dist.barrier()
    sent0 = s.exchange_bytes_sent()
    t0 = time.perf_counter()
    s.render_frames(params)
    idle()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    status = s.sync()
    sent = s.exchange_bytes_sent()
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    s.scene.profile_enable(True)
    s.enable_timing(True)
    s.render_frames(params)
    idle()
    prof = s.scene.profile_read()
    s.scene.profile_enable(False)
    mine = dict({"rank": rank, "band_rows": list(s.band)}, **s.timings())
    mine["k_tile_us"] = round(prof["k_tile"]["total_ms"] / max(prof["k_tile"]["frames"], 1) * 1e3, 2) if "k_tile" in prof else None
    s.sync()
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    band_bytes = (s.band[1] - s.band[0]) * W * 3
    report = None
    if rank == 0:
        from oracle import oracle as O
        frame = s.last_frame_tensor().cpu().numpy().reshape(H, W, 3)
        cpu = O.Scene(W, H, mesh, texs, pipe)
        cpu.clear()
        cpu.set_light_direction(lt)
        cpu.set_camera(*cam)
        assert cpu.render() == 0
        n_shaded = cpu.stats()[0]["frag_accept"]
        diff = np.abs(frame.astype(np.int16) - cpu.get_frame_buffer().astype(np.int16))
        tol = 1 if not T.load_library().tr_specular_exact() else 0
        cpu.close()
        report = {"workload": workload, "n_gpus": world, "steps": steps, "exchange": exchange_kind,
                  "metric": "Mpixels/s shaded (z-test + %s) at %dx%d" % (pipe, W, H),
                  "ms_per_step": round(elapsed / steps * 1e3, 5), "value": round(n_shaded * steps / elapsed / 1e6, 2),
                  "unit": "Mpixels/s", "frames_per_s": round(steps / elapsed, 1), "scaling": "strong",
                  "frames_per_launch": s.scene.frames_per_launch, "per_rank": per_rank,
                  "exchange_bytes_per_frame": int((sent - sent0) // steps) if sent is not None else int(band_bytes * (world - 1)),
                  "exchange_dense_bytes_per_frame": int(band_bytes * (world - 1)),
                  # what the exchange alone costs a frame if every band crosses one xGMI link at its peak: all n - 1
                  # copies of a band at once over separate links (direct), or handed on link by link (ring)
                  "exchange_bound_estimate_us": {"direct": round(band_bytes / (XGMI_LINK_GBPS * 1e3), 1),
                                                 "ring": round(band_bytes * (world - 1) / (XGMI_LINK_GBPS * 1e3), 1),
                                                 "link_GBps": XGMI_LINK_GBPS,
                                                 "note": "band bytes / link peak; ranks that share a GPU exchange through its memory, not xGMI"},
                  "parity_vs_oracle": {"ok": bool(diff.max() <= tol), "max_abs_rgb_diff": int(diff.max()), "tolerance": tol},
                  "device_status": status}
    dist.barrier()   # (the others wait while rank 0 compares: nobody tears its slots down under a peer)
    s.close()
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)  # 70 ms at N=1: the frame pipeline's fill and drain stay below 1 %
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--pipeline", default="phong")
    ap.add_argument("--model", default="diablo")
    ap.add_argument("--grid", type=int, default=1, help="n x n instancing (configs[4] uses 8)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the orbit / latency / read-back legs (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config report of BASELINE.json's five configs")
    ap.add_argument("--exchange", choices=("rccl", "librccl", "peer"), default=os.environ.get("TR_BENCH_EXCHANGE", "rccl"),
                    help="N>1: how the bands travel: rccl = torch.distributed all_gather_into_tensor (RCCL, the default); "
                         "librccl = the library's own RCCL communicator (tr_exchange_*, no torch in the data path); peer = the "
                         "library's peer transport (bands pulled out of IPC-mapped slots by the DMA engines; also runs with "
                         "several ranks on ONE GPU: TR_BENCH_SHARE_GPU=1)")
    ap.add_argument("--sparse", action="store_true",
                    help="N>1 with --exchange peer: send the band tile by tile, skipping tiles that are the cleared colour on "
                         "both sides (tr_exchange_all_gather_tiles)")
    ap.add_argument("--no-scale", action="store_true", help="N>1: skip the scale_config leg (BASELINE configs[4] sharded N ways)")
    ap.add_argument("--scale-size", type=int, default=8192, help="scale_config: frame size (BASELINE configs[4]: 8192)")
    ap.add_argument("--scale-grid", type=int, default=8, help="scale_config: n x n instances of the model (configs[4]: 8)")
    ap.add_argument("--scale-steps", type=int, default=32, help="scale_config: frames in its timed call")
    ap.add_argument("--submit", choices=("frames", "frame"), default="frames",
                    help="N=1: frames = tr_scene_render_frames (groups of frames per launch), frame = four calls per frame")
    ap.add_argument("--frames-per-launch", type=int, default=0, help="tr_options.frames_per_launch (0 = automatic)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        self_launch(args, sys.argv[1:])  # does not return
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d: refusing to label a %d-rank run as %d GPUs"
                         % (args.gpus, world, world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    shared_gpu = world > torch.cuda.device_count()
    if shared_gpu and os.environ.get("TR_BENCH_SHARE_GPU") != "1":
        raise SystemExit("%d ranks but %d GPUs visible" % (world, torch.cuda.device_count()))
    if shared_gpu and args.exchange != "peer":
        # (RCCL refuses a communicator with two ranks on one device -- "Duplicate GPU detected" from deep inside
        # init_process_group; say it here, in one sentence)
        raise SystemExit("bench.py: %d ranks on %d GPU(s): RCCL needs a device per rank -- rehearse several ranks on one GPU "
                         "with --exchange peer (the library's peer transport), or give every rank a GPU"
                         % (world, torch.cuda.device_count()))
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    # TR_BENCH_FORCE_DIST=1 runs the N>1 code path (process group, band scene, double-buffered
    # gather on a second stream) with a single rank: a rehearsal of the plumbing on a one-GPU box.
    use_dist = world > 1 or os.environ.get("TR_BENCH_FORCE_DIST") == "1"
    group_ranks = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = "nccl" if args.exchange == "rccl" else "gloo"  # the library's exchanges only need a rendezvous
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")
        group_ranks = dist.get_world_size()
        if group_ranks != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus says %d" % (group_ranks, args.gpus))
    # ShardedScene's name for the transport
    exchange_kind = {"rccl": "torch", "librccl": "rccl", "peer": "peer-sparse" if args.sparse else "peer"}[args.exchange]

    import tiny_renderer_amd as T

    W = H = args.size
    pipe = args.pipeline
    adir = find_assets(args.model)
    if adir:
        mesh, texs = T.load_assets(adir)
        data = "%s.obj + TGA maps from the reference's assets" % args.model
        wl_model = "%s.obj" % args.model
    else:
        mesh, texs = T.synthetic_scene()
        data = "synthetic (procedural sphere, 5022 polygons, 1024^2 maps; reference assets not on this box)"
        wl_model = "synthetic-sphere-5022"
    if args.grid > 1:
        mesh = T.instanced_grid(mesh, args.grid)
        wl_model += " x%d grid" % (args.grid * args.grid)
    workload = "%s, -s %s, %dx%d" % (wl_model, pipe, W, H)

    cam, lt = camera(0.0), light(0.0)  # the state of the reference's first frame (app.rs:158-159)

    # ---- the scene on this rank -------------------------------------------------------------
    # N = 1: the library's own stream (its frame pipelining hands tile kernels over in batches there)
    #        and its own frame slots.
    # N > 1: a ShardedScene (tiny_renderer_amd/sharded.py): this rank's band scene on a real torch side stream
    #        (never the null stream: tr_options.stream = NULL means a library-owned stream, and a collective on
    #        torch's current stream would not be ordered behind it), double-buffered frame slots, the exchange
    #        of frame f on a second stream under the render of frame f + 1 -- whichever transport moves the bands.
    torch.cuda.synchronize()
    free_before_scene = torch.cuda.mem_get_info(device_index)[0]
    grouped = args.submit == "frames"
    sharded = None
    band, band_bytes = None, 0
    if use_dist:
        from tiny_renderer_amd.sharded import ShardedScene
        sharded = ShardedScene(W, H, mesh, texs, pipe, device=device_index, exchange=exchange_kind,
                               frames_per_launch=args.frames_per_launch)
        scene = sharded.scene
        band = sharded.band
        band_bytes = (band[1] - band[0]) * W * 3
    else:
        scene = T.Scene(W, H, mesh, texs, pipe, device=device_index, frames_per_launch=args.frames_per_launch)
    frames_per_launch = scene.frames_per_launch if grouped else 1

    def read_frame():
        if use_dist:
            return sharded.last_frame_tensor().cpu().numpy().reshape(H, W, 3)
        return scene.get_frame_buffer()

    driver = sharded if use_dist else scene

    def step(cam_now=cam):
        driver.clear()
        driver.set_light_direction(lt)
        driver.set_camera(*cam_now)
        driver.render()

    def frame_params(cams):
        out = np.zeros((len(cams), 12), np.float32)
        out[:, 0:3] = lt
        for i, c in enumerate(cams):
            out[i, 3:6], out[i, 6:9], out[i, 9:12] = c
        return out

    headline_params = frame_params([cam] * max(args.steps, args.warmup, 1))

    def run(k, cams=None, per_frame=False):
        """k steps.  Grouped submission: ONE render_frames call for all of them."""
        if grouped and not per_frame:
            driver.render_frames(headline_params[:k] if cams is None else frame_params(cams))
            return
        for i in range(k):
            step(cam if cams is None else cams[i])

    def barrier():
        if use_dist:
            dist.barrier()

    def device_idle():
        scene.flush()             # the scene may hold renders back to batch them: hand them over,
        if not use_dist:
            # ... wait for the scene's own stream first: hipStreamSynchronize wakes the thread 20-25 us sooner after a
            # long wait than the device-wide hipDeviceSynchronize behind torch.cuda.synchronize (which then returns at
            # once; scripts/probe_driver.py).  A frame status is taken by clean_sync(), not here.
            try:
                scene.sync()
            except T.TinyRendererError:
                pass
        torch.cuda.synchronize()  # then wait for every stream of the device


    def clean_sync():
        """sync on every rank; True when no rank's triangle bins overflowed (the library has grown them by
        then).  Taken together: a rank that repeated a loop on its own would issue more collectives than its peers."""
        if use_dist:
            # (collective, and repairs an overflow itself by rendering the last frame / group again on every rank)
            return True, sharded.sync()
        try:
            return True, scene.sync()
        except T.TinyRendererError as e:
            if e.code != -9:
                raise
            return False, e.code

    for attempt in range(4):   # warm-up; again if it is what made the bins grow
        run(args.warmup)
        device_idle()
        ok, _ = clean_sync()
        if ok:
            break
    barrier()
    device_idle()
    sent_before = sharded.exchange_bytes_sent() if use_dist else None
    t0 = time.perf_counter()
    run(args.steps)
    device_idle()   # this rank's frames are complete (flush + torch.cuda.synchronize()) ...
    barrier()       # ... and so are everybody's: the clock stops at the slowest rank (max over ranks below)
    elapsed = time.perf_counter() - t0
    # what this rank really pushed to its peers per frame of the timed loop (the library's count; dense: the band to each)
    sent_per_frame = (sharded.exchange_bytes_sent() - sent_before) // max(args.steps, 1) if sent_before is not None else None
    ok, status = clean_sync()
    if not ok:
        raise SystemExit("triangle bins overflowed inside the timed region: the timing is void")
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    extras = not args.no_extras
    # ---- metric 2: frames/s with the camera orbiting by 2*pi/200 per frame ---------
    # (deterministic stand-in for the reference's keyboard orbit, app.rs:173-200); not `value`.
    orbit_frames, orbit_elapsed, orbit_status = 200, None, None
    if extras:
        orbit_cams = [camera(2.0 * np.pi * i / orbit_frames) for i in range(orbit_frames)]
        for lap in range(4):   # warm-up laps: the bins grow to what every angle needs (a frame whose bins
            run(orbit_frames, orbit_cams)   # overflowed on a caller's stream is reported, not silently repaired)
            device_idle()
            ok, _ = clean_sync()
            if ok:
                break
        barrier()
        t1 = time.perf_counter()
        run(orbit_frames, orbit_cams)
        device_idle()
        barrier()
        orbit_elapsed = time.perf_counter() - t1
        ok, orbit_status = clean_sync()

    # ---- the same steps through the reference's own per-frame protocol (four calls per frame) ---------
    # (a) as the library runs it by default: frames nobody reads in between are held back and fused;
    # (b) with that switched off (TR_OPT_NO_AUTO_GROUP): one launch of each kernel per frame -- round 1's path
    per_frame_elapsed, per_frame_steps, unfused_elapsed = None, min(args.steps, 500), None
    if extras and grouped and not use_dist:
        run(50, per_frame=True)
        device_idle()
        t1 = time.perf_counter()
        run(per_frame_steps, per_frame=True)
        device_idle()
        per_frame_elapsed = time.perf_counter() - t1
        scene.sync()
        scene.set_auto_group(False)   # (the same scene: a second one would share hardware queues with it)
        run(50, per_frame=True)
        device_idle()
        t1 = time.perf_counter()
        run(per_frame_steps, per_frame=True)
        device_idle()
        unfused_elapsed = time.perf_counter() - t1
        scene.sync()
        scene.set_auto_group(True)

    # ---- single-frame latency: clear -> render -> sync with nothing else in flight -----------------
    latency_us = None
    if extras:
        lat = []
        for _ in range(30):
            device_idle()
            t1 = time.perf_counter()
            step()
            driver.sync()
            lat.append((time.perf_counter() - t1) * 1e6)
        lat.sort()
        latency_us = {"median": round(lat[len(lat) // 2], 1), "min": round(lat[0], 1)}

    # ---- read-back inclusive rate: every frame copied to page-locked host memory (the reference hands
    # every frame to its window, app.rs:213-218); copies queue behind their frames, one sync at the end
    readback = None
    if extras and not use_dist:
        pinned = [scene.pinned_frame() for _ in range(2)]
        n_rb = 40
        for k in range(4):
            step()
            scene.get_frame_buffer_async(pinned[k % 2])
        scene.sync()
        t1 = time.perf_counter()
        for k in range(n_rb):
            step()
            scene.get_frame_buffer_async(pinned[k % 2])
        scene.sync()
        readback = (time.perf_counter() - t1) / n_rb

    # ---- per-kernel device time of the same steps, HIP events on the kernels' own dispatches ----------
    # (ends on the headline frame: the parity check below reads it)
    scene.profile_enable(True)
    if use_dist:
        sharded.enable_timing(True)
    run(min(args.steps, 400) if use_dist else args.steps)
    device_idle()
    prof = scene.profile_read()
    intervals = np.sort(scene.profile_frame_intervals())
    scene.profile_enable(False)

    per_rank = None
    if use_dist:
        mine = dict({"rank": rank, "band_rows": list(band)}, **sharded.timings())
        mine["k_tile_us"] = round(prof["k_tile"]["total_ms"] / max(prof["k_tile"]["frames"], 1) * 1e3, 2) if "k_tile" in prof else None
        sharded.enable_timing(False)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    out = None
    closed_scene = False
    if rank == 0:
        from oracle import oracle as O

        # N_shaded and the algorithmic bytes come from the CPU oracle on the same frame; the
        # same run checks the GPU frame (whole frame: all bands gathered) bit for bit.
        cpu = O.Scene(W, H, mesh, texs, pipe)
        cpu.clear()
        cpu.set_light_direction(lt)
        cpu.set_camera(*cam)
        t_cpu0 = time.perf_counter()
        assert cpu.render() == 0
        t_one = time.perf_counter() - t_cpu0
        stats = cpu.stats()
        color = stats[1] if pipe in ("shadow", "occlusion") else stats[0]
        n_shaded = color["frag_accept"]
        gpu_frame = read_frame()
        ref_frame = cpu.get_frame_buffer()
        diff = np.abs(gpu_frame.astype(np.int16) - ref_frame.astype(np.int16))
        # specular calls powf: exact when the library reproduces the host libm's (tr_specular_exact), else 1 LSB
        tol = 1 if pipe == "specular" and not T.load_library().tr_specular_exact() else 0
        parity_ok = bool(diff.max() <= tol)

        cpu_baseline = None
        if not args.no_cpu:
            frames, spent = 1, t_one
            while spent < args.cpu_seconds and frames < 1000:
                cpu.clear()
                cpu.set_light_direction(lt)
                cpu.set_camera(*cam)
                t1 = time.perf_counter()
                cpu.render()
                spent += time.perf_counter() - t1
                frames += 1
            # clear() is part of the frame; time it separately (same 1-thread loop, scene.rs:128-137)
            t1 = time.perf_counter()
            cpu.clear()
            t_clear = time.perf_counter() - t1
            per_frame = spent / frames + t_clear
            try:
                model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except Exception:
                model = "unknown"
            cpu_baseline = {"value": round(n_shaded / per_frame / 1e6, 3), "unit": "Mpixels/s shaded",
                            "cores": 1, "kind": "port", "host_cpu": model, "host_cores_available": os.cpu_count(),
                            "sample": "%d frames of the same workload (clear+render), CPU oracle "
                                      "(C restatement of the reference, single-threaded like it), "
                                      "%.3f s/frame" % (frames, per_frame)}

        bytes_by_kernel = algorithmic_bytes(W, H, pipe, stats)
        # the dominant kernel = the one with the larger share of device time
        dom = max(bytes_by_kernel, key=lambda k: prof.get(k, {}).get("total_ms", 0.0))
        bytes_alg = bytes_by_kernel[dom]
        tile = prof.get(dom)
        roofline = None
        if tile and tile["launches"]:
            avg_s = tile["total_ms"] / tile["launches"] / 1e3
            # one launch = one pass over the frames of a group (N=1: frames_per_launch of them; the profiled
            # loop's launches / frames are counted by the library) or over this rank's band of one frame
            frames_in_launch = tile["frames"] / tile["launches"]
            bytes_launch = bytes_alg * frames_in_launch / world
            ach = bytes_launch / avg_s / 1e9
            traffic, traffic_note, entry = pmc_traffic(workload, dom, frames_in_launch, world)
            physical = round(traffic / avg_s / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None
            # the kernel's real limiter: what crosses HBM is a fraction of the algorithmic bytes (fast-clear and
            # colour-clean flags), the busy tiles are bound by vector-instruction issue.  issue_frac = share of the
            # launch's SIMD-cycles in which a vector instruction executes (SQ counters of this source, else null).
            issue = entry.get("issue") if (traffic is not None and isinstance(entry, dict)) else None
            roofline = {"bound": "valu_issue" if (physical is not None and physical < 0.3) else "hbm",
                        "issue_frac": issue.get("issue_frac") if issue else None,
                        "issue_source": issue.get("source") if issue else "SQ counters not collected on this source / workload",
                        "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                        "traffic_source": traffic_note,
                        # what actually crosses the HBM interface (the algorithmic model also counts the
                        # z clear of empty tiles, which the fast-clear flags never write)
                        "physical_frac": physical,
                        "limiter": "vector/scalar instruction issue of the tiles with polygons, not HBM "
                                   "(SQ counters under profiles/)",
                        # `frac` is SURVEY 8d's ALGORITHMIC bytes over the launch's duration: an effective-bandwidth figure.
                        # The kernel does not move those bytes -- the clear and the z of empty tiles are flags, a cleared
                        # frame's depth stays in LDS -- so it can exceed 1; `physical_frac` is what crosses HBM.
                        "frac_is": "algorithmic bytes / duration / peak (effective; the kernel moves `traffic` bytes: physical_frac)",
                        "avg_launch_us": round(avg_s * 1e6, 2), "frames_per_launch": round(frames_in_launch, 3),
                        "avg_launch_us_per_frame": round(avg_s * 1e6 / frames_in_launch, 2),
                        "algorithmic_bytes_per_launch": int(bytes_launch),
                        "algorithmic_bytes_per_frame": sum(bytes_by_kernel.values())}
        ms = elapsed / args.steps * 1e3
        t_frame = None
        if len(intervals):
            t_frame = {"median": round(float(intervals[len(intervals) // 2]), 2),
                       "p10": round(float(intervals[len(intervals) // 10]), 2),
                       "p90": round(float(intervals[(len(intervals) * 9) // 10]), 2),
                       "frames": int(len(intervals)),
                       "what": "completion-to-completion of consecutive frames' tile kernels (HIP events, profiled loop)"}
        out = {
            # (BASELINE.json's metric string for the workload it is quoted on; any other --size / --pipeline says what it ran)
            "metric": "Mpixels/s shaded (z-test + Phong) at 4096x4096" if (pipe == "phong" and W == 4096 and H == 4096)
            else "Mpixels/s shaded (z-test + %s) at %dx%d" % (pipe, W, H),
            "value": round(n_shaded * args.steps / elapsed / 1e6, 2),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": data,
            "config": {"workload": workload, "n_shaded_per_frame": n_shaded,
                       "polygons": int(mesh["idx"].shape[0]),
                       "submission": ("tr_scene_render_frames: %d frames per launch of each kernel (the later groups of a call "
                                      "of sixteen groups or more grow to 32), every frame into render targets of its own"
                                      % frames_per_launch) if grouped
                       else "per frame: clear, set_light_direction, set_camera, render",
                       "frames_per_launch": frames_per_launch,
                       "sharding": ("screen row bands (tr_band_rows) + %s of the framebuffer, double-buffered: exchange of frame f "
                                    "under the render of f+1" % {
                                        "torch": "RCCL all-gather (torch.distributed)",
                                        "rccl": "RCCL all-gather (the library's own communicator, tr_exchange)",
                                        "peer": "peer-to-peer band copies (tr_exchange: bands pulled by the DMA engines)",
                                        "peer-sparse": "peer-to-peer sparse tile push (tr_exchange_all_gather_tiles)"}[exchange_kind])
                       if use_dist else "none"},
            "group_ranks": group_ranks if use_dist else 1,
            # what rank 0 ships per frame: its band to each of the others (--exchange peer: the library's count of the timed
            # loop -- with --sparse only the tiles that are not the cleared colour on both sides)
            "exchange_bytes_per_frame": (int(sent_per_frame) if sent_per_frame is not None else int(band_bytes * (world - 1))) if use_dist else 0,
            "exchange_dense_bytes_per_frame": int(band_bytes * (world - 1)) if use_dist else 0,
            "scaling_measured": "unmeasured here: one GPU per box (SCALE is the driver's 8-GPU run)" if world == 1 else "this run",
            "frames_per_s": round(args.steps / elapsed, 1),
            "frames_per_s_orbit": round(orbit_frames / orbit_elapsed, 1) if orbit_elapsed and orbit_status == 0 else None,
            "framebuffer_mpixels_per_s": round(W * H * args.steps / elapsed / 1e6, 1),
            "per_frame_protocol": {"ms_per_step": round(per_frame_elapsed / per_frame_steps * 1e3, 5),
                                   "value": round(n_shaded * per_frame_steps / per_frame_elapsed / 1e6, 2),
                                   "steps": per_frame_steps,
                                   "what": "the same frames through clear / set_light_direction / set_camera / render "
                                           "(frames nobody reads in between are held back and fused by the library)",
                                   "unfused_ms_per_step": round(unfused_elapsed / per_frame_steps * 1e3, 5),
                                   "unfused_value": round(n_shaded * per_frame_steps / unfused_elapsed / 1e6, 2),
                                   "unfused_what": "TR_OPT_NO_AUTO_GROUP: one launch of each kernel per frame"}
            if per_frame_elapsed else None,
            "t_frame_us": t_frame,
            "latency_us": latency_us,
            # everything the scene holds on the device once all the loops above have run (record pools, work lists,
            # render targets of every frame in flight, textures): free memory before the scene minus free memory now
            "device_memory_mb": round((free_before_scene - torch.cuda.mem_get_info(device_index)[0]) / 1e6, 1),
            "readback_inclusive_mpixels_per_s": round(n_shaded / readback / 1e6, 1) if readback else None,
            "readback_inclusive_frame_us": round(readback * 1e6, 1) if readback else None,
            "parity_vs_oracle": {"ok": parity_ok, "max_abs_rgb_diff": int(diff.max()), "tolerance": tol},
            "device_status": status,
            "kernel_us": {k: round(v["total_ms"] / max(v["launches"], 1) * 1e3, 2) for k, v in prof.items()},
            "kernel_us_per_frame": {k: round(v["total_ms"] / max(v["frames"], 1) * 1e3, 2) for k, v in prof.items()},
            "per_rank": per_rank,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        # ---- SURVEY.md 8d "Report": every BASELINE config that fits one GPU, in the same run (default flags only) ----
        default_run = (world == 1 and not use_dist and extras and not args.no_configs and pipe == "phong" and W == 4096
                       and args.model == "diablo" and args.grid == 1)
        if default_run:
            scene.close()   # (its slots and pools: 5 GB the 8192^2 config should not have to share the device with)
            closed_scene = True
            reports = []
            for label, model, cpipe, csize, cgrid, csteps in BASELINE_CONFIGS:
                try:
                    reports.append(measure_config(T, O, torch, label, model, cpipe, csize, cgrid, csteps, device_index))
                except Exception as e:   # one config must not take the headline down
                    reports.append({"config": label, "error": "%s: %s" % (type(e).__name__, e)})
            out["configs"] = reports
    if not closed_scene:
        driver.close()
    scale = None
    # (TR_BENCH_FORCE_SCALE=1 with TR_BENCH_FORCE_DIST=1: the leg with a single rank, a rehearsal of the RCCL transports'
    # plumbing on a one-GPU box, where RCCL cannot have a second rank)
    if use_dist and not args.no_scale and (world > 1 or os.environ.get("TR_BENCH_FORCE_SCALE") == "1"):
        scale = measure_scale_config(T, torch, dist, args, exchange_kind, rank, world, device_index)
    if out is not None:
        if scale is not None:
            out["scale_config"] = scale
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None and not out["parity_vs_oracle"]["ok"]:
        raise SystemExit("GPU frame differs from the oracle")
    if out is not None and out.get("scale_config") and not out["scale_config"]["parity_vs_oracle"]["ok"]:
        raise SystemExit("scale_config: the assembled frame differs from the oracle")
    if out is not None and any(not c.get("parity_ok", False) for c in out.get("configs", [])):
        raise SystemExit("a BASELINE config's GPU frame differs from the oracle (or the config failed): %s"
                         % [c.get("config") for c in out["configs"] if not c.get("parity_ok", False)])


if __name__ == "__main__":
    main()
