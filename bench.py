#!/usr/bin/env python3
"""bench.py -- headline benchmark of the triangle-fill path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one frame through the reference's per-frame protocol (app.rs:170,208-213):
clear -> set_light_direction -> set_camera -> render, with the finished frame left in HBM
(`get_frame_buffer`'s vertical flip is folded into the render's store address; no host
readback inside the timed region).

Workload at N=1: diablo.obj, -s phong, 4096x4096 -- the configuration the metric
"Mpixels/s shaded (z-test + Phong) at 4096x4096" is quoted on.  `value` =
N_shaded * K / t / 1e6 where N_shaded is the number of fragments the reference's serial loop
shades (z-accepts), counted by the CPU oracle on the same frame -- the GPU
shades only the survivors but is credited with the reference's count, never more.

N>1: the frame is sharded by screen rows over the ranks (one process per GPU) and the final
framebuffer is all-gathered over RCCL/xGMI every frame (north_star); total work is fixed, so
"scaling" is "strong".

The reference's assets are used when present ($TR_ASSETS, assets/_ref copied by
__graft_entry__.build(), or /root/reference/assets); otherwise a procedural sphere with the same
polygon count stands in and `config.workload` says so.
"""
import argparse
import json
import os
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)  # 80 ms at N=1: the 5-frame pipeline's fill and drain stay below 1 %
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--pipeline", default="phong")
    ap.add_argument("--model", default="diablo")
    ap.add_argument("--grid", type=int, default=1, help="n x n instancing (configs[4] uses 8)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    # TR_BENCH_FORCE_DIST=1 runs the N>1 code path (process group, band scene, in-place
    # all-gather) with a single rank: a rehearsal of the RCCL plumbing on a one-GPU box.
    use_dist = world > 1 or os.environ.get("TR_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

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
    # N > 1: the scene renders on torch's stream so that the all-gather is ordered behind it.
    # N = 1: the library's own stream (its frame pipelining hands tile kernels over in batches there).
    stream = torch.cuda.current_stream().cuda_stream if use_dist else None
    fb = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    band = None
    if use_dist:
        rows = [(r * H) // world for r in range(world + 1)]
        band = (rows[rank], rows[rank + 1])
        if len({rows[r + 1] - rows[r] for r in range(world)}) != 1:
            raise SystemExit("frame height must divide by the number of GPUs")
    scene = T.Scene(W, H, mesh, texs, pipe, device=local_rank, stream=stream,
                    frame_buffer_device=fb.data_ptr(), band_rows=band)
    chunk = None
    if use_dist:
        n = (band[1] - band[0]) * W * 3
        chunk = fb[rank * n:(rank + 1) * n]

    def step():
        scene.clear()
        scene.set_light_direction(lt)
        scene.set_camera(*cam)
        scene.render()
        if use_dist:
            dist.all_gather_into_tensor(fb, chunk)

    def barrier():
        if use_dist:
            dist.barrier()

    def device_idle():
        scene.flush()             # the scene may hold renders back to batch them: hand them over,
        torch.cuda.synchronize()  # then wait for every stream of the device

    for _ in range(args.warmup):
        step()
    device_idle()
    barrier()
    device_idle()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    device_idle()
    barrier()
    device_idle()
    elapsed = time.perf_counter() - t0
    status = scene.sync()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- metric 2: frames/s with the camera orbiting by 2*pi/200 per frame ---------
    # (deterministic stand-in for the reference's keyboard orbit, app.rs:173-200); not `value`.
    orbit_frames = 200

    def orbit_step(i):
        scene.clear()
        scene.set_light_direction(lt)
        scene.set_camera(*camera(2.0 * np.pi * i / orbit_frames))
        scene.render()
        if use_dist:
            dist.all_gather_into_tensor(fb, chunk)

    for i in range(orbit_frames):   # warm-up lap: lets the bins grow to what every angle needs
        orbit_step(i)
    device_idle()
    barrier()
    t1 = time.perf_counter()
    for i in range(orbit_frames):
        orbit_step(i)
    device_idle()
    barrier()
    orbit_elapsed = time.perf_counter() - t1
    orbit_status = scene.sync()
    step()  # back to the headline frame for the parity check below
    device_idle()

    # ---- per-kernel device time of the same step, HIP events on the scene's stream ------------
    scene.profile_enable(True)
    for _ in range(args.steps):
        step()
    prof = scene.profile_read()
    scene.profile_enable(False)

    out = None
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
        gpu_frame = fb.cpu().numpy().reshape(H, W, 3)
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
            # one launch = one pass over the frame (or over this rank's band of it)
            ach = bytes_alg / world / avg_s / 1e9
            traffic = None
            tf = os.path.join(REPO, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):
                try:
                    # HBM bytes of one k_tile launch from the rocprofv3 PMC passes of this workload
                    # (FETCH_SIZE x2 + WRITE_SIZE, see profiles/pmc_traffic.json); null when this
                    # workload has not been profiled
                    entry = json.load(open(tf)).get(workload, {})
                    # (profiled on one GPU: a band-sharded launch moves a different amount)
                    traffic = entry.get("hbm_bytes_per_launch") if entry.get("kernel") == dom and world == 1 else None
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                        "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_bytes_per_launch": bytes_alg // world,
                        "algorithmic_bytes_per_frame": sum(bytes_by_kernel.values())}
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "Mpixels/s shaded (z-test + Phong) at 4096x4096",
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
                       "sharding": "screen row bands + RCCL all-gather of the framebuffer" if use_dist else "none"},
            "frames_per_s": round(args.steps / elapsed, 1),
            "frames_per_s_orbit": round(orbit_frames / orbit_elapsed, 1) if orbit_status == 0 else None,
            "framebuffer_mpixels_per_s": round(W * H * args.steps / elapsed / 1e6, 1),
            "parity_vs_oracle": {"ok": parity_ok, "max_abs_rgb_diff": int(diff.max()), "tolerance": tol},
            "device_status": status,
            "kernel_us": {k: round(v["total_ms"] / max(v["launches"], 1) * 1e3, 2) for k, v in prof.items()},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    scene.close()
    if out is not None and not out["parity_vs_oracle"]["ok"]:
        raise SystemExit("GPU frame differs from the oracle")


if __name__ == "__main__":
    main()
