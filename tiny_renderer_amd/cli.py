"""Headless counterpart of the reference's CLI (src/main.rs:9-39: `-p <asset dir>`, `-s <pipeline>`).

The reference opens a window and orbits with the keyboard; here the camera / light angles are
explicit and the last frame is written as PNG, TGA or binary PPM.  Rendering happens on the GPU
through the C ABI.  `--gpus N` shards every frame by screen rows over N GPUs of the node (one
process per GPU, started from here; RCCL all-gather of the frame buffer, sharded.py).
"""
import argparse
import os
import sys
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(prog="tiny_renderer_amd", description=__doc__)
    ap.add_argument("-p", dest="asset_path", default="assets/diablo", help="asset folder (main.rs:12)")
    ap.add_argument("-s", dest="pipeline", default="default", help="shader pipeline (main.rs:13)")
    ap.add_argument("--width", type=int, default=800)    # main.rs:6
    ap.add_argument("--height", type=int, default=800)   # main.rs:7
    ap.add_argument("--camera-angle", type=float, default=0.0, help="app.rs:158, radians")
    ap.add_argument("--light-angle", type=float, default=0.0, help="app.rs:159, radians")
    ap.add_argument("--frames", type=int, default=1, help="frames to render (camera orbits 2*pi over them)")
    ap.add_argument("--seconds", type=float, default=0.0,
                    help="run the reference's time-based frame loop for this long instead (app.rs:166-247): the "
                         "camera turns at CAMERA_SPEED = 3 rad/s as if a key were held, `FPS --- n` every second")
    ap.add_argument("--no-readback", action="store_true",
                    help="with --seconds: leave the frames on the GPU (the reference hands every frame to its window)")
    ap.add_argument("--out", default=None, help="write the last frame: .png, .tga (24-bit) or binary PPM otherwise")
    ap.add_argument("--view", choices=("frame", "z", "shadow"), default="frame")  # app.rs:213-215
    ap.add_argument("--device", type=int, default=-1)
    ap.add_argument("--synthetic", action="store_true", help="procedural scene instead of -p")
    ap.add_argument("--gpus", type=int, default=1, help="shard every frame by screen rows over this many GPUs")
    ap.add_argument("--exchange", choices=("torch", "rccl", "peer", "peer-sparse"), default="torch",
                    help="--gpus N: how the bands travel (ShardedScene): torch = RCCL all-gather through torch.distributed, rccl = "
                         "the library's own RCCL communicator, peer / peer-sparse = the library's peer transport (these two also "
                         "run with several ranks on one GPU)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    # --gpus N > 1 outside a launcher: start the N ranks from here, before anything touches a GPU
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        from .sharded import launch_ranks
        return launch_ranks(args.gpus, "tiny_renderer_amd.cli", list(argv) if argv is not None else sys.argv[1:])
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    sharded = world > 1 or os.environ.get("TR_CLI_FORCE_DIST") == "1"
    rank = int(os.environ.get("RANK", "0"))
    say = print if rank == 0 else (lambda *a, **k: None)

    import tiny_renderer_amd as T

    if args.synthetic:
        mesh, texs = T.synthetic_scene()
    else:
        say("loading model from: %s/model.obj" % args.asset_path)
        mesh, texs = T.load_assets(args.asset_path)
    say("number of vertices in a model: %d" % mesh["pos"].shape[0])
    say("number of polygons in a model: %d" % mesh["idx"].shape[0])
    say("cooking up a scene with '%s' shader pipeline" % args.pipeline)
    if sharded:
        import torch
        import torch.distributed as dist
        from .sharded import ShardedScene
        if os.environ.get("TR_CLI_IMPORT_CHECK") == "1":   # test hook: a rank started the way launch_ranks starts it
            print("rank %d of %d: imports ok" % (rank, world))
            return 0
        if args.seconds > 0:
            # every rank must render the same frames: wall-clock driven angles would differ per process
            raise SystemExit("--seconds (the time-based loop) runs on one GPU: use --frames with --gpus")
        if args.view != "frame":
            raise SystemExit("--view %s needs the whole z / shadow buffer on one GPU: use --gpus 1" % args.view)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        n_dev = torch.cuda.device_count()
        if world > n_dev and not args.exchange.startswith("peer"):
            raise SystemExit("%d ranks on %d GPU(s): RCCL needs a device per rank -- use --exchange peer to run several ranks "
                             "on one GPU" % (world, n_dev))
        local %= max(n_dev, 1)
        torch.cuda.set_device(local)
        if args.exchange == "torch":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")   # (the library's exchanges only need a rendezvous for their records)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
        scene = ShardedScene(args.width, args.height, mesh, texs, args.pipeline, device=local, exchange=args.exchange)
    else:
        scene = T.Scene(args.width, args.height, mesh, texs, args.pipeline, device=args.device)
    rc = _run(args, T, scene, sharded, rank, say)
    if sharded:
        import torch.distributed as dist
        scene.close()   # (collective: nobody unmaps a slot a peer may still be reading)
        dist.barrier()
        dist.destroy_process_group()
    return rc


def _run(args, T, scene, sharded, rank, say):

    if args.seconds > 0:
        # app.rs:12-13,160-165,173-199,230-246: angles advance by speed * frame time; a frame counter is
        # printed and reset whenever more than a second has passed
        camera_speed = 3.0
        ca, la = np.float32(args.camera_angle), np.float32(args.light_angle)
        start = last = fps_t = time.perf_counter()
        fps_counter, img = 0, None
        while True:
            now = time.perf_counter()
            if now - start >= args.seconds:
                break
            ca = np.float32(ca + camera_speed * (now - last))
            last = now
            scene.clear()
            scene.set_light_direction([float(np.sin(la)), 0.0, float(np.cos(la))])
            scene.set_camera([float(np.sin(ca)), 0.0, float(np.cos(ca))], [0, 0, 0], [0, 1, 0])
            scene.render()
            if not args.no_readback:
                img = scene.get_frame_buffer()
            elif fps_counter % 64 == 63:
                scene.sync()  # keep the queue bounded
            fps_counter += 1
            if now - fps_t > 1.0:
                say("FPS --- %d" % fps_counter)
                fps_counter, fps_t = 0, now
        scene.sync()
        if args.out:
            img = _view(scene, args.view)
            if rank == 0:
                write_frame(T, args.out, img)
        return 0

    t0 = time.perf_counter()
    angles = [np.float32(args.camera_angle + (2.0 * np.pi * f / args.frames if args.frames > 1 else 0.0))
              for f in range(args.frames)]
    la = np.float32(args.light_angle)
    if args.frames > 1:
        # many frames: the library's throughput path (the same frames, several per kernel launch; with --gpus every
        # rank renders its band of a group, and the bands are exchanged frame by frame)
        p = np.zeros((args.frames, 12), np.float32)
        p[:, 0:3] = [float(np.sin(la)), 0.0, float(np.cos(la))]
        for f, ca in enumerate(angles):
            p[f, 3:6], p[f, 6:9], p[f, 9:12] = [float(np.sin(ca)), 0.0, float(np.cos(ca))], [0, 0, 0], [0, 1, 0]
        scene.render_frames(p)
        angles = []
    for ca in angles:
        scene.clear()                                                        # app.rs:170
        scene.set_light_direction([float(np.sin(la)), 0.0, float(np.cos(la))])   # app.rs:203-208
        scene.set_camera([float(np.sin(ca)), 0.0, float(np.cos(ca))], [0, 0, 0], [0, 1, 0])  # app.rs:200-209
        scene.render()                                                       # app.rs:210
    img = _view(scene, args.view)
    dt = time.perf_counter() - t0
    say("FPS --- %d" % int(args.frames / dt if dt > 0 else 0))              # app.rs:238
    if args.out and rank == 0:
        write_frame(T, args.out, img)
    return 0


def _view(scene, view):
    if view == "frame":
        return scene.get_frame_buffer()
    return {"z": scene.get_z_buffer, "shadow": scene.get_shadow_buffer}[view]()


def write_frame(T, path, img):
    if path.lower().endswith(".tga"):
        T.save_tga(path, img)
    elif path.lower().endswith(".png"):
        T.save_png(path, img)
    else:
        with open(path, "wb") as fh:
            fh.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
            fh.write(img.tobytes())
    print("wrote %s" % path)


if __name__ == "__main__":
    sys.exit(main())
