"""Headless counterpart of the reference's CLI (src/main.rs:9-39: `-p <asset dir>`, `-s <pipeline>`).

The reference opens a window and orbits with the keyboard; here the camera / light angles are
explicit and frames are written as binary PPM.  Rendering happens on the GPU through the C ABI.
"""
import argparse
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
    ap.add_argument("--out", default=None, help="write the last frame: .tga (24-bit) or binary PPM otherwise")
    ap.add_argument("--view", choices=("frame", "z", "shadow"), default="frame")  # app.rs:213-215
    ap.add_argument("--device", type=int, default=-1)
    ap.add_argument("--synthetic", action="store_true", help="procedural scene instead of -p")
    args = ap.parse_args(argv)

    import tiny_renderer_amd as T

    if args.synthetic:
        mesh, texs = T.synthetic_scene()
    else:
        print("loading model from: %s/model.obj" % args.asset_path)
        mesh, texs = T.load_assets(args.asset_path)
    print("number of vertices in a model: %d" % mesh["pos"].shape[0])
    print("number of polygons in a model: %d" % mesh["idx"].shape[0])
    print("cooking up a scene with '%s' shader pipeline" % args.pipeline)
    scene = T.Scene(args.width, args.height, mesh, texs, args.pipeline, device=args.device)

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
                print("FPS --- %d" % fps_counter)
                fps_counter, fps_t = 0, now
        scene.sync()
        if args.out:
            img = {"frame": scene.get_frame_buffer, "z": scene.get_z_buffer, "shadow": scene.get_shadow_buffer}[args.view]()
            write_frame(T, args.out, img)
        return 0

    t0 = time.perf_counter()
    for f in range(args.frames):
        ca = np.float32(args.camera_angle + (2.0 * np.pi * f / args.frames if args.frames > 1 else 0.0))
        la = np.float32(args.light_angle)
        scene.clear()                                                        # app.rs:170
        scene.set_light_direction([float(np.sin(la)), 0.0, float(np.cos(la))])   # app.rs:203-208
        scene.set_camera([float(np.sin(ca)), 0.0, float(np.cos(ca))], [0, 0, 0], [0, 1, 0])  # app.rs:200-209
        scene.render()                                                       # app.rs:210
    img = {"frame": scene.get_frame_buffer, "z": scene.get_z_buffer, "shadow": scene.get_shadow_buffer}[args.view]()
    dt = time.perf_counter() - t0
    print("FPS --- %d" % int(args.frames / dt if dt > 0 else 0))              # app.rs:238
    if args.out:
        write_frame(T, args.out, img)
    return 0


def write_frame(T, path, img):
    if path.lower().endswith(".tga"):
        T.save_tga(path, img)
    else:
        with open(path, "wb") as fh:
            fh.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
            fh.write(img.tobytes())
    print("wrote %s" % path)


if __name__ == "__main__":
    sys.exit(main())
