"""Soak test of the frame groups (not part of the suite): calls of random length with a moving camera and light,
per-frame renders in between, one random kept frame of every k-th call compared with the oracle (z bits, rgb,
shadow bits).  python scripts/soak_groups.py [seconds per pipeline]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tiny_renderer_amd as T
from oracle import oracle as O
from bench import find_assets
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
adir = find_assets("diablo")
mesh, texs = T.load_assets(adir) if adir else T.synthetic_scene()
rng = np.random.default_rng(7)
bad = 0
exact_spec = bool(T.load_library().tr_specular_exact())
for pipe, size in (("phong", 1024), ("shadow", 800), ("occlusion", 512), ("darboux", 800), ("specular", 640), ("phong", 4096), ("shadow", 2048)):
    gpu = T.Scene(size, size, mesh, texs, pipe)
    cpu = O.Scene(size, size, mesh, texs, pipe)
    t0 = time.time(); frames = calls = checks = 0
    every = 40 if size < 2048 else 150
    while time.time() - t0 < budget:
        n = int(rng.integers(1, 300))
        a = (np.float32(0.001) * np.arange(frames, frames + n, dtype=np.float32)).astype(np.float32)
        p = np.zeros((n, 12), np.float32)
        la = a * np.float32(0.5) + np.float32(0.4)
        p[:, 0], p[:, 2] = np.sin(la), np.cos(la)
        p[:, 3], p[:, 5] = np.sin(a), np.cos(a)
        p[:, 10] = 1.0
        gpu.render_frames(p)
        frames += n; calls += 1
        if calls % every == 0:
            back = int(rng.integers(0, gpu.frames_kept()))
            gpu.select_frame(back)
            q = p[n - 1 - back]
            cpu.clear(); cpu.set_light_direction(q[0:3]); cpu.set_camera(q[3:6], q[6:9], q[9:12])
            assert cpu.render() == 0
            okz = np.array_equal(gpu.read_z_f32().view(np.uint32), cpu.z_f32().view(np.uint32))
            d = np.abs(gpu.get_frame_buffer().astype(int) - cpu.get_frame_buffer().astype(int)).max()
            oks = pipe not in ("shadow", "occlusion") or np.array_equal(gpu.read_shadow_f32().view(np.uint32), cpu.shadow_f32().view(np.uint32))
            checks += 1
            if not (okz and oks and d <= (1 if pipe == "specular" and not exact_spec else 0)):
                bad += 1
                print("MISMATCH", pipe, size, "call", calls, "back", back, okz, d, oks, flush=True)
            gpu.select_frame(0)
        if calls % 7 == 0:   # a few per-frame renders between the calls (one accumulating)
            gpu.clear(); gpu.set_light_direction(p[0, 0:3]); gpu.set_camera(p[0, 3:6], p[0, 6:9], p[0, 9:12]); gpu.render()
            gpu.render()
            frames += 2
    st = gpu.sync()
    print("%-9s %4d^2: %8d frames in %d calls, %.0f s (%.0f fps), %d oracle checks, status %d" % (
        pipe, size, frames, calls, time.time() - t0, frames / (time.time() - t0), checks, st), flush=True)
    gpu.close()
print("soak done, mismatches:", bad)
