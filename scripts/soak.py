"""Soak test (not part of the suite): many frames in flight with a moving camera, every k-th frame read
back and compared with the oracle.  python scripts/soak.py [seconds per pipeline]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tiny_renderer_amd as T
from oracle import oracle as O
from bench import find_assets
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
adir = find_assets("diablo")
mesh, texs = T.load_assets(adir) if adir else T.synthetic_scene()
bad = 0
for pipe, size in (("phong", 1024), ("shadow", 800), ("darboux", 800), ("specular", 640), ("phong", 4096)):
    gpu = T.Scene(size, size, mesh, texs, pipe)
    cpu = O.Scene(size, size, mesh, texs, pipe)
    t0 = time.time(); frames = checks = 0
    every = 4001 if size < 4096 else 20011
    while time.time() - t0 < budget:
        a = np.float32(0.001 * frames)
        cam = ([float(np.sin(a)), 0.0, float(np.cos(a))], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0])
        lt = [float(np.sin(a * np.float32(0.5) + np.float32(0.4))), 0.0, float(np.cos(a * np.float32(0.5) + np.float32(0.4)))]
        gpu.clear(); gpu.set_light_direction(lt); gpu.set_camera(*cam); gpu.render()
        frames += 1
        if frames % every == 0:
            fb = gpu.get_frame_buffer()
            cpu.clear(); cpu.set_light_direction(lt); cpu.set_camera(*cam)
            assert cpu.render() == 0
            ok = np.array_equal(fb, cpu.get_frame_buffer())
            checks += 1
            if not ok:
                bad += 1
                print("MISMATCH", pipe, size, "frame", frames, flush=True)
        elif frames % 512 == 0:
            gpu.sync()   # bound the queue
    st = gpu.sync()
    print("%-9s %4d^2: %7d frames in %.0f s (%.0f fps), %d oracle checks, status %d" % (pipe, size, frames, time.time() - t0, frames / (time.time() - t0), checks, st), flush=True)
    gpu.close()
print("soak done, mismatches:", bad)
