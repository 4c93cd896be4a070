"""Which (waves per tile, tile mode) is fastest at which frame size?  diablo / phong."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
mesh, texs = T.load_assets(find_assets("diablo"))
for size in (512, 800, 1024, 1536, 2048, 2560, 3072, 3584, 4096):
    out = []
    for waves, mode in ((4, 1), (4, 2), (8, 1), (8, 2), (16, 1), (16, 2)):
        s = T.Scene(size, size, mesh, texs, "phong", tile_waves=waves, tile_mode=mode)
        def step():
            s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()
        for _ in range(10): step()
        s.sync()
        t0 = time.perf_counter()
        for _ in range(150): step()
        s.sync()
        dt = (time.perf_counter() - t0) / 150 * 1e6
        s.profile_enable(True)
        for _ in range(40): step()
        p = s.profile_read()
        out.append("%dw/%s %5.1f/%5.1f" % (waves, "col" if mode == 1 else "shr", p["k_tile"]["total_ms"] / p["k_tile"]["launches"] * 1e3, dt))
        s.close()
    print("%4d^2 (%5d tiles)  %s" % (size, ((size + 127) // 128) * ((size + 15) // 16), "  ".join(out)), flush=True)
