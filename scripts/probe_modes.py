"""k_tile time of each config under every (waves per tile, tile mode) choice: python scripts/probe_modes.py [tags]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T
from bench import find_assets, camera, light
CFG = {"headline": ("diablo", "phong", 4096, 1, 200), "cfg0": ("african_head", "default", 800, 1, 200),
       "cfg1": ("diablo", "phong", 2048, 1, 200), "cfg2": ("diablo", "darboux", 4096, 1, 100),
       "cfg3": ("diablo", "shadow", 4096, 1, 100), "cfg4": ("diablo", "specular", 8192, 8, 20)}
for tag in (sys.argv[1:] or list(CFG)):
    model, pipe, size, grid, steps = CFG[tag]
    mesh, texs = T.load_assets(find_assets(model))
    if grid > 1:
        mesh = T.instanced_grid(mesh, grid)
    out = []
    for waves in (4, 8, 16):
        for mode in (1, 2):
            s = T.Scene(size, size, mesh, texs, pipe, tile_waves=waves, tile_mode=mode)
            def step():
                s.clear(); s.set_light_direction(light(0.0)); s.set_camera(*camera(0.0)); s.render()
            for _ in range(10): step()
            s.sync()
            t0 = time.perf_counter()
            for _ in range(steps): step()
            s.sync()
            dt = (time.perf_counter() - t0) / steps * 1e6
            s.profile_enable(True)
            for _ in range(min(steps, 50)): step()
            p = s.profile_read()
            kt = sum(v["total_ms"] / v["launches"] * 1e3 for k, v in p.items() if k.startswith("k_tile"))
            out.append("%dw/%s %6.1f (frame %6.1f)" % (waves, "col" if mode == 1 else "shr", kt, dt))
            s.close()
    print("%-9s %s" % (tag, "  ".join(out)), flush=True)
