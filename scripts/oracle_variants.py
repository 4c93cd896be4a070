"""How much does "parity unpinned" leave open?  (CPU only; run where the reference's assets are.)

The oracle restates nalgebra 0.31.4's operation order from knowledge of the crate (SURVEY.md
Appendix A marks every row UNVERIFIED: the crate's source is not in /root/reference and there is no
Rust toolchain).  This script builds oracle/tr_oracle.c once per ALTERNATIVE reading (TRO_VARIANT
1..6, listed in that file), renders BASELINE.json's configs with each and counts the pixels whose
winner index, z bits or rgb differ from the normative oracle: if upstream's arithmetic were the
variant, that many pixels of a GPU frame that matches our oracle would be off.

    python scripts/oracle_variants.py [--full] [out.json]

--full renders the configs at BASELINE's sizes (800 / 2048 / 4096 / 4096 / 8192: about 15 minutes of
single-core CPU); the default halves the 4096 and 8192 sizes (a few minutes).
"""
import json, os, subprocess, sys
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle as O  # noqa: E402
import tiny_renderer_amd as T   # noqa: E402  (loaders + instancing only; nothing is rendered on a GPU)
from bench import find_assets, camera, light  # noqa: E402

VARIANTS = {1: "dot3 associates right", 2: "4x4 gemv sums pairwise", 3: "gemv accumulates from +0",
            4: "normalize = v * (1/n)", 5: "3x3 inverse = cofactor * (1/det)", 6: "4x4 inverse = cofactor / det"}
full = "--full" in sys.argv
out_path = [a for a in sys.argv[1:] if not a.startswith("--")]
out_path = out_path[0] if out_path else os.path.join(REPO, "profiles", "r02_oracle_variants.json")
S = 1 if full else 2
CONFIGS = [("configs[0] african_head/default/800", "african_head", "default", 800, 1),
           ("configs[1] diablo/phong/2048", "diablo", "phong", 2048, 1),
           ("configs[2] diablo/darboux/%d" % (4096 // S), "diablo", "darboux", 4096 // S, 1),
           ("configs[3] diablo/shadow/%d" % (4096 // S), "diablo", "shadow", 4096 // S, 1),
           ("configs[4] diablo x64/specular/%d" % (8192 // S), "diablo", "specular", 8192 // S, 8)]
ANGLES = [(0.0, 0.0), (0.7, -1.1)]   # the reference's first frame, and a view where no matrix entry is exact

vdir = os.path.join(REPO, "oracle", "_variants")
os.makedirs(vdir, exist_ok=True)
cflags = "-O2 -std=c11 -fPIC -ffp-contract=off -fno-fast-math -fexcess-precision=standard".split()
libs = {}
for k in VARIANTS:
    libs[k] = os.path.join(vdir, "libtr_oracle_v%d.so" % k)
    subprocess.check_call(["gcc"] + cflags + ["-DTRO_VARIANT=%d" % k, "-shared", "-o", libs[k],
                                              os.path.join(REPO, "oracle", "tr_oracle.c"), "-lm"])


def frames(mesh, texs, pipe, size, ca, la):
    s = O.Scene(size, size, mesh, texs, pipe)
    s.clear(); s.set_light_direction(light(la)); s.set_camera(*camera(ca))
    err = s.render()
    r = (err, s.winner_u32(), s.z_f32().view(np.uint32), s.get_frame_buffer())
    s.close()
    return r


report = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "sizes": "BASELINE" if full else "4096 -> 2048, 8192 -> 4096",
          "variants": VARIANTS, "rows": []}
for name, model, pipe, size, grid in CONFIGS:
    mesh, texs = T.load_assets(find_assets(model))
    if grid > 1:
        mesh = T.instanced_grid(mesh, grid)
    for ca, la in ANGLES:
        O.use_library(None)
        base = frames(mesh, texs, pipe, size, ca, la)
        lit = int((base[1] != 0xFFFFFFFF).sum())
        for k in VARIANTS:
            O.use_library(libs[k])
            v = frames(mesh, texs, pipe, size, ca, la)
            d_rgb = (v[3] != base[3]).any(-1)
            row = {"config": name, "camera_light": [ca, la], "variant": k, "lit_pixels": lit,
                   "winner_diff": int((v[1] != base[1]).sum()), "z_diff": int((v[2] != base[2]).sum()),
                   "rgb_diff": int(d_rgb.sum()),
                   "rgb_max_abs": int(np.abs(v[3].astype(np.int16) - base[3].astype(np.int16)).max())}
            report["rows"].append(row)
            print("%-36s cam %.1f light %.1f  v%d %-32s winner %7d  z %8d  rgb %8d (max |d| %d) of %d lit" % (
                name, ca, la, k, VARIANTS[k], row["winner_diff"], row["z_diff"], row["rgb_diff"], row["rgb_max_abs"], lit), flush=True)
O.use_library(None)
json.dump(report, open(out_path, "w"), indent=1)
print("wrote", out_path)
