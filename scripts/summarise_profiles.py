"""Turns gpurun_out/prof (scripts/collect_profiles.sh) into the files kept under profiles/.

    python scripts/summarise_profiles.py [round-tag]     (default r03)

profiles/pmc_traffic.json is stamped with bench.source_fingerprint() of the tree it is run in: run it
right after the GPU call, on the sources that were profiled; bench.py reports `roofline.traffic`
only while that fingerprint matches.
"""
import collections, csv, glob, json, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import source_fingerprint  # noqa: E402

src = os.path.join(REPO, "gpurun_out", "prof")
dst = os.path.join(REPO, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"

WORKLOADS = {  # tag of collect_profiles.sh -> (bench workload string, dominant kernel)
    "headline": ("diablo.obj, -s phong, 4096x4096", "k_tile"),
    "cfg0": ("african_head.obj, -s default, 800x800", "k_tile"),
    "cfg1": ("diablo.obj, -s phong, 2048x2048", "k_tile"),
    "cfg2": ("diablo.obj, -s darboux, 4096x4096", "k_tile"),
    "cfg3": ("diablo.obj, -s shadow, 4096x4096", "k_tile"),
    "cfg4": ("diablo.obj x64 grid, -s specular, 8192x8192", "k_tile"),
}


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


def kernel_of(name):
    if "k_tile" in name:
        return "k_tile_depth" if "<7," in name.replace(" ", "") or "FS_DEPTH" in name else "k_tile"
    for k in ("k_setup", "k_order", "k_bin", "k_lit", "k_read_back", "k_push_tiles", "k_materialize_depth", "k_fill_u32", "k_depth_view"):
        if k in name:
            return k
    return name[:48]


def counters(dirpattern):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in glob.glob(os.path.join(src, dirpattern)):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[kernel_of(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


shutil.copy(one("bench_4096_phong.log"), os.path.join(dst, tag + "_bench_4096_phong.log"))
shutil.copy(one("bench_under_rocprof.log"), os.path.join(dst, tag + "_bench_4096_phong_under_rocprof.log"))
shutil.copy(one("trace/**/*kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats_4096_phong.csv"))

db = {"source_fingerprint": source_fingerprint(),
      "correction": "gfx950: FETCH_SIZE x2 (128-B requests tallied at 64 B), WRITE_SIZE as is; separate --pmc passes "
                    "(MI355X_MICROARCH.md, HBM); means over the launches of scripts/frame_loop.py (a launch covers frames_per_launch frames)",
      "workloads": {}}
rows = []
for wtag, (workload, dom) in WORKLOADS.items():
    traffic = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, cs in counters("pmc_%s_%s" % (c, wtag)).items():
            traffic.setdefault(k, {}).update(cs)
    if dom not in traffic:
        print("no counters for", wtag)
        continue
    for k, cs in sorted(traffic.items()):
        for c, v in sorted(cs.items()):
            rows.append((wtag, k, c, v))
    kt = traffic[dom]
    fetch_kb, write_kb = kt.get("FETCH_SIZE", 0.0), kt.get("WRITE_SIZE", 0.0)
    fpl = 1
    for line in open(one("pmc_WRITE_SIZE_%s.log" % wtag)):
        if line.startswith("frames "):
            fpl = int(line.split()[3])   # "frames N frames_per_launch G" (scripts/frame_loop.py)
    db["workloads"][workload] = {
        "kernel": dom,
        "frames_per_launch": fpl,
        "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024.0)),
        "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
        # every tile kernel of the workload (a two-pass pipeline has two; bench.py asks for the dominant one)
        "per_kernel_hbm_bytes_per_launch": {k: int(round((2.0 * cs.get("FETCH_SIZE", 0.0) + cs.get("WRITE_SIZE", 0.0)) * 1024.0))
                                            for k, cs in traffic.items() if k.startswith("k_tile")},
        "other_kernels_kb": {k: cs for k, cs in traffic.items() if k != dom},
        "source": "profiles/%s_pmc_fetch_write.csv" % tag}
with open(os.path.join(dst, tag + "_pmc_fetch_write.csv"), "w") as f:
    f.write("workload,kernel,counter,mean_per_launch_kb\n")
    for r in rows:
        f.write("%s,%s,%s,%.3f\n" % r)
json.dump(db, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)

sq = counters("sq_*")
# Vector-issue share of the headline's tile kernel: SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD executes a
# vector instruction, summed over the 1 024 SIMDs; the launch lasted GRBM_GUI_ACTIVE / 8 cycles (the counter is summed
# over the 8 XCDs: MI355X_MICROARCH.md, DVFS).  bench.py reports it as roofline.issue_frac while the fingerprint matches.
head = db["workloads"].get(WORKLOADS["headline"][0])
kt = sq.get("k_tile", {})
if head is not None and kt.get("SQ_ACTIVE_INST_VALU") and kt.get("GRBM_GUI_ACTIVE"):
    launch_cycles = kt["GRBM_GUI_ACTIVE"] / 8.0
    head["issue"] = {"SQ_INSTS_VALU": kt.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": kt.get("SQ_INSTS_SALU"),
                     "SQ_ACTIVE_INST_VALU_quad_cycles": kt["SQ_ACTIVE_INST_VALU"], "GRBM_GUI_ACTIVE": kt["GRBM_GUI_ACTIVE"],
                     "launch_cycles": launch_cycles, "simds": 1024,
                     "issue_frac": round(4.0 * kt["SQ_ACTIVE_INST_VALU"] / (1024.0 * launch_cycles), 4),
                     "source": "profiles/%s_pmc_sq_4096_phong.json" % tag}
    json.dump(db, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
json.dump({k: {c: round(v, 1) for c, v in cs.items()} for k, cs in sq.items() if k.startswith("k_")},
          open(os.path.join(dst, tag + "_pmc_sq_4096_phong.json"), "w"), indent=1, sort_keys=True)
# configs[4]'s instruction counters (before / after): "before" = round 2's library measured with the same
# command on the same box earlier this round (profiles/r03_notes.md), "after" = this build
sq4 = counters("sqcfg4_*")
if sq4.get("k_tile"):
    json.dump({"workload": WORKLOADS["cfg4"][0], "per": "launch of 4 frames (mean over the launches of scripts/frame_loop.py 8192 specular 8 diablo 8)",
               "before_round2_library": {"k_tile": {"SQ_INSTS_VALU": 628.4e6, "SQ_INSTS_SALU": 296.8e6, "SQ_INSTS_LDS": 21.0e6}, "note": "same command, same box, round 2's library (gpurun_out/r3c; profiles/r03_notes.md)"},
               "after": {k: {c: round(v, 1) for c, v in cs.items()} for k, cs in sq4.items() if k.startswith("k_")}},
              open(os.path.join(dst, tag + "_pmc_sq_cfg4.json"), "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, tag + "_bench_4096_phong.log")).read().strip().splitlines()[-1][:600])
for w, e in db["workloads"].items():
    print("%-50s %s %.1f MB/launch of %d frames" % (w, e["kernel"], e["hbm_bytes_per_launch"] / 1e6, e["frames_per_launch"]))
