"""Turns gpurun_out/prof (scripts/collect_profiles.sh) into the files kept under profiles/.

    python scripts/summarise_profiles.py [round-tag]     (default r01)
"""
import collections, csv, glob, json, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", "prof")
dst = os.path.join(REPO, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


def kernel_of(name):
    for k in ("k_tile", "k_setup", "k_order_count", "k_order_place", "k_materialize_depth", "k_fill_u32", "k_depth_view"):
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

traffic = counters("pmc_*")
rows = []
for k, cs in sorted(traffic.items()):
    for c, v in sorted(cs.items()):
        rows.append((k, c, v))
with open(os.path.join(dst, tag + "_pmc_fetch_write.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch\n")
    for r in rows:
        f.write("%s,%s,%.3f\n" % r)
kt = traffic.get("k_tile", {})
fetch_kb, write_kb = kt.get("FETCH_SIZE", 0.0), kt.get("WRITE_SIZE", 0.0)
json.dump({"diablo.obj, -s phong, 4096x4096": {
    "kernel": "k_tile",
    "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024.0)),
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
    "other_kernels_kb": {k: {c: v for c, v in cs.items()} for k, cs in traffic.items() if k != "k_tile"},
    "correction": "gfx950: FETCH_SIZE x2 (128-B requests tallied at 64 B), WRITE_SIZE as is; separate --pmc passes (MI355X_MICROARCH.md, HBM)",
    "source": "profiles/%s_pmc_fetch_write.csv" % tag}}, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)

sq = counters("sq_*")
json.dump({k: {c: round(v, 1) for c, v in cs.items()} for k, cs in sq.items() if k.startswith("k_")},
          open(os.path.join(dst, tag + "_pmc_sq_4096_phong.json"), "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, tag + "_bench_4096_phong.log")).read().strip().splitlines()[-1])
print(json.dumps(json.load(open(os.path.join(dst, "pmc_traffic.json"))), indent=1)[:600])
