"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter mean over dispatches."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            full = r["Kernel_Name"]
            name = next((k for k in ("k_tile", "k_setup") if k in full), full)[:40] + ("<%s>" % full.split("<")[1].split(">")[0] if "<" in full else "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "k_tile" in k or "k_setup" in k:
                print(d, k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
