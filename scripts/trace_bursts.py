"""Summary of every burst of dispatches (separated by > 1 ms of idle) in a rocprofv3 --kernel-trace CSV:
    python scripts/trace_bursts.py <kernel_trace.csv>"""
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur) > 1_000_000:
        bursts.append(cur); cur = []
    cur.append(r)
bursts.append(cur)
def short(n):
    for k in ("k_setup", "k_order", "k_bin", "k_tile", "k_clear", "k_read_back", "copyBuffer", "fillBuffer"):
        if k in n: return k
    return n[:20]
for i, b in enumerate(bursts):
    t0 = b[0][0]
    names = {}
    for st, en, n in b: names[short(n)] = names.get(short(n), 0) + 1
    tiles = [((st - t0) / 1e3, (en - t0) / 1e3) for st, en, n in b if "k_tile" in n]
    print("burst %2d: %4d dispatches %8.1f us  %s  tiles %s" % (i, len(b), (max(x[1] for x in b) - t0) / 1e3, names, ["%.0f-%.0f" % t for t in tiles[:8]]))
