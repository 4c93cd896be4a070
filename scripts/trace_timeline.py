"""Prints the kernel timeline of the LAST burst of dispatches in a rocprofv3 --kernel-trace CSV (bursts are separated
by >5 ms of idle): python scripts/trace_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur) > 5_000_000:
        bursts.append(cur); cur = []
    cur.append(r)
bursts.append(cur)
b = bursts[-1] if len(sys.argv) < 3 else bursts[int(sys.argv[2])]
t0 = b[0][0]
def short(n):
    for k in ("k_setup", "k_order", "k_bin", "k_tile", "k_clear", "k_read_back"):
        if k in n: return k
    return n[:24]
for st, en, n in b:
    print("%9.1f  %9.1f  %7.1f  %s" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, short(n)))
print("burst: %d kernels, %.1f us from first start to last end" % (len(b), (max(x[1] for x in b) - t0) / 1e3))
