"""Timeline of a rocprofv3 --kernel-trace csv: gaps between consecutive k_tile launches and where the
setup-stream kernels ran.  python scripts/trace_timeline.py out_kernel_trace.csv [first_row]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    short = "k_tile" if "k_tile" in n else "k_setup" if "k_setup" in n else "k_order_count" if "k_order_count" in n else "k_order_place" if "k_order_place" in n else None
    if short: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "")))
ev.sort()
t0 = ev[0][0]
tiles = [e for e in ev if e[2] == "k_tile"]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(tiles) // 2
print("k_tile launches", len(tiles))
for i in range(skip, min(skip + 8, len(tiles) - 1)):
    a, b = tiles[i], tiles[i + 1]
    inside = [e for e in ev if e[2] != "k_tile" and e[0] < b[1] and e[1] > a[0]]
    print("tile %3d: start %9.1f dur %6.1f | gap to next %5.1f" % (i, (a[0] - t0) / 1e3, (a[1] - a[0]) / 1e3, (b[0] - a[1]) / 1e3))
    for e in inside:
        if e[0] >= a[0] and e[0] < b[0]:
            print("      %-14s start +%6.1f dur %6.1f (ends %+6.1f vs tile end)" % (e[2], (e[0] - a[0]) / 1e3, (e[1] - e[0]) / 1e3, (e[1] - a[1]) / 1e3))
gaps = [(tiles[i + 1][0] - tiles[i][1]) / 1e3 for i in range(len(tiles) // 4, len(tiles) - 1)]
durs = [(t[1] - t[0]) / 1e3 for t in tiles[len(tiles) // 4:]]
import statistics
print("steady state: k_tile dur median %.1f, gap median %.1f mean %.1f max %.1f" % (statistics.median(durs), statistics.median(gaps), sum(gaps) / len(gaps), max(gaps)))
