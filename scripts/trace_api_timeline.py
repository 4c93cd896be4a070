"""Merged timeline (host HIP calls, copies, kernels) of the LAST burst in a rocprofv3 --hip-trace --kernel-trace
--memory-copy-trace run: python scripts/trace_api_timeline.py <dir with out_*_trace.csv> [burst index]"""
import csv, glob, os, sys
d = sys.argv[1]
def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
ev = []
for r in rows("*kernel_trace.csv"):
    n = r["Kernel_Name"]
    for k in ("k_setup", "k_order", "k_bin", "k_tile", "k_lit", "k_read_back"):
        if k in n: n = k
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU  " + n[:40]))
for r in rows("*memory_copy_trace.csv"):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Name", "")[:30]))
for r in rows("*hip_api_trace.csv"):
    f = r["Function"]
    if f in ("hipGetLastError", "hipGetDevice", "hipSetDevice", "__hipPushCallConfiguration", "__hipPopCallConfiguration", "hipPeekAtLastError"):
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host " + f))
ev.sort()
gpu = [e for e in ev if e[2].startswith("GPU")]
bursts, cur = [], [gpu[0]]
for e in gpu[1:]:
    if e[0] - max(x[1] for x in cur) > 5_000_000:
        bursts.append(cur); cur = []
    cur.append(e)
bursts.append(cur)
b = bursts[-1] if len(sys.argv) < 3 else bursts[int(sys.argv[2])]
lo, hi = b[0][0] - 150_000, max(x[1] for x in b) + 100_000
t0 = b[0][0]
for st, en, n in ev:
    if st < lo or st > hi: continue
    print("%9.1f  %9.1f  %7.1f  %s" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, n))
