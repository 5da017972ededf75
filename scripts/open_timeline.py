"""Idle / busy analysis of one open from a rocprofv3 kernel trace (all streams together): the open window runs from the start of
k_aggregate to the start of the next commit's first row FFT.  usage: open_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?")), int(r.get("Grid_Size_X", 0) or 0)))
rows.sort()
agg = [i for i, r in enumerate(rows) if "k_aggregate" in r[2]]
if len(agg) < 3:
    sys.exit("need a few steps in the trace")
i0 = agg[-2]                                           # the last-but-one open (the last one may be the profiled extra step)
t0 = rows[i0][0]
lc = next(i for i in range(i0, len(rows)) if "k_leaf_chain" in rows[i][2])                                # the next commit's leaf chain
last = max(i for i in range(i0, lc) if "k_merkle_paths" in rows[i][2])                                     # the last query round of the last _whir_prove
nxt = next(i for i in range(last, lc) if "k_fft4096" in rows[i][2])                                        # the next commit's first row FFT
t1 = rows[nxt][0]
win = [r[:4] for r in rows[i0:nxt]]
busy = 0; cur_s, cur_e = None, None
for s, e, _, _ in sorted(win):
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
streams = {}
for s, e, n, q in win:
    streams.setdefault(q, [0, 0]); streams[q][0] += e - s; streams[q][1] += 1
print("open window %.3f ms, GPU busy (union over streams) %.3f ms, idle %.3f ms, launches %d" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(win)))
for q, (ns, n) in sorted(streams.items(), key=lambda kv: -kv[1][0]):
    print("  stream/queue %s: %.3f ms in %d launches" % (q, ns / 1e6, n))
# the largest gaps of the union
ev = sorted(win); gaps = []; end = ev[0][1]
for s, e, n, q in ev[1:]:
    if s > end: gaps.append((s - end, n))
    end = max(end, e)
print("  largest gaps (us, next kernel):", [(round(g / 1e3, 1), n.split("(")[0][-40:]) for g, n in sorted(gaps, reverse=True)[:8]])
