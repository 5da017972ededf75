"""Where the open's wall time goes: from a rocprofv3 kernel trace of scripts/opentrace.py-like runs, the busy time and the idle gaps
between consecutive kernels of ONE open_standard (the last one in the trace), grouped by the kernel that FOLLOWS the gap.
usage: open_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hobbit::", "").split("<")[0]) for r in rows))
# the last open: from the last k_aggregate_arg to the end
idx = max(i for i, e in enumerate(ev) if e[2].startswith("k_aggregate"))
ev = ev[idx:]
busy = sum(e[1] - e[0] for e in ev); span = ev[-1][1] - ev[0][0]
print("kernels %d, span %.3f ms, busy %.3f ms, idle %.3f ms" % (len(ev), span / 1e6, busy / 1e6, (span - busy) / 1e6))
gaps = collections.defaultdict(lambda: [0, 0]); big = []
for a, b in zip(ev, ev[1:]):
    g = b[0] - a[1]
    if g > 0:
        gaps[b[2]][0] += g; gaps[b[2]][1] += 1
        if g > 30000: big.append((g, a[2], b[2]))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("  before %-22s %7.3f ms in %4d gaps (avg %5.1f us)" % (k, v[0] / 1e6, v[1], v[0] / v[1] / 1e3))
print("gaps > 30 us: %d, total %.3f ms" % (len(big), sum(g for g, _, _ in big) / 1e6))
hist = collections.Counter(min(int(g / 5000), 20) for a, b in zip(ev, ev[1:]) for g in [b[0] - a[1]] if g > 0)
print("gap histogram (5 us bins):", sorted(hist.items()))
