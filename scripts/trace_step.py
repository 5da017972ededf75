"""One middle step of a rocprofv3 kernel trace of bench.py as a timeline: start / end / duration (ms from the commit's first kernel), queue, kernel,
grid, plus what else was running when each kernel started.  usage: trace_step.py <kernel_trace.csv> [which leaf chain, default 4]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 4
def nm(r): return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hobbit::", "").split("<")[0]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
big = [i for i, r in enumerate(rows) if nm(r) == "k_fft4096" and int(r["Grid_Size_X"]) >= (1 << 23)]
starts = [i for k, i in enumerate(big) if k == 0 or not any(nm(rows[j]) == "k_fft4096" and int(rows[j]["Grid_Size_X"]) >= (1 << 23) for j in range(max(0, i - 6), i))]
# a commit = a run of big FFTs; its first launch follows >= 6 launches with no big FFT
a, b = starts[which], starts[which + 1]
t0 = S(rows[a])
print("step: %.3f ms" % ((S(rows[b]) - t0) / 1e6))
busy_end = 0
for r in rows[a:b]:
    running = [nm(x) + ":q" + x["Queue_Id"] for x in rows[a:b] if x is not r and S(x) <= S(r) < E(x)]
    gap = (S(r) - busy_end) / 1e6 if busy_end else 0.0
    busy_end = max(busy_end, E(r))
    print("%8.3f %8.3f %7.3f q%s %-22s %9s  %s%s" % ((S(r) - t0) / 1e6, (E(r) - t0) / 1e6, (E(r) - S(r)) / 1e6, r["Queue_Id"], nm(r), r["Grid_Size_X"],
                                                   ("IDLE %.3f  " % gap) if gap > 0.05 else "", ",".join(running[:4])))
