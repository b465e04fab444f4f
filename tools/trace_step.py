"""Timeline of ONE time step from a rocprofv3 --kernel-trace CSV: kernels in start order with their durations and the
gaps in front of them.  usage: python tools/trace_step.py <kernel_trace.csv> [marker kernel substring] [which occurrence]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
mark = sys.argv[2] if len(sys.argv) > 2 else "k_rhs_p1w<true"
occ = int(sys.argv[3]) if len(sys.argv) > 3 else -3
idx = [i for i, r in enumerate(rows) if mark in r[2]]
i0, i1 = idx[occ], idx[occ + 1]
# a step = from the limiter in front of the marked RHS to the one in front of the next
while i0 > 0 and "k_rhs_p1w" not in rows[i0 - 1][2]:
    i0 -= 1
while i1 > 0 and "k_rhs_p1w" not in rows[i1 - 1][2]:
    i1 -= 1
tot_k = tot_g = 0
prev = rows[i0 - 1][1]
for s, e, n in rows[i0:i1]:
    short = n.split("(")[0][-60:]
    print("%8.1f us gap %7.1f us  %s" % ((s - prev) / 1e3, (e - s) / 1e3, short))
    tot_g += s - prev; tot_k += e - s; prev = e
print("step: kernels %.1f us, gaps %.1f us, total %.1f us, launches %d" % (tot_k / 1e3, tot_g / 1e3, (tot_k + tot_g) / 1e3, i1 - i0))
