"""Idle gaps on the GPU timeline of a rocprofv3 kernel trace: python tools/gaps.py <prof dir> [min_gap_us]
Prints busy/idle totals of the last third of the trace and the largest gaps with the kernels around them."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) * 2 // 3:]
t_end, idle, gaps = int(rows[0]["End_Timestamp"]), 0, []
for prev, r in zip(rows, rows[1:]):
    s = int(r["Start_Timestamp"])
    if s > t_end:
        g = (s - t_end) / 1e3
        idle += g
        if g >= thr:
            gaps.append((g, prev["Kernel_Name"][:60], r["Kernel_Name"][:60]))
    t_end = max(t_end, int(r["End_Timestamp"]))
span = (t_end - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"span {span/1e3:.2f} ms, idle {idle/1e3:.2f} ms ({100*idle/span:.1f} %), gaps >= {thr} us: {len(gaps)}")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"{g:8.1f} us  after {a}  before {b}")
