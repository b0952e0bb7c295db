"""Summarise a rocprofv3 kernel_stats.csv: python tools_profsum.py gpurun_out/prof_x [N]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} tot_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/tot*100:5.1f}%")
