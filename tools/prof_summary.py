"""Diagnostic (not collected): print the top kernels of a rocprofv3 --stats run."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>6s} tot_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={float(r['Percentage']):6.2f}")
