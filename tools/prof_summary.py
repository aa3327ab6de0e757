#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv (top kernels, per-forward counts)."""
import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/prof*/*/*_kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f"# {f}\n# total GPU kernel time {tot/1e6:.1f} ms")
print(f"{'kernel':100s} {'calls':>7s} {'avg_us':>9s} {'total_ms':>9s} {'pct':>6s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.1f} {int(r['TotalDurationNs'])/1e6:9.1f} {float(r['Percentage']):6.2f}")
