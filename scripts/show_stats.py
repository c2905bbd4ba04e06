"""prints the top rows of a rocprofv3 kernel_stats csv: python scripts/show_stats.py <csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['Name'][:88]:88s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
