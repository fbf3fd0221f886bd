#!/usr/bin/env python3
"""Print the first rows of the kernel_stats.csv that `rocprofv3 --kernel-trace --stats --output-format csv -d DIR` left under DIR.

    python tools/kernel_stats_top.py DIR [rows]
"""
import csv
import glob
import sys

d = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:rows]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(6), ("%.3f ms total" % (int(r["TotalDurationNs"]) / 1e6)).rjust(18),
          ("%.1f us avg" % (float(r["AverageNs"]) / 1e3)).rjust(16), r["Percentage"].rjust(6))
