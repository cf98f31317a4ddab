#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for i, r in enumerate(csv.reader(open(f))):
    if i <= n:
        print("%-72s %s" % (r[0][:72], "  ".join(r[1:5])))
