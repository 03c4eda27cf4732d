#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite db) as CSV."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("name,total_calls,total_duration_us,average_us,percentage")
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(",".join(str(x).replace(",", ";") for x in r))
