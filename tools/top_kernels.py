#!/usr/bin/env python3
"""top kernels of a rocprofv3 --kernel-trace --stats run (rocpd SQLite): python tools/top_kernels.py <results.db> [rows]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 12
print(f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
for name, calls, total, avg, pct in list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:rows]:
    print(f"{name[:100]:100s} {calls:6d} {total / 1e3:12.1f} {avg / 1e3:10.2f} {pct:6.2f}")
