"""Kernel timeline between the last occurrences of an anchor kernel in a rocprofv3 --kernel-trace run (rocpd sqlite): start, duration,
gap to the previous kernel's end, grid, stream, kernel -- for steps made of several short launches (tools/timeline.py keys on the
long scan kernels).  usage: timeline_any.py <db> <anchor substring> [steps]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
anchor, steps = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = list(cur.execute("select name, start, end, grid_x, stream_id from kernels order by start"))
marks = [i for i, r in enumerate(rows) if anchor in r[0]]
sel = rows[marks[-steps - 1]:marks[-1]]
t0, prev_end = sel[0][1], sel[0][1]
print("start us, duration us, gap us, grid, stream, kernel:")
for n, s, e, g, st in sel:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f} {g:9d} {st:3d}  {n[:70]}")
    prev_end = max(prev_end, e)
per = [rows[marks[i + 1]][1] - rows[marks[i]][1] for i in range(len(marks) // 2, len(marks) - 1)]
print(f"step period (anchor start to anchor start), second half of the run: avg {sum(per) / len(per) / 1e3:.1f} us")
