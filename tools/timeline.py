"""Kernel timeline of the last steps of a rocprofv3 --kernel-trace run (rocpd sqlite): start, duration, grid, stream, queue."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(cur.execute("select name, start, end, grid_x, stream_id, queue_id from kernels order by start"))
scans = [(s, e) for n, s, e, g, st, q in rows if ("stream_kernel" in n or "lev1_asm_kernel" in n) and e - s > 1_000_000]
steps = [scans[i + 1][0] - scans[i][0] for i in range(3, len(scans) - 1)]
print(f"main scan: n={len(scans)} avg {sum(e - s for s, e in scans[3:]) / max(1, len(scans[3:])) / 1e3:.1f} us; step period (scan start to scan start) avg {sum(steps) / max(1, len(steps)) / 1e3:.1f} us min {min(steps) / 1e3:.1f} max {max(steps) / 1e3:.1f}")
t0 = rows[0][1]
print("start us, duration us, grid, stream, queue, kernel:")
for n, s, e, g, st, q in [r for r in rows if r[1] >= scans[-last][1] and r[1] <= scans[-1][1]]:
    print(f"{(s - t0) / 1e3:12.1f} {(e - s) / 1e3:9.1f} {g:9d} {st:3d} {q:3d}  {n[:80]}")
