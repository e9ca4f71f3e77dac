#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite outputs (kernel trace + PMC passes) into the small text/JSON summaries
committed under profiles/.  Usage:
  tools/rocpd_summary.py --kernel-trace gpurun_out/prof_kt/kt_results.db --pmc gpurun_out/prof_pmc*/p*_results.db \
      --match scan_kernel --out profiles/c2_levenshtein_r01
FETCH_SIZE correction: on gfx950 rocprofv3 reports exactly 1/2 of the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM section), so hbm_read_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as is."""
import argparse
import json
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel-trace")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--match", default="scan_kernel")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--skip", type=int, default=3, help="warm-up dispatches left out of the steady-state average")
    ap.add_argument("--traffic-key", default=None, help="also record the HBM traffic under this key in --traffic-json")
    ap.add_argument("--traffic-json", default="profiles/traffic.json")
    a = ap.parse_args()
    lines, summary = [], {"note": a.note}
    if a.kernel_trace:
        cur = sqlite3.connect(a.kernel_trace).cursor()
        lines.append("== rocprofv3 --kernel-trace --stats : top kernels (name, calls, total_us, avg_us, pct)")
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append(f"{name[:110]:110s} {calls:6d} {total:14.3f} {avg:12.3f} {pct:7.2f}")
            if a.match in name and ("kernel" not in summary or total > summary["kernel"]["total_us"]):
                summary["kernel"] = {"name": name, "calls": calls, "avg_us": avg, "total_us": total}
        dominant = summary.get("kernel", {}).get("name", "")
        rows = list(cur.execute("select name, duration, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size from kernels where name = ? order by start", (dominant,)))
        if rows:  # the sample pass of a top-k call is the same kernel on a small grid: keep the full-size dispatches
            gmax = max(r[2] for r in rows)
            rows = [r for r in rows if r[2] == gmax]
        if rows:
            d = [r[1] for r in rows]
            lines.append(f"== {a.match}: {len(d)} dispatches, duration ns min/avg/max = {min(d)}/{sum(d)/len(d):.0f}/{max(d)}; grid_x={rows[0][2]} wg_x={rows[0][3]} vgpr={rows[0][4]} sgpr={rows[0][5]} lds={rows[0][6]}")
            # bench.py times only the steps after its warm-up: the same window here (the first dispatches run on cold clocks / caches)
            main = [r[1] for r in rows if r[0] == rows[-1][0]]
            steady = main[a.skip:] if len(main) > a.skip else main
            lines.append(f"== steady state (dispatches after the first {a.skip} of {rows[-1][0][:60]}): n={len(steady)} avg {sum(steady)/len(steady):.0f} ns  median {sorted(steady)[len(steady)//2]} ns")
            summary["steady_avg_us"] = sum(steady) / len(steady) / 1e3
    counters = {}
    for f in a.pmc:
        cur = sqlite3.connect(f).cursor()
        best = {}
        for name, grid, cname, val, n in cur.execute("select kernel_name, grid_size, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, grid_size, counter_name", (f"%{a.match}%",)):
            if cname not in best or grid * n > best[cname][0]:  # the dominant (largest-grid, most-dispatched) instance
                best[cname] = (grid * n, val)
        for cname, (_, val) in best.items():
            counters[cname] = val
    if counters:
        lines.append("== PMC (average per dispatch of the matched kernel; one rocprofv3 --pmc pass per group)")
        for k in sorted(counters):
            lines.append(f"{k:28s} {counters[k]:20.1f}")
        summary["pmc"] = counters
        if "FETCH_SIZE" in counters or "WRITE_SIZE" in counters:
            rd = 2 * counters.get("FETCH_SIZE", 0) * 1024
            wr = counters.get("WRITE_SIZE", 0) * 1024
            summary["hbm_traffic_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr, "correction": "read = 2 x FETCH_SIZE KiB (gfx950 wide-stream undercount), write = WRITE_SIZE KiB"}
            lines.append(f"== HBM traffic per launch: read {rd/1e9:.3f} GB (2 x FETCH_SIZE), write {wr/1e9:.3f} GB, total {(rd+wr)/1e9:.3f} GB")
    if a.traffic_key and "hbm_traffic_bytes_per_launch" in summary:
        try:
            table = json.load(open(a.traffic_json))
        except (OSError, ValueError):
            table = {}
        e = dict(summary["hbm_traffic_bytes_per_launch"])
        e["source"] = a.out + ".json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        if "steady_avg_us" in summary:
            e["kernel_us_at_collection"] = summary["steady_avg_us"]  # bench.py drops the entry when the kernel has changed since
        table[a.traffic_key] = e
        json.dump(table, open(a.traffic_json, "w"), indent=1, sort_keys=True)
    open(a.out + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(a.out + ".json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
