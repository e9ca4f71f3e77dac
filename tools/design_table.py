#!/usr/bin/env python3
"""Regenerates the measurement table of DESIGN.md section 6 from the committed bench lines (profiles/bench_*.json), between the
markers `<!-- bench-table:begin -->` and `<!-- bench-table:end -->`, so the prose cannot drift from the evidence
(tests/test_docs.py checks the table is current).   python tools/design_table.py [--check]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [  # (file stem, label)
    ("c2_levenshtein", "C2 Levenshtein, query 64 x 100 M len 64, no cutoff (the default `bench.py` line)"),
    ("q32_levenshtein", "same corpus, query 32 (32-bit kernel)"),
    ("ragged_levenshtein", "ragged: 100 M candidates, lengths uniform in [1, 64] (configs[0]'s distribution), Levenshtein query 64 (`--ragged`)"),
    ("ragged_q32_levenshtein", "ragged, query 32"),
    ("ragged_osa", "ragged, OSA"),
    ("ragged_indel", "ragged, Indel"),
    ("ragged_jaro_winkler", "ragged, Jaro-Winkler (f64 out)"),
    ("ragged_indel_slots", "ragged, Indel, results in slot order (`RF_FLAG_SLOT_ORDER`: no gather pass, round 6)"),
    ("ragged_levenshtein_slots", "ragged, Levenshtein query 64, slot order"),
    ("ragged_jaro_winkler_slots", "ragged, Jaro-Winkler, slot order"),
    ("ragged_lognormal_indel", "ragged, log-normal lengths (median 24) and Zipf(1.1) symbols, Indel"),
    ("ragged_cutoff3", "ragged [1, 64], `score_cutoff = 3` (4 of 64 lengths inside the window; the rest is the None pre-fill)"),
    ("ragged57_cutoff3", "ragged, lengths uniform in [57, 64], `score_cutoff = 3` (half the corpus inside the window: length-run views, round 4)"),
    ("ragged57_osa_cutoff3", "the same, OSA"),
    ("ragged57_filter_cutoff3", "the same corpus, Levenshtein, compact pairs (`--mode filter`: slot-ordered temporary + compaction)"),
    ("c3_levenshtein_256", "C3 Levenshtein, query 256 x 10 M len 256 (4-word asm scan, Ukkonen band: round 5)"),
    ("levenshtein_320", "query 320 x 4 M len 320 (5-word asm scan, round 5)"),
    ("levenshtein_512", "query 512 x 2.5 M len 512 (8-word asm scan, round 5)"),
    ("nohint_neardup90", "C3 shape, 90 % of the candidates near-duplicates of the query, no hint"),
    ("hint16_neardup90", "the same, `score_hint = 16` (band pass + dense re-scan of the unresolved, round 5)"),
    ("hint16_neardup99", "99 % near-duplicates, `score_hint = 16`"),
    ("hint16_neardup50", "50 % near-duplicates, `score_hint = 16` (the hint is wrong for half the corpus: slower than no hint)"),
    ("q128_levenshtein", "query 128 x 20 M len 128 (2-word asm scan)"),
    ("c3_cutoff8", "C3 corpus, `score_cutoff = 8`"),
    ("c3_cutoff8_neardup1", "the same, 1 % of the candidates near the query (tiles with a lane or two left are handed to the dense second pass)"),
    ("c3_cutoff8_neardup50", "the same, 50 % near the query"),
    ("c4_indel", "C4 Indel (asm scan over the 6-bit payload, round 5)"),
    ("q32_indel", "same corpus, Indel, query 32 (32-bit words over the 6-bit payload, round 5)"),
    ("c4_lcs_seq", "C4 LCS"),
    ("c4_jaro", "C4 Jaro (f64 out)"),
    ("c4_jaro_winkler", "C4 Jaro-Winkler (f64 out)"),
    ("osa", "OSA"),
    ("c5_cutoff3_many", "C5 shape, 100 M: `score_cutoff = 3`, one u32 per candidate"),
    ("c5_cutoff3_topk", "C5 shape, 100 M: `score_cutoff = 3`, top-16 only"),
    ("filter_cutoff3", "the same, compact (index, score) pairs only: `rf_filter_u32`, one host synchronization per call inside the step (`--mode filter`, round 6)"),
    ("survivors1_cutoff3_many", "C5 shape, 1 % of the candidates carry the query's first 8..12 symbols (`--head-share 0.01`): one u32 per candidate (lane compaction, round 6; round 5's path: 137)"),
    ("survivors1_cutoff3_topk", "the same, top-16 only (round 5's path: 143)"),
    ("survivors1_cutoff3_filter", "the same, compact pairs"),
    ("survivors5_cutoff3_many", "5 % prefix sharers, one u32 per candidate (round 5's path: 80)"),
    ("zipf_cutoff3_many", "Zipf(1.1) symbols instead of uniform ones, `score_cutoff = 3`, one u32 per candidate"),
    ("c5_1B_world1", "C5 = BASELINE configs[4] at N = 1: 1 B candidates, cutoff 3, top-16 + RCCL gather + merge per step (`--config c5`)"),
    ("topk16_nocutoff", "top-16, no cutoff, no per-candidate output"),
    ("sharded_path_world1", "the sharded step at world size 1 (top-16 + per-candidate distances + all-gather + merge)"),
    ("multi4_levenshtein", "4 queries x C2 corpus, Levenshtein (`--queries 4`)"),
    ("multi4_indel", "4 queries x C2 corpus, Indel"),
    ("indel_cutoff12", "Indel, `score_cutoff = 12` (a 0.9 `fuzz::ratio` threshold)"),
    ("osa_cutoff3", "OSA, `score_cutoff = 3`"),
    ("cutoff5_many", "Levenshtein, `score_cutoff = 5` (the first look as a streaming pass over the head plane)"),
    ("jw_cutoff0.9", "Jaro-Winkler, `score_cutoff = 0.9`"),
    ("jw_filter0.9", "the same, compact pairs (`rf_filter_f64`)"),
    ("wf_weights_1_2_3", "Levenshtein weights (1,2,3), 20 M candidates (`wf_reg_kernel<64>`)"),
]


def table():
    out = ["| workload (1 x MI355X, 100 M candidates unless noted) | Gpairs/s | moved GB/s (frac of 8 TB/s) | SURVEY 8(d) GB/s (frac) | issue ceiling Gpairs/s (kernel / ceiling) | CPU oracle, 1 thread | parity |",
           "|---|---|---|---|---|---|---|"]
    for stem, label in ROWS:
        p = os.path.join(ROOT, "profiles", f"bench_{stem}.json")
        try:
            d = json.load(open(p))
        except (OSError, ValueError):
            continue
        r = d["roofline"]
        s8 = r.get("survey_8d", {"achieved": r["achieved"], "frac": r["frac"]})
        ib = r.get("issue_bound")
        ceil = f'{ib["ceiling"]:.1f} ({ib["frac"]:.2f})' if ib and ib["ceiling"] < 1e4 else "-"
        if ib and "asm_column" in ib:  # (the 6-bit LCS scans: the asm column's own ceiling at the clock sampled under the scan, not the compiled 8-bit column's)
            ceil = f'{ib["asm_column"]["ceiling_at_scan_clock"]:.1f} ({ib["asm_column"]["frac"]:.2f})'
        cpu = d.get("cpu_baseline")
        par = d.get("parity")
        out.append(f'| {label} | {d["value"]:.2f} | {r["achieved"]:.0f} ({r["frac"]:.3f}) | {s8["achieved"]:.0f} ({s8["frac"]:.3f}) | {ceil} | '
                   f'{cpu["value"]:.4f} Gpairs/s' + (f' ({d["value"] / cpu["value"]:.0f}x)' if cpu else "") if cpu else
                   f'| {label} | {d["value"]:.2f} | {r["achieved"]:.0f} ({r["frac"]:.3f}) | {s8["achieved"]:.0f} ({s8["frac"]:.3f}) | {ceil} | -')
        out[-1] += f' | {par["mismatches"]} / {par["checked"]} |' if par else " | tests |"
    return "\n".join(out)


if __name__ == "__main__":
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    b, e = "<!-- bench-table:begin -->", "<!-- bench-table:end -->"
    new = s[: s.index(b) + len(b)] + "\n" + table() + "\n" + s[s.index(e):] if b in s and e in s else None
    if "--check" in sys.argv:
        sys.exit(0 if new is not None and new == s else 1)
    if new is None:
        print(table())
    else:
        open(path, "w").write(new)
        print("DESIGN.md table refreshed")
