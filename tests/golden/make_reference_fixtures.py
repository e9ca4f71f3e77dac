#!/usr/bin/env python3
"""Extracts DATA (inputs + expected outputs) from the reference's own test modules into fixtures.

Run once in the build container (needs /root/reference); the outputs are committed:
  ocr_example1.bin / ocr_example2.bin   the two byte strings of src/distance/example/ocr.rs:2,5077
                                         (expected Levenshtein distance 5278, levenshtein.rs:2139-2161)
  jaro_table.json                        names + 20x20 expected similarities (jaro.rs:1094-1141)
  jaro_winkler_table.json                names + 22x22 expected similarities (jaro_winkler.rs:693-770)
  reference_api_surface.json             the NAMES of the public items of the modules on the batch path (free functions,
                                         BatchComparator methods, Args builders): what the Rust companion crate must offer
Only numbers, string literals and item names are extracted -- no reference code.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def ocr():
    txt = open(os.path.join(REF, "src/distance/example/ocr.rs")).read()
    arrays = re.findall(r"static\s+(OCR_EXAMPLE\d)\s*:\s*\[u8;\s*(\d+)\]\s*=\s*\[(.*?)\];", txt, re.S)
    assert len(arrays) == 2, len(arrays)
    for name, n, body in arrays:
        vals = [int(v) for v in re.findall(r"\d+", body)]
        assert len(vals) == int(n), (name, len(vals), n)
        fn = {"OCR_EXAMPLE1": "ocr_example1.bin", "OCR_EXAMPLE2": "ocr_example2.bin"}[name]
        open(os.path.join(OUT, fn), "wb").write(bytes(vals))
        print(fn, len(vals), "bytes, max value", max(vals))


def table(src, out):
    txt = open(os.path.join(REF, src)).read()
    t = txt[txt.index("fn test_flag_chars"):]
    names_body = re.search(r"let names = \[(.*?)\];", t, re.S).group(1)
    names = re.findall(r'"([^"]*)"', names_body)
    scores_body = re.search(r"let scores = \[(.*?)\];", t, re.S).group(1)
    scores = [float(v) for v in re.findall(r"[0-9]+\.[0-9]+", scores_body)]
    assert len(scores) == len(names) ** 2, (len(scores), len(names))
    json.dump({"source": src, "names": names, "scores": scores}, open(os.path.join(OUT, out), "w"), indent=0)
    print(out, len(names), "names", len(scores), "scores")


def api_surface():
    """names of `pub fn` items: module level, and per `impl` of a public type (levenshtein.rs:1380-1817 and its siblings, fuzz.rs:8-150)"""
    mods = {"distance::levenshtein": "src/distance/levenshtein.rs", "distance::indel": "src/distance/indel.rs", "distance::lcs_seq": "src/distance/lcs_seq.rs",
            "distance::osa": "src/distance/osa.rs", "distance::jaro": "src/distance/jaro.rs", "distance::jaro_winkler": "src/distance/jaro_winkler.rs", "fuzz": "src/fuzz.rs"}
    out = {}
    for mod, src in mods.items():
        txt = open(os.path.join(REF, src)).read()
        txt = txt.split("#[cfg(test)]")[0]
        free, methods = [], {}
        cur, depth, impl_depth = None, 0, None
        for line in txt.splitlines():
            m = re.match(r"\s*impl(?:<[^>]*>)?\s+(?:\w+\s+for\s+)?(\w+)", line)
            if m and depth == 0:
                cur, impl_depth = m.group(1), depth
            f = re.match(r"\s*pub fn (\w+)", line)
            if f:
                if cur is not None and depth > 0:
                    methods.setdefault(cur, []).append(f.group(1))
                elif depth == 0:
                    free.append(f.group(1))
            depth += line.count("{") - line.count("}")
            if cur is not None and depth == 0 and "}" in line:
                cur = None
        out[mod] = {"free_functions": sorted(set(free)), "methods": {k: sorted(set(v)) for k, v in methods.items() if k in ("BatchComparator", "RatioBatchComparator", "Args")}}
    json.dump({"source": "pub fn item names of rapidfuzz-rs v0.5.0, extracted by tests/golden/make_reference_fixtures.py", "modules": out},
              open(os.path.join(OUT, "reference_api_surface.json"), "w"), indent=1, sort_keys=True)
    print("reference_api_surface.json", {k: (len(v["free_functions"]), {t: len(ms) for t, ms in v["methods"].items()}) for k, v in out.items()})


if __name__ == "__main__":
    ocr()
    table("src/distance/jaro.rs", "jaro_table.json")
    table("src/distance/jaro_winkler.rs", "jaro_winkler_table.json")
    api_surface()
