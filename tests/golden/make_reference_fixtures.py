#!/usr/bin/env python3
"""Extracts DATA (inputs + expected outputs) from the reference's own test modules into fixtures.

Run once in the build container (needs /root/reference); the outputs are committed:
  ocr_example1.bin / ocr_example2.bin   the two byte strings of src/distance/example/ocr.rs:2,5077
                                         (expected Levenshtein distance 5278, levenshtein.rs:2139-2161)
  jaro_table.json                        names + 20x20 expected similarities (jaro.rs:1094-1141)
  jaro_winkler_table.json                names + 22x22 expected similarities (jaro_winkler.rs:693-770)
Only numbers and string literals are extracted -- no reference code.
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


if __name__ == "__main__":
    ocr()
    table("src/distance/jaro.rs", "jaro_table.json")
    table("src/distance/jaro_winkler.rs", "jaro_winkler_table.json")
