"""Pins the CPU oracle (oracle/) against EVERY known-answer test the reference's own test modules and
doctests hold for the one-vs-many path.  Each case goes through the reference's 4-way
helper (free fn both argument orders + BatchComparator both orders, e.g. levenshtein.rs:1847-1875), so
the BatchComparator path -- the one the GPU replaces -- is pinned by every vector.

The unicode vectors (levenshtein.rs:2163-2169 etc.) are run on bytes after an injective char->byte
renaming: every metric here depends only on the equality pattern of the two strings.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as o

USIZE_MAX = 2**64 - 1


# ---------------------------------------------------------------- helpers (the reference's test helpers)
def _four_way(mod, op, s1, s2, tol=None, **kw):
    f = getattr(mod, op)
    r1 = f(s1, s2, **kw)
    r2 = f(s2, s1, **kw)
    r3 = getattr(mod.BatchComparator(s1), op)(s2, **kw)
    r4 = getattr(mod.BatchComparator(s2), op)(s1, **kw)
    for r in (r2, r3, r4):
        if tol is None:
            assert r == r1, (op, s1, s2, kw, r1, r2, r3, r4)
        else:
            assert (r is None) == (r1 is None), (op, s1, s2, kw, r1, r2, r3, r4)
            if r1 is not None:
                assert abs(r - r1) <= tol, (op, s1, s2, kw, r1, r2, r3, r4)
    return r1


def _rename(*strings):
    """Injective char -> byte renaming shared by all strings (for the `.chars()` unicode vectors)."""
    table = {}
    out = []
    for s in strings:
        bs = bytearray()
        for ch in s:
            if ch not in table:
                table[ch] = len(table) + 1
            bs.append(table[ch])
        out.append(bytes(bs))
    assert len(table) < 256
    return out


def _approx(expected, got, tol=1e-4):
    assert (expected is None) == (got is None), (expected, got)
    if expected is not None:
        assert abs(expected - got) <= tol, (expected, got)


W112 = (1, 1, 2)

# ---------------------------------------------------------------- levenshtein.rs:1933-2169 + doctests
LEV_SIMPLE = [("aaaa", "aaaa", 0, 1.0), ("aaaa", "aaa", 1, 0.75), ("aaaa", "aaab", 1, 0.75), ("abaa", "baaa", 2, 0.5), ("aaaa", "bbbb", 4, 0.0)]


def test_lev_empty():  # levenshtein.rs:1933-1937
    assert _four_way(o.levenshtein, "distance", "", "") == 0
    assert _four_way(o.levenshtein, "distance", "aaaa", "") == 4


@pytest.mark.parametrize("a,b,dist,nsim", LEV_SIMPLE)
def test_lev_simple(a, b, dist, nsim):  # levenshtein.rs:1940-1977
    assert _four_way(o.levenshtein, "distance", a, b) == dist
    _approx(nsim, _four_way(o.levenshtein, "normalized_similarity", a, b, tol=1e-4, score_cutoff=0.0))


@pytest.mark.parametrize("a,b,dist,nsim", [("aaaa", "aaaa", 0, 1.0), ("aaaa", "aaa", 1, 0.8571), ("abaa", "baaa", 2, 0.75), ("aaaa", "aaab", 2, 0.75), ("aaaa", "bbbb", 8, 0.0)])
def test_lev_weighted_simple(a, b, dist, nsim):  # levenshtein.rs:1980-2020
    assert _four_way(o.levenshtein, "distance", a, b, weights=W112) == dist
    _approx(nsim, _four_way(o.levenshtein, "normalized_similarity", a, b, tol=1e-4, weights=W112, score_cutoff=0.0))


def test_lev_mbleven():  # levenshtein.rs:2023-2066
    a, b = "South Korea", "North Korea"
    d = lambda **kw: _four_way(o.levenshtein, "distance", a, b, **kw)
    assert d() == 2
    assert [d(score_cutoff=k) for k in (4, 3, 2, 1, 0)] == [2, 2, 2, None, None]
    assert d(weights=W112) == 4
    assert [d(weights=W112, score_cutoff=k) for k in (4, 3, 2, 1)] == [4, None, None, None]
    a, b = "aabc", "cccd"
    assert d() == 4
    assert [d(score_cutoff=k) for k in (4, 3, 2, 1, 0)] == [4, None, None, None, None]
    assert d(weights=W112) == 6
    assert [d(weights=W112, score_cutoff=k) for k in (6, 5, 4, 3, 2, 1, 0)] == [6] + [None] * 6
    # the BatchComparator really lands in mbleven for cutoff < 4 (levenshtein.rs:1094-1101)
    o.levenshtein.BatchComparator("South Korea").distance("North Korea", score_cutoff=3)
    assert o.last_lev_path() == "mbleven"


BANDED = [  # levenshtein.rs:2069-2130
    ("kkkkbbbbfkkkkkkibfkkkafakkfekgkkkkkkkkkkbdbbddddddddddafkkkekkkhkk", "khddddddddkkkkdgkdikkccccckcckkkekkkkdddddddddddafkkhckkkkkdckkkcc", 36, {31: None}),
    ("ccddcddddddddddddddddddddddddddddddddddddddddddddddddddddaaaaaaaaaaa", "aaaaaaaaaaaaaadddddddddbddddddddddddddddddddddddddddddddddbddddddddd", 26, {31: 26}),
    (
        "accccccccccaaaaaaaccccccccccccccccccccccccccccccacccccccccccccccccccccccccccccc" "ccccccccccccccccccccaaaaaaaaaaaaacccccccccccccccccccccc",
        "ccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccc" "ccccccccccccccccccccccccccccccccccccbcccb",
        24,
        {25: 24},
    ),
    (
        "miiiiiiiiiiliiiiiiibghiiaaaaaaaaaaaaaaacccfccccedddaaaaaaaaaaaaaaaaaaaaaaaaaaaa" "aaaaaaaaaaaaa",
        "aaaaaaajaaaaaaaabghiiaaaaaaaaaaaaaaacccfccccedddaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa" "aajjdim",
        27,
        {27: 27},
    ),
    (
        "lllllfllllllllllllllllllllllllllllllllllllllllllllllllglllllilldcaaaaaaaaaaaaaa" "aaaaadbbllllllllllhllllllllllllllllllllllllllgl",
        "aaaaaaaaaaaaaadbbllllllllllllllelllllllllllllllllllllllllllllllglllllilldcaaaaa" "aaaaaaaaaaaaaadbbllllllllllllllellllllllllllllhlllllllllill",
        23,
        {27: 23, 28: 23},
    ),
    ("llccacaaaaaaaaaccccccccccccccccddffaccccaccecccggggclallhcccccljif", "bddcbllllllbcccccccccccccccccddffccccccccebcccggggclbllhcccccljifbddcccccc", 27, {27: 27, 28: 27}),
]


@pytest.mark.parametrize("s1,s2,dist,cut", BANDED)
def test_lev_banded(s1, s2, dist, cut):
    assert _four_way(o.levenshtein, "distance", s1, s2) == dist
    for k, exp in cut.items():
        assert _four_way(o.levenshtein, "distance", s1, s2, score_cutoff=k) == exp


def test_lev_banded_hits_small_band_and_block_paths():
    """test_banded is path-targeted upstream: make sure the BatchComparator side really reaches
    hyrroe2003_small_band_with_pm (len1 > 64, 2k+1 <= 64) and hyrroe2003_block."""
    s1, s2, _, _ = BANDED[2]
    assert len(s1) > 64
    o.levenshtein.BatchComparator(s1).distance(s2, score_cutoff=25)
    assert o.last_lev_path() == "small_band"
    o.levenshtein.BatchComparator(s1).distance(s2)
    assert o.last_lev_path() == "block"
    s1, s2, _, _ = BANDED[0]
    assert len(s1) > 64 and 2 * 31 + 1 <= 64
    o.levenshtein.BatchComparator(s1).distance(s2, score_cutoff=31)
    assert o.last_lev_path() == "small_band"


def test_lev_blockwise():  # levenshtein.rs:2132-2137
    assert _four_way(o.levenshtein, "distance", "a" * 128, "b" * 128) == 128


def test_lev_large_band(golden_dir):  # levenshtein.rs:2139-2161 (free function only upstream; we add the batch leg)
    e1 = open(os.path.join(golden_dir, "ocr_example1.bin"), "rb").read()
    e2 = open(os.path.join(golden_dir, "ocr_example2.bin"), "rb").read()
    assert (len(e1), len(e2)) == (106514, 107244)
    assert o.levenshtein.distance(e1, e2) == 5278
    assert o.levenshtein.distance(e1, e2, score_cutoff=2500) is None
    assert o.levenshtein.distance(e1, e2, score_hint=0) == 5278
    bc = o.levenshtein.BatchComparator(e1)
    assert bc.distance(e2) == 5278
    assert bc.distance(e2, score_cutoff=2500) is None
    assert bc.distance(e2, score_hint=0) == 5278
    assert o.last_lev_path() == "block"


def test_lev_unicode():  # levenshtein.rs:2163-2169
    a, b = _rename("Иванко", "Петрунко")
    assert _four_way(o.levenshtein, "distance", a, b) == 5


def test_lev_doctests():  # src/lib.rs:32-71, levenshtein.rs:1378,1633
    assert o.levenshtein.distance("kitten", "sitting") == 3
    assert o.levenshtein.distance("kitten", "sitting", score_cutoff=2) is None
    assert o.levenshtein.distance("kitten", "sitting", score_hint=2) == 3
    assert o.levenshtein.BatchComparator("kitten").distance("kitten") == 0
    assert o.levenshtein.BatchComparator("kitten").distance("sitting") == 3
    assert o.levenshtein.distance("CA", "ABC") == 3
    assert o.levenshtein.BatchComparator("CA").distance("ABC") == 3


# ---------------------------------------------------------------- lcs_seq.rs:1139-1266 + doctests
def test_lcs_similar_and_different():
    assert _four_way(o.lcs_seq, "distance", "a", "a") == 0
    assert _four_way(o.lcs_seq, "distance", "aaaa", "aaaa") == 0
    assert _four_way(o.lcs_seq, "similarity", "aaaa", "aaaa") == 4
    _approx(0.0, _four_way(o.lcs_seq, "normalized_distance", "aaaa", "aaaa", tol=1e-4, score_cutoff=1.0))
    _approx(1.0, _four_way(o.lcs_seq, "normalized_similarity", "aaaa", "aaaa", tol=1e-4, score_cutoff=0.0))
    assert _four_way(o.lcs_seq, "distance", "aaaa", "bbbb") == 4
    assert _four_way(o.lcs_seq, "similarity", "aaaa", "bbbb") == 0
    _approx(1.0, _four_way(o.lcs_seq, "normalized_distance", "aaaa", "bbbb", tol=1e-4, score_cutoff=1.0))
    _approx(0.0, _four_way(o.lcs_seq, "normalized_similarity", "aaaa", "bbbb", tol=1e-4, score_cutoff=0.0))


def test_lcs_mbleven():  # lcs_seq.rs:1184-1245
    a, b = "South Korea", "North Korea"
    sim = lambda **kw: _four_way(o.lcs_seq, "similarity", a, b, **kw)
    dist = lambda **kw: _four_way(o.lcs_seq, "distance", a, b, **kw)
    assert sim() == 9 and sim(score_cutoff=9) == 9 and sim(score_cutoff=10) is None
    assert dist() == 2
    assert [dist(score_cutoff=k) for k in (4, 3, 2, 1, 0)] == [2, 2, 2, None, None]
    a, b = "aabc", "cccd"
    assert sim() == 1 and sim(score_cutoff=1) == 1 and sim(score_cutoff=2) is None
    assert dist() == 3
    assert [dist(score_cutoff=k) for k in (4, 3, 2, 1, 0)] == [3, 3, None, None, None]


def test_lcs_misc():
    assert _four_way(o.lcs_seq, "similarity", "001", "220") == 1  # test_cached lcs_seq.rs:1247-1252
    a, b = _rename("Иванко", "Петрунко")
    assert _four_way(o.lcs_seq, "distance", a, b) == 5  # lcs_seq.rs:1253-1259
    assert _four_way(o.lcs_seq, "distance", "ab", "ac") == 1  # fuzzing_regressions :1260-1266
    assert o.lcs_seq.distance("lewenstein", "levenshtein") == 2  # doctests :581,630,764
    assert o.lcs_seq.similarity("lewenstein", "levenshtein") == 9
    assert o.lcs_seq.BatchComparator("lewenstein").similarity("levenshtein") == 9


# ---------------------------------------------------------------- indel.rs:710-864 + doctests
def test_indel_similar_and_different():
    assert _four_way(o.indel, "distance", "aaaa", "aaaa") == 0
    assert _four_way(o.indel, "similarity", "aaaa", "aaaa") == 8
    _approx(0.0, _four_way(o.indel, "normalized_distance", "aaaa", "aaaa", tol=1e-4, score_cutoff=1.0))
    _approx(1.0, _four_way(o.indel, "normalized_similarity", "aaaa", "aaaa", tol=1e-4, score_cutoff=0.0))
    assert _four_way(o.indel, "distance", "aaaa", "bbbb") == 8
    assert _four_way(o.indel, "similarity", "aaaa", "bbbb") == 0
    _approx(1.0, _four_way(o.indel, "normalized_distance", "aaaa", "bbbb", tol=1e-4, score_cutoff=1.0))
    _approx(0.0, _four_way(o.indel, "normalized_similarity", "aaaa", "bbbb", tol=1e-4, score_cutoff=0.0))


def test_indel_mbleven():  # indel.rs:741-803
    d = lambda a, b, **kw: _four_way(o.indel, "distance", a, b, **kw)
    a, b = "South Korea", "North Korea"
    assert d(a, b) == 4
    assert [d(a, b, score_cutoff=k) for k in (5, 4, 3, 2, 1, 0)] == [4, 4, None, None, None, None]
    a, b = "aabc", "cccd"
    assert d(a, b) == 6
    assert [d(a, b, score_cutoff=k) for k in (6, 5, 4, 3, 2, 1, 0)] == [6] + [None] * 6


def test_indel_issue_unknown():  # indel.rs:806-816
    _approx(0.3333333, _four_way(o.indel, "normalized_similarity", "001", "220", tol=1e-4, score_cutoff=0.0))


INDEL_LONG_S2 = (
    "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa"
    "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaacca"
    "cccaccaaaaaaaadaaaaaaaaccccaccccccaaaaaaaccccaaacccaccccadddaaaaaaaaaaaaaaaaa"
    "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaccccccccacccaaaaaacccaaaaaacc"
    "cacccaaaaaacccdccccccaccccccccccccccccccccccccccccccccccccccccccccccccccccccc"
    "ccccccddddddaaaaaaaaaaaaaaaaaaaaaaaaaacacccaaaaaacccddddaaaaaaaaaaaaaaaaaaaaa"
    "aaaaaaaaccccaaaaaaaaaaccccccaadddaaaaaaaaaaaaaaaaaaaaaacaaaaaa"
)


def test_indel_banded_implementation():  # indel.rs:818-848
    s1 = "ddccbccc"
    d = lambda **kw: _four_way(o.indel, "distance", s1, INDEL_LONG_S2, **kw)
    assert d() == 508
    assert d(score_cutoff=508) == 508
    assert d(score_cutoff=507) is None
    assert d(score_cutoff=USIZE_MAX) == 508
    s1b = "bbbdbbmbbbbbbbbbBbfbbbbbbbbbbbbbbbbbbbrbbbbbrbbbbbdbnbbbjbhbbbbbbbbbhbbb" "bbCbobbbxbbbbbkbbbAbxbbwbbbtbcbbbbebbiblbbbbqbbbbbbpbbbbbbubbbkbbDbbbhbkbC" "bbgbbrbbbbbbbbbbbkbyvbbsbAbbbbz"
    s2b = "jaaagaaqyaaaanrCfwaaxaeahtaaaCzaaaspaaBkvaaaaqDaacndaaeolwiaaauaaaaaaamA"
    assert _four_way(o.indel, "distance", s1b, s2b) == 231


def test_indel_misc():
    a, b = _rename("Иванко", "Петрунко")
    assert _four_way(o.indel, "distance", a, b) == 8  # indel.rs:850-856
    assert _four_way(o.indel, "distance", "ab", "ac") == 2  # :858-864
    assert o.indel.distance("lewenstein", "levenshtein") == 3  # doctests :119-122, :320
    assert o.indel.distance("lewenstein", "levenshtein", score_cutoff=2) is None
    assert o.indel.BatchComparator("lewenstein").distance("levenshtein") == 3


# ---------------------------------------------------------------- jaro.rs:1080-1218, jaro_winkler.rs:676-809
def _load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def test_jaro_no_cutoff():  # jaro.rs:1080-1092
    _approx(0.455556, _four_way(o.jaro, "similarity", "james", "robert", tol=1e-4, score_cutoff=0.0))
    _approx(1.0 - 0.455556, _four_way(o.jaro, "distance", "james", "robert", tol=1e-4, score_cutoff=1.0))


def test_jaro_flag_chars_table(golden_dir):  # jaro.rs:1094-1189: 20 x 20 names x 12 cutoffs
    t = _load(golden_dir, "jaro_table.json")
    names, scores = t["names"], t["scores"]
    assert len(names) == 20
    for cutoff in [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1]:
        for i, n1 in enumerate(names):
            for j, n2 in enumerate(names):
                score = scores[i * len(names) + j]
                exp_sim = score if cutoff <= score else None
                exp_dist = None if exp_sim is None else 1.0 - exp_sim
                _approx(exp_sim, _four_way(o.jaro, "similarity", n1, n2, tol=1e-4, score_cutoff=cutoff))
                _approx(exp_dist, _four_way(o.jaro, "distance", n1, n2, tol=1e-4, score_cutoff=1.0 - cutoff))


def test_jaro_unicode_and_fuzz_regression():
    a, b = _rename("Иванко", "Петрунко")
    _approx(0.375, _four_way(o.jaro, "distance", a, b, tol=1e-4, score_cutoff=1.0))  # jaro.rs:1191-1199
    # jaro.rs:1201-1218: > 64 chars on both sides => block path; tolerance 0.32144 upstream
    s1 = (
        "afddddddddddddddddddddddddddddddddddddddddadacccccccdddddddddd%,ccaa{1}ccccdccccccccccccccccccccc"
        "cccccccccccccccccccccccccccccccccccccccccccccccczcecccccccccccccccccccccccccccccccccccccccccccccc"
        "cccccccccdddddddd디ccc디Gcddddccccccccccccccccccccccccccccccccccccccccccccccccccccccaccccccccccccc"
        "ccccccccccccccccccccccccccccccccccccccccccccea,ccccccccccccccccccccccccccccccccccccccc"
    )
    s2 = "ccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccddddd" "dddddddddddddddddddddddddddddf,ccczюec*ceч;e,"
    a, b = _rename(s1, s2)
    got = _four_way(o.jaro, "distance", a, b, tol=1e-4, score_cutoff=1.0)
    assert got is not None and abs(got - 0.1) <= 0.32144


def test_jaro_winkler_no_cutoff():  # jaro_winkler.rs:676-691
    _approx(0.455556, _four_way(o.jaro_winkler, "similarity", "james", "robert", tol=1e-4, score_cutoff=0.0))
    _approx(1.0 - 0.455556, _four_way(o.jaro_winkler, "distance", "james", "robert", tol=1e-4, score_cutoff=1.0))


def test_jaro_winkler_table(golden_dir):  # jaro_winkler.rs:693-798: 22 x 22, cutoff 0.0 only upstream
    t = _load(golden_dir, "jaro_winkler_table.json")
    names, scores = t["names"], t["scores"]
    assert len(names) == 22
    for i, n1 in enumerate(names):
        for j, n2 in enumerate(names):
            score = scores[i * len(names) + j]
            _approx(score, _four_way(o.jaro_winkler, "similarity", n1, n2, tol=1e-4, score_cutoff=0.0))
            _approx(1.0 - score, _four_way(o.jaro_winkler, "distance", n1, n2, tol=1e-4, score_cutoff=1.0))
    k = names.index("aaaaaaaa"), names.index("aabaaab")
    assert abs(scores[k[0] * 22 + k[1]] - 0.82381) < 1e-9  # the cell that pins the Winkler boost


def test_jaro_winkler_unicode():  # jaro_winkler.rs:800-808
    a, b = _rename("Иванко", "Петрунко")
    _approx(0.375, _four_way(o.jaro_winkler, "distance", a, b, tol=1e-4, score_cutoff=1.0))


# ---------------------------------------------------------------- osa.rs:618-693 ("next" row f3 of SURVEY 8)
def test_osa_simple_and_unicode():
    d = lambda a, b, **kw: _four_way(o.osa, "distance", a, b, **kw)
    big = 2**64 - 1  # the upstream helper passes score_cutoff = usize::MAX
    assert d("", "", score_cutoff=big) == 0
    assert d("aaaa", "", score_cutoff=big) == 4
    assert d("aaaa", "", score_cutoff=1) is None
    assert d("CA", "ABC", score_cutoff=big) == 3
    assert d("CA", "AC", score_cutoff=big) == 1
    filler = "a" * 64
    s1, s2 = "a" + filler + "CA" + filler + "a", "b" + filler + "AC" + filler + "b"
    assert d(s1, s2, score_cutoff=big) == 3  # > 64 symbols: hyrroe2003_block with the cross-word transposition term
    a, b = _rename("Иванко", "Петрунко")
    assert d(a, b) == 5


# ---------------------------------------------------------------- fuzz.rs:186-301
def test_fuzz_ratio():
    s1, s3 = "new york mets", "the wonderful new york mets"
    for s in (s1, "test", "{", "{a"):
        _approx(1.0, o.fuzz.ratio(s, s))  # test_equal
    _approx(0.65, o.fuzz.ratio(s1, s3))  # test_partial_ratio :205-213
    _approx(1.0, o.fuzz.ratio("", ""))
    _approx(0.0, o.fuzz.ratio("test", ""))
    _approx(0.0, o.fuzz.ratio("", "test"))
    for a, b in (("South Korea", "North Korea"), ("bc", "bca")):  # issue206 / issue210 :247-301
        score = o.fuzz.ratio(a, b)
        assert o.fuzz.ratio(a, b, score_cutoff=score + 0.0001) is None
        _approx(score, o.fuzz.ratio(a, b, score_cutoff=score - 0.0001))


def test_fuzz_ratio_batch_quirk_q1():
    """src/fuzz.rs:141 normalises through the inner lcs_seq comparator: LCS / max(len)."""
    a, b = "this is a test", "this is a test!"
    assert abs(o.fuzz.ratio(a, b) - 28 / 29) < 1e-12  # 0.9655 as the doc comment says
    assert abs(o.fuzz.RatioBatchComparator(a).similarity(b) - 14 / 15) < 1e-12


# ---------------------------------------------------------------- PM table layout (pattern_match_vector.rs:213-224)
def test_pm_layout():
    q = bytes((i * 7 + 3) % 251 for i in range(150))
    pm = o.levenshtein.BatchComparator(q).pm()
    assert pm.shape == (256, 3)
    exp = np.zeros((256, 3), dtype=np.uint64)
    for i, c in enumerate(q):
        exp[c, i // 64] |= np.uint64(1) << np.uint64(i % 64)
    assert (pm == exp).all()


def test_gpu_selfcheck_fixture_is_what_the_oracle_says():
    """tests/golden/gpu_selfcheck.json (the oracle-free GPU check of smoke()) must equal the oracle's output today."""
    import json
    import os

    import numpy as np

    from rapidfuzz_rs_amd import _native as N

    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpu_selfcheck.json")))
    q = fx["query"].encode("latin-1")
    cands = [c.encode("latin-1") for c in fx["candidates"]]
    data = np.frombuffer(b"".join(cands), dtype=np.uint8)
    offsets = np.zeros(len(cands) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(c) for c in cands])
    ops = {"distance": N.OP_DISTANCE, "similarity": N.OP_SIMILARITY, "normalized_similarity": N.OP_NORMALIZED_SIMILARITY}
    for key, exp in fx["expected"].items():
        metric, op, cutoff = key.split(":")
        cut = None if cutoff == "None" else (float(cutoff) if "." in cutoff else int(cutoff))
        got = getattr(o, metric).BatchComparator(q).many(ops[op], data, offsets, score_cutoff=cut)
        for g, e in zip(got.tolist(), exp):
            if got.dtype == np.float64:
                assert (e is None and np.isnan(g)) or (e is not None and g == float.fromhex(e)), key
            else:
                assert (e is None and g == 2**64 - 1) or (e is not None and g == e), key
