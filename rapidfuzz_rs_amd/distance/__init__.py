"""`rapidfuzz::distance` (src/distance.rs:1-10): the metric modules that have a bit-parallel batch path."""
from .. import _native as _N
from .._comparator import MetricModule as _M

levenshtein = _M("levenshtein", _N.LEVENSHTEIN, False)    # src/distance/levenshtein.rs
indel = _M("indel", _N.INDEL, False)                      # src/distance/indel.rs
lcs_seq = _M("lcs_seq", _N.LCS_SEQ, False)                # src/distance/lcs_seq.rs
osa = _M("osa", _N.OSA, False)                            # src/distance/osa.rs
jaro = _M("jaro", _N.JARO, True)                          # src/distance/jaro.rs
jaro_winkler = _M("jaro_winkler", _N.JARO_WINKLER, True)  # src/distance/jaro_winkler.rs

__all__ = ["levenshtein", "indel", "lcs_seq", "osa", "jaro", "jaro_winkler"]
