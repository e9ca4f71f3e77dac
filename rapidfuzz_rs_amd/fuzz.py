"""`rapidfuzz::fuzz` (src/fuzz.rs:48-150): `ratio` and `RatioBatchComparator`."""
from __future__ import annotations

from . import _native as N
from ._comparator import Args, BatchComparator


class RatioBatchComparator(BatchComparator):
    """src/fuzz.rs:98-150.  By default reproduces the reference exactly, including that the batch form
    normalises through the inner lcs_seq comparator (fuzz.rs:141: LCS / max(len1, len2)); pass
    `Args().ratio_indel_normalization()` for the documented Indel ratio."""

    METRIC = N.FUZZ_RATIO
    FLOAT = True

    def similarity(self, s2, args=None, **kw):
        return self._one(N.OP_SIMILARITY, s2, args, kw)

    similarity_with_args = similarity

    def similarity_many(self, corpus, args=None, **kw):
        return self.many(N.OP_SIMILARITY, corpus, args, **kw)


def ratio(s1, s2, args=None, **kw):
    """fuzz::ratio_with_args (src/fuzz.rs:60-85): normalized Indel similarity in [0, 1]."""
    from .distance import indel

    return indel.normalized_similarity(s1, s2, args, **kw)


ratio_with_args = ratio
__all__ = ["ratio", "ratio_with_args", "RatioBatchComparator", "Args"]
