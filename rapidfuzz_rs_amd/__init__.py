"""rapidfuzz_rs_amd -- MI355X (gfx950) one-vs-many fuzzy matching behind rapidfuzz-rs' batch API.

    from rapidfuzz_rs_amd import distance, fuzz, Corpus
    scorer = distance.levenshtein.BatchComparator(b"kitten")
    corpus = Corpus.from_list([b"sitting", b"mitten", ...])        # packed once, kept in HBM
    scorer.distance_many(corpus, score_cutoff=3)                    # uint32[n], 0xFFFFFFFF = None

The package is a thin host layer over librfgpu.so (C ABI in include/rfgpu.h, hand-written HIP kernels in
csrc/).  There is no CPU fallback: if the extension cannot be loaded, importing symbols that need it raises.
"""
from . import _native
from . import _native as N  # op / metric / status constants: rf.N.OP_DISTANCE, ...
from ._comparator import Args, BatchComparator, WeightTable
from ._native import RfError, build
from .corpus import Corpus, host_layout, ragged

from . import distance, fuzz  # noqa: E402

__all__ = ["distance", "fuzz", "Corpus", "Args", "WeightTable", "BatchComparator", "RfError", "build", "host_layout", "ragged"]
