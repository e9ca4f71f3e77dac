"""The reference's own known-answer tests (tests/test_oracle_known_answers.py, i.e. SURVEY App. B) run a second
time with the DEVICE behind the same names: `o.levenshtein.distance(...)`, `o.levenshtein.BatchComparator(a).distance(b)`,
`o.fuzz.ratio(...)` resolve to rapidfuzz_rs_amd instead of the oracle, so every vector the reference holds for this path --
including its 4-way helper (free function and BatchComparator, both argument orders) -- goes through the HIP kernels.

Left out: the two oracle-internal tests (which CPU kernel a call ended in; the PM table layout, covered by test_abi.py).
"""
import inspect
import types

import pytest

import rapidfuzz_rs_amd as rf
import test_oracle_known_answers as ka

pytestmark = pytest.mark.gpu

class _AnyPath:
    """`o.last_lev_path()` is oracle instrumentation (which CPU kernel a call ended in); the device has one exact kernel
    per word class, so those assertions are vacuous here."""

    def __eq__(self, other):
        return True


DEVICE = types.SimpleNamespace(
    last_lev_path=lambda: _AnyPath(),
    levenshtein=rf.distance.levenshtein,
    indel=rf.distance.indel,
    lcs_seq=rf.distance.lcs_seq,
    osa=rf.distance.osa,
    jaro=rf.distance.jaro,
    jaro_winkler=rf.distance.jaro_winkler,
    fuzz=rf.fuzz,
)
SKIP = {"test_lev_banded_hits_small_band_and_block_paths", "test_pm_layout",
        "test_gpu_selfcheck_fixture_is_what_the_oracle_says"}  # (oracle-side guard of a fixture: uses the oracle's own batch entry point)


def _cases():
    for name, fn in sorted(vars(ka).items()):
        if not name.startswith("test_") or name in SKIP or not callable(fn):
            continue
        marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
        if not marks:
            yield pytest.param(name, None, id=name)
            continue
        assert len(marks) == 1
        argnames = [a.strip() for a in marks[0].args[0].split(",")]
        for i, values in enumerate(marks[0].args[1]):
            values = values if isinstance(values, (tuple, list)) else (values,)
            yield pytest.param(name, dict(zip(argnames, values)), id=f"{name}[{i}]")


@pytest.mark.parametrize("name,params", list(_cases()))
def test_reference_known_answer_on_device(name, params, monkeypatch, golden_dir):
    monkeypatch.setattr(ka, "o", DEVICE)
    fn = getattr(ka, name)
    kwargs = dict(params or {})
    if "golden_dir" in inspect.signature(fn).parameters:
        kwargs["golden_dir"] = golden_dir
    fn(**kwargs)


def _unicode_cases():
    for p in _cases():
        name = p.values[0]
        if "_rename(" in inspect.getsource(getattr(ka, name)):
            yield p


@pytest.mark.parametrize("name,params", list(_unicode_cases()))
def test_reference_unicode_vectors_as_chars_on_device(name, params, monkeypatch, golden_dir):
    """The `.chars()` vectors once more WITHOUT the char -> byte renaming: the strings go in as they are and take the
    u32-element path (rf_comparator_new_u32 / rf_corpus_pack_u32)."""
    monkeypatch.setattr(ka, "o", DEVICE)
    monkeypatch.setattr(ka, "_rename", lambda *strings: list(strings))
    fn = getattr(ka, name)
    kwargs = dict(params or {})
    if "golden_dir" in inspect.signature(fn).parameters:
        kwargs["golden_dir"] = golden_dir
    fn(**kwargs)
