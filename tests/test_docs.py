"""The measurement table of DESIGN.md section 6 is generated from the committed bench lines (profiles/bench_*.json): it must be
current, so the prose cannot drift from the evidence."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_bench_table_is_generated_from_profiles():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), "--check"])
    assert r.returncode == 0, "DESIGN.md section 6 is stale: run python tools/design_table.py"


def test_every_profile_the_design_cites_exists():
    import re

    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    for name in set(re.findall(r"profiles/([A-Za-z0-9_.]+\.(?:txt|json))", text)):
        assert os.path.exists(os.path.join(ROOT, "profiles", name)), name
