"""CPU: the current-state design document stays checkable: at most 150 lines, and the files it and the
README point the reader to exist."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_is_at_most_150_lines_and_its_pointers_exist():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert len(text.rstrip("\n").split("\n")) <= 150
    for name in ("docs/HISTORY.md", "INTEGRATION.md", "profiles/r05/README.md", "profiles/r05/lib_sha256.txt", "profiles/pmc_sq.json",
                 "profiles/pmc_traffic.json", "include/auv_hip.h", "oracle/auv_oracle.c", "tools/fma_issue_bench.hip", "tests/test_gpu_multi.py",
                 "tests/test_gpu_fresh.py"):
        assert os.path.exists(os.path.join(ROOT, name)), name
    # every profiles/r05 file DESIGN.md names by its full name is there (bare names are looked up under profiles/r05)
    for m in re.findall(r"`((?:profiles/r0[45]/)?[a-z0-9][a-z0-9_]+\.(?:jsonl|json|log|txt))`", text):   # (not the `_sub1.json` shorthands)
        path = m if m.startswith("profiles/") else os.path.join("profiles", "r05", m)
        assert os.path.exists(os.path.join(ROOT, path)), m
