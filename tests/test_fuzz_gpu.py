"""A short fixed-seed pass of the parity fuzzer (tests/fuzz_parity.py): random shapes, batch sizes, image kinds and pipeline
options, every output against the CPU oracle.  The long passes are run by hand (numbers in the fuzzer's header)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("options,seed,lo,hi", [("0", 21, 6, 90), ("1", 22, 16, 160)])
def test_fuzzed_cases_match_the_oracle(options, seed, lo, hi):
    env = dict(os.environ, FUZZ_N="14", FUZZ_SEED=str(seed), FUZZ_MIN=str(lo), FUZZ_MAX=str(hi), FUZZ_OPTIONS=options,
               GRAFT_REPO_ROOT=str(ROOT))
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "fuzz_parity.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "FUZZ DONE: mismatching cases 0" in r.stdout, r.stdout[-3000:]
    assert "RAISED" not in r.stdout, r.stdout[-3000:]


def test_fuzzed_grabcut_class_matches_the_oracle():
    env = dict(os.environ, FUZZ_KIND="grabcut", FUZZ_N="40", FUZZ_SEED="23", GRAFT_REPO_ROOT=str(ROOT))
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "fuzz_parity.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "FUZZ DONE: mismatching cases 0" in r.stdout, r.stdout[-3000:]
