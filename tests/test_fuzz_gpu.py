"""Random sets x segment lengths x models x scan variants through the C-ABI against the oracle (scripts/fuzz_parity.py,
which runs for as long as asked; here for half a minute with a fixed seed -- some 400 cases)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, needs_hooks

pytestmark = pytest.mark.gpu


def test_random_cases_against_the_oracle():
    needs_hooks("ROUTE_TINY", "1")  # (the fuzzers draw scan variants from the test hooks)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "30", "4711"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-5:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "all equal to the oracle" in tail, tail
    cases = int(tail.split("fuzz:")[1].split("cases")[0])
    assert cases >= 50, tail


def test_random_routed_large_calls():
    """A minute of scripts/fuzz_large.py: calls of the size at which pass A is routed per pair (10-14 genomes of 3-5 Mbp:
    star, tree, structured, close, joined, both strands, mixed) -- the call as it comes, the lane scan and the forced
    wavefront kernel agree bit for bit and a sampled subject row equals the oracle's."""
    needs_hooks("POOL_MATCH", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_large.py"), "60", "4712"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-5:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "all equal" in tail, tail
    assert int(tail.split("fuzz_large:")[1].split("cases")[0]) >= 3, tail
