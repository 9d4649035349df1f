"""Random sets x segment lengths x models x scan variants through the C-ABI against the oracle (scripts/fuzz_parity.py,
which runs for as long as asked; here for half a minute with a fixed seed -- some 400 cases)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_cases_against_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "30", "4711"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-5:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "all equal to the oracle" in tail, tail
    cases = int(tail.split("fuzz:")[1].split("cases")[0])
    assert cases >= 50, tail
