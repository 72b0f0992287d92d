"""Randomised end-to-end parity (tools/stress.py): random small configurations against the oracle."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT


@pytest.mark.gpu
def test_random_configurations_against_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress.py"), "16", "5000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "16 configurations, 0 failures" in r.stdout
