"""In-launch hand-offs under repetition: tools/soak.py (split-K reduction and one-launch decode attention, fixed inputs,
alternating shapes / chunk counts / streams, cache-churning work in between) must give bit-identical results on every
launch and leave its tickets at zero.  A short run here; the long one (250 000 launches each) is recorded in DESIGN.md."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_handoffs_are_deterministic_under_repetition():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "3000"], capture_output=True, text=True,
                         timeout=900)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert "split-K: 3000 launches, 0 mismatches" in res.stdout
    assert "decode attention: 3000 launches, 0 mismatches; tickets zero: True" in res.stdout
