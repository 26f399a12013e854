"""Soak of the one-launch split composite (round 6): the phase-2 workers of k_render<.., SPLIT> wait — bounded — on device-coherent words
written by other workgroups of the same launch; a missed hand-off would show as a differing image, a non-zero error word
(RasterState.split_errors) or leftover items.  300 training steps on the 1 M surface-like cloud, interleaved with the encoder-like cloud
(with and without the SPLIT kernel instances), every split step bit for bit the first one (scripts/soak_split.py; 20 000 steps were run
by hand in round 6: 0 differing)."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def test_split_forward_and_backward_are_bit_stable_over_300_steps(gpu):
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "scripts" / "soak_split.py"), "300"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert " 0 differed" in r.stdout
