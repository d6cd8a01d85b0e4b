"""Randomised differential test (scripts/fuzz_parity.py): random sizes, contents (texture, low contrast, noise, clustered
corners, blocks) and extractor parameters, every keypoint field and descriptor byte against the CPU oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,orient", [(3, "0"), (11, "0"), (21, "1")])
def test_randomised_parity(seed, orient):
    """orient = 1 mixes in the IC-angle orientation mode (MCORB_ORIENT_IC_ANGLE, the reference's dormant IC_Angle)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "40", str(seed)], capture_output=True,
                         text=True, timeout=600, env=dict(os.environ, FUZZ_ORIENT=orient))
    assert out.returncode == 0 and "0 bad" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


@pytest.mark.gpu
def test_randomised_rig_parity():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "12", "5", "rig"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "0 bad" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


@pytest.mark.gpu
def test_soak_create_destroy_and_threads():
    """scripts/soak.py: 40 create/use/destroy cycles, then two threads mixing synchronous and asynchronous calls on four slots."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "x40 ok" in out.stdout and "mixed) ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
