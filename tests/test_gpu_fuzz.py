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


@pytest.mark.gpu
@pytest.mark.parametrize("frames", ["2", "3"])   # 8 images per job: results through host-mapped memory; 12: copied
def test_soak_both_result_paths_with_admission_limit(frames):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak.py"), frames], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MCORB_GPU_JOBS="2"))
    assert out.returncode == 0 and "x40 ok" in out.stdout and "mixed) ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_repeated_and_alternating_images_agree():
    """scripts/stress_consistency.py: a fresh extractor per case, two images alternating A B A B; a result the host read before it had
    landed would show as an empty first result or as the other image's keypoints (the long form of this run found the one-in-10^4
    ordering problem of the first small-batch hand-off)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_consistency.py"), "300", "17"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and " 0 bad" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("slots,gpu_jobs", [("4", "0"), ("5", "3")])
def test_pipeline_results_stay_the_same_under_load(slots, gpu_jobs):
    """scripts/stress_pipeline.py: rolling submission of 64-image jobs on several slots; every job's keypoints, descriptors and tracks
    must equal the slot's first round, and slot 0's first frame the oracle's"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_pipeline.py"), "40", slots, "16", gpu_jobs], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and " 0 bad" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
