"""CPU-only sanitizer run (ASan + UBSan): the host selection stage checked against the oracle inside the same binary,
and the oracle's own extraction / matching / track merge on two synthetic frames."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_selection_under_asan_ubsan():
    exe = os.path.join(ROOT, "tests", "cpp", "test_select_sanitize")
    src = [os.path.join(ROOT, "tests", "cpp", "test_select_sanitize.cpp"), os.path.join(ROOT, "mc-slam_amd", "csrc", "mcorb_select.cpp"),
           os.path.join(ROOT, "oracle", "mcorb_oracle.cpp")]
    synth_c, synth_o = os.path.join(ROOT, "mc-slam_amd", "csrc", "mcorb_synth.c"), os.path.join(ROOT, "tests", "cpp", "_synth_san.o")
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in src):
        subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-c", synth_c, "-o", synth_o])
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=all", "-D__host__=", "-D__device__=",
                               "-I" + os.path.join(ROOT, "mc-slam_amd", "csrc"), "-I" + os.path.join(ROOT, "oracle"),
                               "-I" + os.path.join(ROOT, "include")] + src + [synth_o, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad=0" in out.stdout


def test_sort_model_equals_std_sort():
    """mcorb_sortmodel.h -- libstdc++'s std::sort as the GPU selection kernel runs it (closed-form partition, independent
    sub-ranges, block-wise stable finish, shared heap sort) -- against std::sort itself: 12 000 random multisets full of ties,
    structured sequences, and median-of-three killers that force the heap-sort branch (asserted to have been taken)."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_sortmodel")
    src = [os.path.join(ROOT, "tests", "cpp", "test_sortmodel.cpp"), os.path.join(ROOT, "mc-slam_amd", "csrc", "mcorb_sortmodel.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-I" + os.path.join(ROOT, "mc-slam_amd", "csrc"), src[0], "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad=0" in out.stdout and "heap_cases=0" not in out.stdout


def test_signal_word_layout_and_checksum():
    """mcorb_signal.h: the word k_assemble hands the host per image (done / fallback / count / monoIndex / checksum) round-trips at its
    extremes, and the checksum notices a changed, stale or swapped entry"""
    exe = os.path.join(ROOT, "tests", "cpp", "test_signal")
    src = [os.path.join(ROOT, "tests", "cpp", "test_signal.cpp"), os.path.join(ROOT, "mc-slam_amd", "csrc", "mcorb_signal.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-I" + os.path.join(ROOT, "mc-slam_amd", "csrc"), src[0], "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bad=0" in out.stdout, out.stdout + out.stderr
