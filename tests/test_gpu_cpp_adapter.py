"""The C++ host mirror (include/mcorb_adapter.hpp) driven like MC-SLAM drives ORBextractor /
MultiCameraFrame, checked against the oracle through checksums."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_adapter")


def fnv(h, b):
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def build_exe():
    src = os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp")
    hdr = os.path.join(ROOT, "include", "mcorb_adapter.hpp")
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE,
                           "-L" + os.path.join(ROOT, "mc-slam_amd"), "-lmcorb",
                           "-Wl,-rpath," + os.path.join(ROOT, "mc-slam_amd")])


def test_adapter_header_compiles_without_a_gpu():
    build_exe()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_adapter_matches_oracle():
    import mcorb
    build_exe()
    C, W, H, N, frame = 3, 640, 480, 1000, 1
    vocab = O.make_vocabulary(10, 3, seed=5)
    vpath = os.path.join(ROOT, "tests", "cpp", "_vocab_test.txt")
    O.write_vocabulary_text(vocab, vpath)
    out = subprocess.run([EXE, str(C), str(W), str(H), str(N), str(frame), vpath], capture_output=True, text=True, timeout=120)
    os.remove(vpath)
    assert out.returncode == 0, out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines() if " " in l)
    imgs = [mcorb.synth_rig_frame(frame, C, c, W, H) for c in range(C)]
    ora = [O.OracleExtractor(N)(im) for im in imgs]
    F0 = 1469598103934665603
    mono, k, d = ora[0]
    h = fnv(fnv(F0, k.tobytes()), d.tobytes())
    assert "mono=%d n=%d hash=%016x levels=8" % (mono, len(k), h) in lines["extractor"]
    assert out.stdout.splitlines()[1].strip() == "empty=-1"           # the reference's return -1 (ORBextractor.cpp:1091)
    hr = F0
    for (_, kk, dd) in ora:
        hr = fnv(fnv(hr, kk.tobytes()), dd.tobytes())
    hm, nm = F0, 0
    for i in range(C - 1):
        for j in range(i + 1, C):
            i1, i2 = O.bruteforce_match(ora[i][2], ora[j][2])
            hm = fnv(fnv(hm, i1.astype(np.uint32).tobytes()), i2.astype(np.uint32).tobytes())
            nm += len(i1)
    tr, mg = O.intra_matches([o[2] for o in ora])
    ht = F0
    for row in tr:
        ht = fnv(ht, row.astype(np.int32).tobytes())
    want = "features=%016x matches=%d/%016x tracks=%d/%016x mergeable=%d" % (hr, nm, hm, len(tr), ht, mg)
    assert lines["rig"].strip() == want
    Fm = np.tile(np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]]), (C * (C - 1) // 2, 1, 1))
    tr, mg = O.intra_matches([o[2] for o in ora], F=Fm, kps=[o[1] for o in ora], sigma2=O.OracleExtractor(N).tables()["sigma2"])
    he = F0
    for row in tr:
        he = fnv(he, row.astype(np.int32).tobytes())
    assert lines["epipolar"].strip() == "tracks=%d/%016x mergeable=%d" % (len(tr), he, mg)
    # BoW-guided matcher and transform() through the C++ mirror, vocabulary read from its text file
    fvs, bows = [], []
    for o in ora:
        b, f = O.bow_transform(vocab, o[2], 2)
        bows.append(b); fvs.append(f)
    tr, nr, words = O.intra_matches_bow([o[2] for o in ora], [o[1]["y"] for o in ora], fvs)
    hb = F0
    for row, n in zip(tr, nr):
        hb = fnv(fnv(hb, row.astype(np.int32).tobytes()), np.int32(n).tobytes())
    hb = fnv(hb, words.astype(np.uint32).tobytes())
    hv = F0
    for i, v in zip(*bows[0]):
        hv = fnv(fnv(hv, np.uint32(i).tobytes()), np.float64(v).tobytes())
    for node in sorted(fvs[0]):
        hv = fnv(fnv(hv, np.uint32(node).tobytes()), fvs[0][node].astype(np.uint32).tobytes())
    assert lines["bow"].strip() == "tracks=%d words=%d hash=%016x transform=%d/%d/%016x" % (
        len(tr), len(words), hb, len(bows[0][0]), len(fvs[0]), hv)
