"""Host-side stages of the product (tables, geometry, resize coefficient tables, quad-tree
selection, Hamming helper, synthetic generator) against the oracle.  No GPU needed."""
import ctypes as C
from importlib import import_module

import numpy as np
import pytest

import oracle_lib as O

pkg = import_module("mc-slam_amd")
_lib = pkg._lib
L = _lib.load()


@pytest.mark.parametrize("nfeat,sf,nl", [(2000, 1.2, 8), (1000, 1.2, 8), (500, 1.5, 4), (3000, 1.1, 12), (300, 1.2, 1)])
def test_tables_match_oracle(nfeat, sf, nl):
    p = _lib.default_params(nfeatures=nfeat, scale_factor=sf, nlevels=nl)
    mine = pkg.get_tables(p)
    ref = O.OracleExtractor(nfeat, sf, nl).tables()
    for k in ("scale", "inv_scale", "sigma2", "inv_sigma2", "quota"):
        assert np.array_equal(mine[k], ref[k]), k


@pytest.mark.parametrize("w,h", [(640, 480), (1280, 720), (1920, 1080), (752, 480), (1241, 376)])
def test_geometry_matches_reference_arithmetic(w, h):
    p = _lib.default_params()
    six = np.zeros((8, 6), np.int32)
    assert L.mcorb_host_geometry(C.byref(p), w, h, six.ctypes.data) == 0
    ex = O.OracleExtractor()
    for l in range(8):
        lw, lh = ex.level_size(l, w, h)
        assert (six[l, 0], six[l, 1]) == (lw, lh)
        width, height = np.float32(lw - 32), np.float32(lh - 32)          # ORBextractor.cpp:796-802
        ncols, nrows = int(width / np.float32(35)), int(height / np.float32(35))
        assert (six[l, 2], six[l, 3]) == (ncols, nrows)
        assert six[l, 4] == int(np.ceil(width / ncols)) and six[l, 5] == int(np.ceil(height / nrows))


def test_geometry_rejects_sizes_the_reference_cannot_handle():
    p = _lib.default_params()
    six = np.zeros((8, 6), np.int32)
    assert L.mcorb_host_geometry(C.byref(p), 160, 120, six.ctypes.data) == _lib.E_SIZE    # level 7 narrower than a cell
    assert L.mcorb_host_geometry(C.byref(p), 480, 1400, six.ctypes.data) == _lib.E_SIZE   # nIni = round(w/h) = 0


@pytest.mark.parametrize("ssize,dsize", [(1280, 1067), (720, 600), (1067, 889), (640, 533), (357, 298), (201, 168), (50, 50), (7, 20)])
def test_resize_axis_tables_match_oracle(ssize, dsize):
    ofs, coef = O.resize_tables(ssize, dsize)             # x axis (clamped), as cv::resize builds it
    q = np.zeros((dsize, 4), np.int32)
    assert L.mcorb_host_resize_axis(ssize, dsize, 1, q.ctypes.data) == 0
    assert np.array_equal(q[:, 0], ofs)
    tail = ofs + 1 >= ssize
    assert np.array_equal(q[~tail, 2:], coef[~tail].astype(np.int32))
    assert np.all(q[tail, 2] == 2048) and np.all(q[tail, 3] == 0)        # HResizeLinear tail: S[sx]*ONE
    assert np.all(q[:, 1] == np.minimum(ofs + 1, ssize - 1))


def _pack(x, y, r):
    return (y.astype(np.uint32) << 20) | (x.astype(np.uint32) << 8) | r.astype(np.uint32)


def _select(x, y, r, W, H, N, wcell=0, hcell=0):
    packed = _pack(x, y, r)
    out = np.zeros(N + 80, np.int32)
    n = L.mcorb_host_select(packed.ctypes.data, len(packed), 16, W - 16, 16, H - 16, N, wcell, hcell,
                            out.ctypes.data, len(out))
    return n, out[:max(n, 0)]


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("W,H,N,npts", [(640, 480, 217, 3000), (1280, 720, 434, 9000), (357, 201, 122, 900),
                                        (300, 300, 50, 40), (640, 480, 5, 700)])
def test_selection_stage_matches_oracle_on_random_candidates(seed, W, H, N, npts):
    rng = np.random.default_rng(seed * 7 + N)
    cells = rng.permutation((W - 32) * (H - 32))[:npts]
    x = (cells % (W - 32)).astype(np.int32)
    y = (cells // (W - 32)).astype(np.int32)
    order = np.lexsort((x, y))                              # any fixed order; ties in response are common
    x, y = x[order], y[order]
    r = rng.integers(7, 60, npts).astype(np.int32)
    n_o, idx_o = O.distribute_octree(x.astype(np.float32), y.astype(np.float32), r.astype(np.float32),
                                     16, W - 16, 16, H - 16, N)
    n_p, idx_p = _select(x, y, r, W, H, N)
    assert n_p == n_o
    assert np.array_equal(idx_p, idx_o), "selection differs in content or ORDER"


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("W,H,N,npts,box", [(1280, 720, 434, 3000, 90), (640, 480, 217, 2500, 60), (1280, 720, 434, 6000, 200)])
def test_selection_stage_on_clustered_candidates(seed, W, H, N, npts, box):
    """All corners inside one small square (+ a few strays): the tree has to go several levels deeper than the
    GPU bucketing, i.e. the host partitions and scans keys itself -- and must agree with the bucket winners on
    the "first maximum in vToDistributeKeys order" tie rule (few distinct responses => many ties)."""
    rng = np.random.default_rng(100 + seed)
    bx, by = rng.integers(0, W - 32 - box), rng.integers(0, H - 32 - box)
    cells = rng.permutation(box * box)[:min(npts, box * box)]
    x = np.concatenate([bx + cells % box, rng.integers(0, W - 32, 25)]).astype(np.int32)
    y = np.concatenate([by + cells // box, rng.integers(0, H - 32, 25)]).astype(np.int32)
    _, uniq = np.unique(y.astype(np.int64) * 4096 + x, return_index=True)
    x, y = x[uniq], y[uniq]                                  # unique pixels, raster order
    r = rng.integers(20, 24, len(x)).astype(np.int32)
    n_o, idx_o = O.distribute_octree(x.astype(np.float32), y.astype(np.float32), r.astype(np.float32),
                                     16, W - 16, 16, H - 16, N)
    n_p, idx_p = _select(x, y, r, W, H, N)
    assert n_p == n_o and np.array_equal(idx_p, idx_o)


def test_selection_stage_on_real_fast_candidates():
    synth = import_module("mc-slam_amd.synth")
    img = synth.synth_rig_frame_numpy(1, 2, 1, 640, 480)
    ex = O.OracleExtractor(1000)
    ex(img)
    t = ex.tables()
    six = np.zeros((8, 6), np.int32)
    assert L.mcorb_host_geometry(C.byref(_lib.default_params(nfeatures=1000)), 640, 480, six.ctypes.data) == 0
    for l in range(8):
        x, y, r = ex.candidates(l)          # vToDistributeKeys order: cell row, cell col, then raster
        lw, lh = ex.level_size(l, 640, 480)
        n_o, idx_o = O.distribute_octree(x, y, r, 16, lw - 16, 16, lh - 16, int(t["quota"][l]))
        n_p, idx_p = _select(x.astype(np.int32), y.astype(np.int32), r.astype(np.int32), lw, lh, int(t["quota"][l]),
                             int(six[l, 4]), int(six[l, 5]))
        assert n_p == n_o and np.array_equal(idx_p, idx_o), "level %d" % l


def test_selection_edge_cases():
    assert L.mcorb_host_select(None, 0, 16, 600, 16, 400, 10, 0, 0, np.zeros(4, np.int32).ctypes.data, 4) == 0
    x = np.array([5], np.int32); y = np.array([7], np.int32); r = np.array([30], np.int32)
    n, idx = _select(x, y, r, 640, 480, 100)
    assert n == 1 and idx.tolist() == [0]
    # too tall a level -> the reference has no root node
    out = np.zeros(8, np.int32)
    p = _pack(x, y, r)
    assert L.mcorb_host_select(p.ctypes.data, 1, 16, 116, 16, 416, 10, 0, 0, out.ctypes.data, 8) == _lib.E_SIZE


def test_selection_rejects_sizes_beyond_the_packed_sort_key():
    """the host stage packs compareNodes' (key count, UL.x) into 32 bits (20 + 12): a level 4096 px wide would silently reorder
    ties -- it is refused instead (advisor finding, round 3)"""
    x = np.array([5, 4100], np.int32); y = np.array([7, 9], np.int32); r = np.array([30, 40], np.int32)
    out = np.zeros(80, np.int32)
    p = _pack(x, y, r)
    assert L.mcorb_host_select(p.ctypes.data, 2, 16, 16 + 4096, 16, 16 + 4000, 10, 0, 0, out.ctypes.data, 80) == _lib.E_SIZE
    assert L.mcorb_host_select(p.ctypes.data, 1, 16, 16 + 4095, 16, 16 + 4000, 10, 0, 0, out.ctypes.data, 80) == 1


def test_merge_tracks_bounds_check_is_unsigned():
    """an index of 0x80000000 in a gathered pair list is out of range, not negative (advisor finding, round 3)"""
    import ctypes as C
    counts = np.array([4, 4], np.int32)
    i1 = np.array([0x80000000], np.uint32); i2 = np.array([1], np.uint32)
    p1, p2 = (C.c_void_p * 1)(i1.ctypes.data), (C.c_void_p * 1)(i2.ctypes.data)
    npair = np.array([1], np.int32)
    tr = np.full((4, 2), -1, np.int32)
    n, mg = C.c_int(), C.c_int()
    assert L.mcorb_host_merge_tracks(2, counts.ctypes.data, p1, p2, npair.ctypes.data, tr.ctypes.data, 4, C.byref(n), C.byref(mg)) == _lib.E_ARG
    bad = np.array([-1, 4], np.int32)
    i1[0] = 0
    assert L.mcorb_host_merge_tracks(2, bad.ctypes.data, p1, p2, npair.ctypes.data, tr.ctypes.data, 4, C.byref(n), C.byref(mg)) == _lib.E_ARG
    assert L.mcorb_host_merge_tracks(2, counts.ctypes.data, p1, p2, npair.ctypes.data, tr.ctypes.data, 4, C.byref(n), C.byref(mg)) == 0 and n.value == 1


def test_hamming256_matches_reference_swar():
    rng = np.random.default_rng(0)
    for _ in range(100):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert pkg.hamming256(a, b) == O.descriptor_distance(a, b)


@pytest.mark.parametrize("w,h,nc", [(320, 240, 2), (640, 480, 4), (200, 150, 1)])
def test_synthetic_generator_c_equals_numpy(w, h, nc):
    for f in (0, 3):
        for c in range(nc):
            assert np.array_equal(pkg.synth_rig_frame(f, nc, c, w, h), pkg.synth_rig_frame_numpy(f, nc, c, w, h))
    # cameras are horizontal-disparity crops of one canvas
    a, b = pkg.synth_rig_frame(0, nc, 0, w, h), pkg.synth_rig_frame(0, max(nc, 2), 1, w, h)
    if nc >= 2:
        assert np.array_equal(a[:, 24:], b[:, :-24])


def test_representative_descriptor_matches_oracle():
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 4, 5, 8):
        for _ in range(20):
            base = rng.integers(0, 256, 32, dtype=np.uint8)
            d = np.stack([base ^ (rng.integers(0, 256, 32, dtype=np.uint8) & rng.integers(0, 256, 32, dtype=np.uint8)
                                  & rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(n)])
            assert pkg.representative_desc(d) == O.representative_desc(d)


def test_opencv_flavour_of_the_adapter_is_well_formed():
    """SYNTAX ONLY -- pins nothing.  OpenCV is not in this image, so the -DMCORB_WITH_OPENCV half of
    include/mcorb_adapter.hpp (the code a maintainer would compile next to MC-SLAM) is compiled against
    tests/cpp/cvmock, this repository's own minimal stand-in for the cv:: declarations it touches
    (InputArray / OutputArray / Mat / Mat_ / Range / KeyPoint).  A green test says the C++ is well-formed against
    those signatures; what OpenCV computes is not involved."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(root, "tests", "cpp", "cvmock"),
           "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "adapter_opencv_syntax.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_opencv_crosscheck_kit_is_well_formed():
    """tools/crosscheck/crosscheck_opencv.cpp (the program that would pin the oracle against real OpenCV and the reference's own
    ORBextractor.cpp) must at least be well-formed C++: -fsyntax-only against tests/cpp/cvmock (declarations, no behaviour) with
    -DCROSSCHECK_NO_REFERENCE (the reference's header needs the real <opencv2/opencv.hpp>).  Pins nothing, says so itself."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DCROSSCHECK_NO_REFERENCE", "-I" + os.path.join(root, "tests", "cpp", "cvmock"),
           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "oracle"), os.path.join(root, "tools", "crosscheck", "crosscheck_opencv.cpp")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    src = open(os.path.join(root, "tools", "crosscheck", "crosscheck_opencv.cpp")).read()
    assert "PINS NOTHING UNTIL SOMEONE RUNS IT" in src
