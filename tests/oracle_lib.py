"""ctypes binding of the CPU oracle (oracle/mcorb_oracle.{h,cpp}).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libmcorb_oracle.so")


class OrcKeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_i16p = C.POINTER(C.c_int16)
_ip = C.POINTER(C.c_int)


def build():
    src = [os.path.join(ORACLE_DIR, f) for f in ("mcorb_oracle.cpp", "mcorb_oracle.h", "Makefile")]
    if os.path.exists(ORACLE_SO) and all(os.path.getmtime(ORACLE_SO) >= os.path.getmtime(s) for s in src):
        return ORACLE_SO
    subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_get_tables.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _ip, _ip]
    L.orc_level_size.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _ip, _ip]
    L.orc_extract.restype = C.c_int
    L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_void_p, C.c_void_p, C.c_int, _ip]
    for name in ("orc_last_level", "orc_last_level_bordered", "orc_last_blurred"):
        f = getattr(L, name)
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _ip]
    L.orc_last_candidates.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, _f32p, C.c_int]
    L.orc_last_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_cv_round_f.argtypes = [C.c_float]
    L.orc_cv_round_d.argtypes = [C.c_double]
    L.orc_cv_floor_f.argtypes = [C.c_float]
    L.orc_cv_ceil_f.argtypes = [C.c_float]
    L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_resize_tables.argtypes = [C.c_int, C.c_int, _ip, _i16p]
    L.orc_copy_make_border_101.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.orc_fast_9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, C.c_int]
    L.orc_fast_corner_score.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_gaussian_blur_7x7_s2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.orc_gaussian_kernel_q8.argtypes = [_ip]
    L.orc_fast_atan2.restype = C.c_float
    L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.orc_stage_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.orc_distribute_octree.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, _ip, C.c_int]
    L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_get_matches_dist_ratio.argtypes = [C.c_void_p, _u32p, C.c_int, C.c_void_p, _u32p, C.c_int,
                                             C.c_double, _u32p, _u32p, _ip]
    L.orc_knn2.restype = None
    L.orc_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, _i32p, _i32p]
    L.orc_bruteforce_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                       _u32p, _u32p, C.c_int]
    L.orc_intra_matches.argtypes = [C.POINTER(C.c_void_p), _ip, C.c_int, C.c_float, C.c_float, _i32p,
                                    C.c_int, _ip]
    L.orc_representative_desc.argtypes = [C.c_void_p, C.c_int]
    L.orc_intra_matches_bow.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _ip, C.c_double, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_int, _ip]
    L.orc_bow_transform.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, _ip, C.c_void_p,
                                    C.c_void_p, _ip, C.c_void_p]
    _lib = L
    return L


def _ptr(a, t=None):
    return a.ctypes.data_as(t) if t is not None else a.ctypes.data


class OracleExtractor:
    """Mirror of the reference's ORBextractor (ORBextractor.h:43-116) on the oracle."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, orientation=0):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th, orientation)
        if not self.h:
            raise ValueError("orc_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        q = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.orc_get_tables(self.h, _ptr(sc, _f32p), _ptr(isc, _f32p), _ptr(s2, _f32p), _ptr(is2, _f32p),
                              _ptr(q, _ip), _ptr(um, _ip))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, quota=q, umax=um)

    def level_size(self, level, w, h):
        lw, lh = C.c_int(), C.c_int()
        self.L.orc_level_size(self.h, level, w, h, C.byref(lw), C.byref(lh))
        return lw.value, lh.value

    def __call__(self, img, lap=(0, 0), cap=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = cap or (self.nfeatures + 64 * self.nlevels)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        mono = self.L.orc_extract(self.h, _ptr(img), w, h, img.strides[0], lap[0], lap[1], _ptr(kps),
                                  _ptr(desc), cap, C.byref(n))
        if mono < 0:
            return mono, None, None
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def _plane(self, fn, level):
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        p = fn(self.h, level, C.byref(w), C.byref(h), C.byref(s))
        if not p:
            return None
        buf = (C.c_uint8 * (s.value * h.value)).from_address(p)
        # the last row of an interior view is shorter than stride; slice defensively
        a = np.ctypeslib.as_array(buf).reshape(h.value, s.value)[:, :w.value]
        return a.copy()

    def level(self, level):
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        p = self.L.orc_last_level_bordered(self.h, level, C.byref(w), C.byref(h), C.byref(s))
        buf = (C.c_uint8 * (s.value * h.value)).from_address(p)
        a = np.ctypeslib.as_array(buf).reshape(h.value, s.value)
        return a[19:-19, 19:-19].copy()

    def level_bordered(self, level):
        return self._plane(self.L.orc_last_level_bordered, level)

    def blurred(self, level):
        return self._plane(self.L.orc_last_blurred, level)

    def candidates(self, level, cap=200000):
        x, y, r = (np.zeros(cap, np.float32) for _ in range(3))
        n = self.L.orc_last_candidates(self.h, level, _ptr(x, _f32p), _ptr(y, _f32p), _ptr(r, _f32p), cap)
        return x[:n].copy(), y[:n].copy(), r[:n].copy()

    def level_keypoints(self, level, cap=20000):
        k = np.zeros(cap, KP_DTYPE)
        n = self.L.orc_last_level_keypoints(self.h, level, _ptr(k), cap)
        return k[:n].copy()


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(dst), dw, dh, dw)
    return dst


def resize_tables(ssize, dsize):
    ofs = np.zeros(dsize, np.int32)
    coef = np.zeros(2 * dsize, np.int16)
    lib().orc_resize_tables(ssize, dsize, _ptr(ofs, _ip), _ptr(coef, _i16p))
    return ofs, coef.reshape(dsize, 2)


def copy_make_border(src, border):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    lib().orc_copy_make_border_101(_ptr(src), w, h, src.strides[0], _ptr(dst), dst.strides[0], border)
    return dst


def fast(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = max(w * h, 1)
    xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().orc_fast_9_16(_ptr(img), img.strides[0], w, h, threshold, int(nonmax), _ptr(xs, _ip),
                            _ptr(ys, _ip), _ptr(sc, _ip), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def corner_score(img, x, y, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_fast_corner_score(_ptr(img), img.strides[0], x, y, threshold)


def gaussian_blur(src):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros((h, w), np.uint8)
    lib().orc_gaussian_blur_7x7_s2(_ptr(src), w, h, src.strides[0], _ptr(dst), w)
    return dst


def gaussian_kernel_q8():
    k = np.zeros(7, np.int32)
    lib().orc_gaussian_kernel_q8(_ptr(k, _ip))
    return k


def stage_f32(img):
    img = np.ascontiguousarray(img, np.float32)
    ch = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    out = np.zeros((h, w), np.uint8)
    lib().orc_stage_f32(_ptr(img), w, h, img.strides[0], ch, _ptr(out), w)
    return out


def distribute_octree(x, y, resp, minX, maxX, minY, maxY, N):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    resp = np.ascontiguousarray(resp, np.float32)
    out = np.zeros(len(x) + 8, np.int32)
    n = lib().orc_distribute_octree(_ptr(x, _f32p), _ptr(y, _f32p), _ptr(resp, _f32p), len(x), minX, maxX,
                                    minY, maxY, N, _ptr(out, _ip), len(out))
    return n, out[:max(n, 0)].copy()


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_ptr(a), _ptr(b))


def knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    lib().orc_knn2(_ptr(q), len(q), _ptr(t), len(t), _ptr(idx, _i32p), _ptr(dist, _i32p))
    return idx, dist


def bruteforce_match(q, t, dist_thresh=75.0, ratio=0.85):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    i1 = np.zeros(len(q) + 1, np.uint32)
    i2 = np.zeros(len(q) + 1, np.uint32)
    n = lib().orc_bruteforce_match(_ptr(q), len(q), _ptr(t), len(t), dist_thresh, ratio, _ptr(i1, _u32p),
                                   _ptr(i2, _u32p), len(q))
    return i1[:n].copy(), i2[:n].copy()


def intra_matches(descs, dist_thresh=75.0, ratio=0.85, F=None, kps=None, sigma2=None):
    """F given -> computeIntraMatches(matches, old=true) with the epipolar check."""
    descs = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in descs]
    nc = len(descs)
    ptrs = (C.c_void_p * nc)(*[_ptr(d) for d in descs])
    ns = np.array([len(d) for d in descs], np.int32)
    cap = int(ns.sum()) + 1
    tracks = np.full((cap, nc), -1, np.int32)
    merg = C.c_int()
    if F is not None:
        F = np.ascontiguousarray(F, np.float64)
        kps = [np.ascontiguousarray(k, KP_DTYPE) for k in kps]
        kp_ptrs = (C.c_void_p * nc)(*[_ptr(k) for k in kps])
        sigma2 = np.ascontiguousarray(sigma2, np.float32)
        f = lib().orc_intra_matches_epi
        f.argtypes = [C.POINTER(C.c_void_p), _ip, C.c_int, C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_void_p),
                      C.c_void_p, _i32p, C.c_int, _ip]
        f.restype = C.c_int
        n = f(ptrs, _ptr(ns, _ip), nc, dist_thresh, ratio, _ptr(F), kp_ptrs, _ptr(sigma2), _ptr(tracks, _i32p), cap, C.byref(merg))
        return tracks[:n].copy(), merg.value
    n = lib().orc_intra_matches(ptrs, _ptr(ns, _ip), nc, dist_thresh, ratio, _ptr(tracks, _i32p), cap,
                                C.byref(merg))
    return tracks[:n].copy(), merg.value


def get_matches_dist_ratio(A, iA, B, iB, ratio=0.85):
    A = np.ascontiguousarray(A, np.uint8).reshape(-1, 32)
    B = np.ascontiguousarray(B, np.uint8).reshape(-1, 32)
    iA = np.ascontiguousarray(iA, np.uint32)
    iB = np.ascontiguousarray(iB, np.uint32)
    mA = np.zeros(len(iA) + 1, np.uint32)
    mB = np.zeros(len(iA) + 1, np.uint32)
    bk = C.c_int(0)
    n = lib().orc_get_matches_dist_ratio(_ptr(A), _ptr(iA, _u32p), len(iA), _ptr(B), _ptr(iB, _u32p), len(iB),
                                         ratio, _ptr(mA, _u32p), _ptr(mB, _u32p), C.byref(bk))
    return mA[:n].copy(), mB[:n].copy(), bk.value


def argsorte(data, ascen=True):
    """argsorte (MCSlam/include/MCSlam/utils.h:21-30), std::sort tie placement included"""
    data = np.ascontiguousarray(data, np.float32)
    out = np.zeros(max(len(data), 1), np.int32)
    lib().orc_argsorte(data.ctypes.data_as(C.POINTER(C.c_float)), len(data), int(bool(ascen)), out.ctypes.data_as(C.POINTER(C.c_int)))
    return out[:len(data)]


def obtain_lf_features(kps, descs, tracks, K_mats, R_mats, t_mats, words=None, seg_masks=None, kps_undist=None, total_feats=3000):
    """FrontEnd::obtainLfFeatures (MCSlam/src/FrontEnd.cpp:213-593), the live branch (tiling = refview = false), restated
    statement by statement on numpy arrays.  Triangulation: cv::sfm::triangulatePoints (un-vendored opencv_contrib) -- two views:
    4x4 DLT design, more: the 3n x (4+n) design; null vector = numpy's SVD (LAPACK), FP64.  Returns (list of dict features,
    intramatch_size, mono_size, sorted unique words_fil)."""
    Cn = len(kps)
    und = kps_undist if kps_undist is not None else kps
    keypoint_mask = [np.ones(len(k), bool) for k in kps]
    prj = [np.hstack([np.asarray(R_mats[i], np.float64).reshape(3, 3), np.asarray(t_mats[i], np.float64).reshape(3, 1)]) for i in range(Cn)]
    K = [np.asarray(k, np.float64).reshape(3, 3) for k in K_mats]
    intra, mono, responses, wfil = [], [], [], set()

    def seg(i, x, y):
        if seg_masks is None or seg_masks[i] is None:
            return 0.0
        return float(seg_masks[i][int(y), int(x)])     # .at<float>(p.y, p.x): float -> int truncates

    intramatch_size = 0
    for ind, row in enumerate(np.asarray(tracks, np.int32).reshape(-1, Cn)):
        mi = [int(v) for v in row]
        view_inds, dd = [], []
        for i in range(Cn):
            if mi[i] != -1:
                kp = kps[i][mi[i]]
                if seg(i, kp["x"], kp["y"]) < 0.7:
                    dd.append(descs[i][mi[i]])
                    view_inds.append(i)
                else:
                    mi[i] = -1
        nv = len(view_inds)
        if nv > 1:
            xx = []
            for v in view_inds:
                kp = kps[v][mi[v]]
                xx.append(((float(kp["x"]) - K[v][0, 2]) / K[v][0, 0], (float(kp["y"]) - K[v][1, 2]) / K[v][1, 1]))
            if nv == 2:
                Pl, Pr = prj[view_inds[0]], prj[view_inds[1]]
                D = np.stack([xx[0][0] * Pl[2] - Pl[0], xx[0][1] * Pl[2] - Pl[1], xx[1][0] * Pr[2] - Pr[0], xx[1][1] * Pr[2] - Pr[1]])
                h = np.linalg.svd(D)[2][-1]
            else:
                D = np.zeros((3 * nv, 4 + nv))
                for i, v in enumerate(view_inds):
                    D[3 * i:3 * i + 3, :4] = -prj[v]
                    D[3 * i, 4 + i], D[3 * i + 1, 4 + i], D[3 * i + 2, 4 + i] = xx[i][0], xx[i][1], 1.0
                h = np.linalg.svd(D)[2][-1][:4]
            X = h[:3] / h[3]
            if X[2] < 40 and X[2] > 0.5:
                pr = K[0] @ X
                if words is not None:
                    wfil.add(int(words[ind]))
                rep = representative_desc(np.stack(dd))
                intra.append(dict(match_index=list(mi), uv_ref=(np.float32(pr[0] / pr[2]), np.float32(pr[1] / pr[2])), mono=0, n_rays=nv,
                                  point3d=X, desc=dd[rep]))
                intramatch_size += 1
                for v in view_inds:
                    keypoint_mask[v][mi[v]] = False
        elif nv == 1:
            v = view_inds[0]
            u = und[v][mi[v]]
            mono.append(dict(match_index=list(mi), uv_ref=(u["x"], u["y"]), mono=1, n_rays=1, point3d=np.zeros(3), desc=dd[0]))
            responses.append(u["response"])
            keypoint_mask[v][mi[v]] = False
    for i in range(Cn):
        for j in range(len(kps[i])):
            if keypoint_mask[i][j]:
                mi = [-1] * Cn
                mi[i] = j
                mono.append(dict(match_index=mi, uv_ref=(und[i][j]["x"], und[i][j]["y"]), mono=1, n_rays=1, point3d=np.zeros(3), desc=descs[i][j]))
                responses.append(und[i][j]["response"])
                keypoint_mask[i][j] = False
    order = argsorte(np.array(responses, np.float32), False)
    mono_size = 0
    for i in range(len(order)):
        if i >= total_feats - intramatch_size:
            break
        intra.append(mono[order[i]])
        mono_size += 1
    return intra, intramatch_size, mono_size, sorted(wfil)


def representative_desc(descs):
    descs = np.ascontiguousarray(descs, np.uint8).reshape(-1, 32)
    return lib().orc_representative_desc(_ptr(descs), len(descs))


def bow_transform(vocab, feats, levelsup=4):
    """vocab: dict(k, L, scoring, weighting, parent, is_leaf, desc, weight) in loadFromTextFile order."""
    feats = np.ascontiguousarray(feats, np.uint8).reshape(-1, 32)
    parent = np.ascontiguousarray(vocab["parent"], np.int32)
    leaf = np.ascontiguousarray(vocab["is_leaf"], np.uint8)
    nd = np.ascontiguousarray(vocab["desc"], np.uint8).reshape(-1, 32)
    nw = np.ascontiguousarray(vocab["weight"], np.float64)
    cap = max(len(feats), 1)
    ids, vals = np.zeros(cap, np.uint32), np.zeros(cap, np.float64)
    nodes, offs, ff = np.zeros(cap, np.uint32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
    nb, nf = C.c_int(), C.c_int()
    lib().orc_bow_transform(vocab["k"], vocab["L"], vocab["scoring"], vocab["weighting"], _ptr(parent), _ptr(leaf), _ptr(nd),
                            _ptr(nw), len(parent), _ptr(feats), len(feats), levelsup, _ptr(ids), _ptr(vals), C.byref(nb),
                            _ptr(nodes), _ptr(offs), C.byref(nf), _ptr(ff))
    bow = (ids[:nb.value].copy(), vals[:nb.value].copy())
    fv = {int(nodes[i]): ff[offs[i]:offs[i + 1]].copy() for i in range(nf.value)}
    return bow, fv


def make_vocabulary(k=10, L=3, seed=0, scoring=0, weighting=0, ragged=True, zero_weight_frac=0.05):
    """Synthetic vocabulary tree in loadFromTextFile order (breadth-first blocks of k children; the ORB
    vocabulary file itself is not part of the reference repository).  With ragged=True a few inner
    positions become leaves early, as k-means produces when a cluster cannot be split."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc, weight = [], [], [], []
    frontier = [(0, 0)]          # (node id, depth)
    nid = 0
    while frontier:
        pid, depth = frontier.pop(0)
        for _ in range(k):
            nid += 1
            is_leaf = depth + 1 == L or (ragged and depth + 1 >= 2 and rng.random() < 0.1)
            parent.append(pid)
            leaf.append(1 if is_leaf else 0)
            desc.append(rng.integers(0, 256, 32, dtype=np.uint8))
            w = 0.0
            if is_leaf:
                w = 0.0 if rng.random() < zero_weight_frac else float(rng.uniform(0.1, 9.0))
                if weighting in (1, 3):
                    w = 1.0 if w > 0 else 0.0
            weight.append(w)
            if not is_leaf:
                frontier.append((nid, depth + 1))
    return dict(k=k, L=L, scoring=scoring, weighting=weighting, parent=np.array(parent, np.int32),
                is_leaf=np.array(leaf, np.uint8), desc=np.stack(desc), weight=np.array(weight, np.float64))


def write_vocabulary_text(vocab, path):
    """saveToTextFile layout: 'k L scoring weighting' then 'parent isLeaf d0..d31 weight' per node."""
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (vocab["k"], vocab["L"], vocab["scoring"], vocab["weighting"]))
        for p, l, d, w in zip(vocab["parent"], vocab["is_leaf"], vocab["desc"], vocab["weight"]):
            f.write("%d %d %s %r\n" % (p, l, " ".join(str(int(b)) for b in d), float(w)))


def intra_matches_bow(descs, kp_y, fvs, ratio=0.85):
    """fvs: per camera FeatureVector as {node id: feature indices} (ascending iteration like std::map)."""
    nc = len(descs)
    descs = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in descs]
    ys = [np.ascontiguousarray(y, np.float32) for y in kp_y]
    nodes, offs, feats = [], [], []
    for fv in fvs:
        ks = sorted(fv)
        nodes.append(np.array(ks, np.uint32))
        o = np.zeros(len(ks) + 1, np.int32)
        o[1:] = np.cumsum([len(fv[k]) for k in ks])
        offs.append(o)
        feats.append(np.concatenate([np.asarray(fv[k], np.int32) for k in ks]) if ks else np.zeros(0, np.int32))
    vp = lambda arrs: (C.c_void_p * nc)(*[_ptr(a) for a in arrs])
    nfv = np.array([len(n) for n in nodes], np.int32)
    cap = int(sum(len(d) for d in descs)) + 1
    tracks = np.full((cap, nc), -1, np.int32)
    nr = np.zeros(cap, np.int32)
    words = np.zeros(cap, np.uint32)
    nw = C.c_int()
    n = lib().orc_intra_matches_bow(vp(descs), vp(ys), nc, vp(nodes), vp(offs), vp(feats), _ptr(nfv, _ip), ratio, _ptr(tracks),
                                    _ptr(nr), cap, _ptr(words), cap, C.byref(nw))
    return tracks[:n].copy(), nr[:n].copy(), words[:nw.value].copy()
