"""mc-slam_amd -- MI355X-native multi-camera ORB front-end (host-side mirror).

Python mirrors of the reference interfaces this path replaces, all thin wrappers
over the C ABI of libmcorb.so (include/mcorb.h):

  ORBextractor        MCSlam/include/MCSlam/ORBextractor.h:43-116
  MultiCameraFrame    MCSlam/include/MCSlam/MultiCameraFrame.h:59-92 (extract + intra-rig match members)
  Rig                 batch/async engine underneath (cameras x frames per launch)

The directory name is not a Python identifier; import it with
``importlib.import_module("mc-slam_amd")`` or through the ``mcorb`` shim at the
repository root.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (E_ARG, E_CAP, E_EMPTY, E_HIP, E_NODEVICE, E_OVERFLOW, E_SIZE, E_STATE, KP_DTYPE, OK,
                   ORIENT_IC_ANGLE, ORIENT_NONE, McorbError, Params, default_params)
from .synth import synth_rig_frame, synth_rig_frame_numpy

TH_HIGH = 100   # ORBextractor.h:26
TH_LOW = 75     # ORBextractor.h:27
HISTO_LENGTH = 30


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def device_count():
    return _lib.load().mcorb_device_count()


def get_tables(params):
    """Scale tables + per-level quotas (ORBextractor.cpp:413-444)."""
    n = params.nlevels
    sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
    q = np.zeros(n, np.int32)
    _lib.check(_lib.load().mcorb_get_tables(C.byref(params), sc.ctypes.data, isc.ctypes.data, s2.ctypes.data,
                                            is2.ctypes.data, q.ctypes.data))
    return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, quota=q)


def hamming256(a, b):
    """ORBextractor::DescriptorDistance (ORBextractor.cpp:1202-1218)."""
    a, b = _u8(a).reshape(32), _u8(b).reshape(32)
    return _lib.load().mcorb_hamming256(a.ctypes.data, b.ctypes.data)


def representative_desc(descs):
    """MultiCameraFrame::computeRepresentativeDesc (MultiCameraFrame.cpp:530-567): index of the chosen row."""
    d = _u8(descs).reshape(-1, 32)
    r = _lib.load().mcorb_representative_desc(d.ctypes.data, len(d))
    if r < 0:
        _lib.check(r)
    return r


class Rig:
    """One engine per GPU: `ncams` cameras of w x h, up to `max_frames` rig frames per batch."""

    def __init__(self, ncams, width, height, max_frames=1, nslots=1, params=None, **kw):
        self.L = _lib.load()
        self.params = params if params is not None else default_params(**kw)
        self.ncams, self.w, self.h, self.max_frames, self.nslots = ncams, width, height, max_frames, nslots
        h = C.c_void_p()
        _lib.check(self.L.mcorb_rig_create(C.byref(self.params), ncams, width, height, max_frames, nslots, C.byref(h)))
        self.h_rig = h
        self.kcap = self.L.mcorb_rig_kcap(h)
        self.nlevels = self.params.nlevels

    def close(self):
        if getattr(self, "h_rig", None):
            self.L.mcorb_rig_destroy(self.h_rig)
            self.h_rig = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- staging ------------------------------------------------------------
    def upload(self, images, slot=0):
        """images: list of HxW uint8 arrays (u8 fast path) or float32 [0,1] HxW / HxWx3 (reference format)."""
        first = np.asarray(images[0])
        if first.dtype == np.uint8:
            arrs = [_u8(im) for im in images]
            for a in arrs:
                if a.shape != (self.h, self.w):
                    raise ValueError("image shape %s != (%d,%d)" % (a.shape, self.h, self.w))
            ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
            _lib.check(self.L.mcorb_rig_upload_u8(self.h_rig, slot, ptrs, len(arrs), self.w))
        else:
            arrs = [np.ascontiguousarray(im, dtype=np.float32) for im in images]
            ch = 1 if arrs[0].ndim == 2 else arrs[0].shape[2]
            ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
            _lib.check(self.L.mcorb_rig_upload_f32(self.h_rig, slot, ptrs, len(arrs), arrs[0].strides[0], ch))
        return len(arrs)

    # -- extraction ---------------------------------------------------------
    def staging(self, m, slot=0):
        """The slot's pinned staging plane of image m as a writable (H, W) uint8 array (zero-copy hand-off)."""
        ptr, stride = C.c_void_p(), C.c_int()
        _lib.check(self.L.mcorb_rig_staging(self.h_rig, slot, m, C.byref(ptr), C.byref(stride)))
        buf = (C.c_uint8 * (self.h * stride.value)).from_address(ptr.value)
        return np.frombuffer(buf, np.uint8).reshape(self.h, stride.value)[:, :self.w]

    def upload_staged(self, nimg, slot=0):
        _lib.check(self.L.mcorb_rig_upload_staged(self.h_rig, slot, nimg))

    def extract(self, nimg, slot=0, lap=(0, 0)):
        _lib.check(self.L.mcorb_rig_extract(self.h_rig, slot, nimg, lap[0], lap[1]))

    def extract_submit(self, nimg, slot=0, lap=(0, 0)):
        _lib.check(self.L.mcorb_rig_extract_submit(self.h_rig, slot, nimg, lap[0], lap[1]))

    def extract_wait(self, slot=0):
        _lib.check(self.L.mcorb_rig_extract_wait(self.h_rig, slot))

    def process_submit(self, nframes, slot=0, lap=(0, 0), dist_thresh=75.0, ratio=0.85):
        _lib.check(self.L.mcorb_rig_process_submit(self.h_rig, slot, nframes, lap[0], lap[1], dist_thresh, ratio))

    def process(self, nframes, slot=0, lap=(0, 0), dist_thresh=75.0, ratio=0.85):
        """extract + intra-rig match, synchronously on the calling thread."""
        _lib.check(self.L.mcorb_rig_process(self.h_rig, slot, nframes, lap[0], lap[1], dist_thresh, ratio))

    def process_wait(self, slot=0):
        _lib.check(self.L.mcorb_rig_process_wait(self.h_rig, slot))

    def features(self, m, slot=0):
        n = self.L.mcorb_rig_num_keypoints(self.h_rig, slot, m)
        if n < 0:
            _lib.check(n)
        kps = np.zeros(n, KP_DTYPE)
        desc = np.zeros((n, 32), np.uint8)
        nn, mono = C.c_int(), C.c_int()
        _lib.check(self.L.mcorb_rig_get_features(self.h_rig, slot, m, kps.ctypes.data, desc.ctypes.data, n,
                                                 C.byref(nn), C.byref(mono)))
        return mono.value, kps, desc

    # -- matching -----------------------------------------------------------
    def match(self, nframes, slot=0, dist_thresh=75.0, ratio=0.85):
        _lib.check(self.L.mcorb_rig_match(self.h_rig, slot, nframes, dist_thresh, ratio))

    def pair_matches(self, frame, i, j, slot=0):
        i1 = np.zeros(self.kcap, np.uint32)
        i2 = np.zeros(self.kcap, np.uint32)
        n = C.c_int()
        _lib.check(self.L.mcorb_rig_get_pair_matches(self.h_rig, slot, frame, i, j, i1.ctypes.data, i2.ctypes.data,
                                                     self.kcap, C.byref(n)))
        return i1[:n.value].copy(), i2[:n.value].copy()

    def pair_knn2(self, frame, i, j, slot=0):
        idx = np.zeros((self.kcap, 2), np.int32)
        dist = np.zeros((self.kcap, 2), np.int32)
        n = C.c_int()
        _lib.check(self.L.mcorb_rig_get_pair_knn2(self.h_rig, slot, frame, i, j, idx.ctypes.data, dist.ctypes.data,
                                                  self.kcap, C.byref(n)))
        return idx[:n.value].copy(), dist[:n.value].copy()

    def tracks(self, frame, slot=0):
        cap = self.kcap * self.ncams
        tr = np.full((cap, self.ncams), -1, np.int32)
        n, mg = C.c_int(), C.c_int()
        _lib.check(self.L.mcorb_rig_get_tracks(self.h_rig, slot, frame, tr.ctypes.data, cap, C.byref(n), C.byref(mg)))
        return tr[:n.value].copy(), mg.value

    def tracks_epipolar(self, frame, F, kps_undist=None, slot=0):
        """computeIntraMatches(matches, old=true): F is (npairs, 3, 3) float64 in pair order (0,1),(0,2).."""
        npairs = self.ncams * (self.ncams - 1) // 2
        F = np.ascontiguousarray(F, np.float64).reshape(npairs, 3, 3)
        kp_ptrs = None
        if kps_undist is not None:
            keep = [np.ascontiguousarray(k, KP_DTYPE) for k in kps_undist]
            kp_ptrs = (C.c_void_p * self.ncams)(*[k.ctypes.data for k in keep])
        cap = self.kcap * self.ncams
        tr = np.full((cap, self.ncams), -1, np.int32)
        n, mg = C.c_int(), C.c_int()
        _lib.check(self.L.mcorb_rig_get_tracks_epipolar(self.h_rig, slot, frame, F.ctypes.data, kp_ptrs, tr.ctypes.data, cap,
                                                        C.byref(n), C.byref(mg)))
        return tr[:n.value].copy(), mg.value

    # -- intermediates (parity tests) -----------------------------------------
    def obtain_lf_features(self, frame, tracks, K_mats, R_mats, t_mats, words=None, seg_masks=None, kps_undist=None,
                           total_feats=3000, slot=0):
        """FrontEnd::obtainLfFeatures (FrontEnd.cpp:213-593) for one frame of the slot: returns (features as a structured
        array [match_index, uv_ref, mono, n_rays, point3d, desc], intramatch_size, mono_size, words_fil)."""
        Cn = self.ncams
        tracks = np.ascontiguousarray(tracks, np.int32).reshape(-1, Cn)
        cams = (_lib.Camera * Cn)()
        for c in range(Cn):
            K = np.asarray(K_mats[c], np.float64).reshape(3, 3)
            Rt = np.hstack([np.asarray(R_mats[c], np.float64).reshape(3, 3), np.asarray(t_mats[c], np.float64).reshape(3, 1)])   # build_Rt
            cams[c].K[:] = K.ravel().tolist()
            cams[c].Rt[:] = Rt.ravel().tolist()
        keep, segp, stride = [], None, 0
        if seg_masks is not None:
            segp = (C.c_void_p * Cn)()
            for c in range(Cn):
                if seg_masks[c] is not None:
                    a = np.ascontiguousarray(seg_masks[c], np.float32)
                    assert stride in (0, a.shape[1]), "all segmentation masks must share one row stride"
                    keep.append(a)
                    segp[c] = a.ctypes.data
                    stride = a.shape[1]
        undp = None
        if kps_undist is not None:
            undp = (C.c_void_p * Cn)()
            for c in range(Cn):
                a = np.ascontiguousarray(kps_undist[c], _lib.KP_DTYPE)
                keep.append(a)
                undp[c] = a.ctypes.data
        wp = None
        if words is not None:
            words = np.ascontiguousarray(words, np.uint32)
            assert len(words) >= len(tracks)
            wp = words.ctypes.data
        cap = max(total_feats, len(tracks)) + 1     # intramatch_size <= tracks, and the mono fill stops at total_feats
        out = np.empty(cap, _lib.LF_DTYPE)
        wf = np.zeros(len(tracks) + 1, np.uint32)
        n, ni, nm, nw = (C.c_int() for _ in range(4))
        _lib.check(self.L.mcorb_rig_obtain_lf_features(self.h_rig, slot, frame, tracks.ctypes.data, len(tracks), wp, cams, segp, stride, undp,
                                                       total_feats, out.ctypes.data, cap, C.byref(n), C.byref(ni), C.byref(nm),
                                                       wf.ctypes.data, len(wf), C.byref(nw)))
        return out[:n.value].copy(), ni.value, nm.value, wf[:nw.value].copy()

    def obtain_lf_features_frames(self, frame0, tracks_per_frame, K_mats, R_mats, t_mats, words_per_frame=None, seg_masks=None,
                                  kps_undist=None, total_feats=3000, slot=0):
        """obtainLfFeatures for frames [frame0, frame0 + len(tracks_per_frame)) of the slot in ONE call (one worker-pool task per
        frame).  seg_masks / kps_undist: lists indexed frame * ncams + cam (None entries allowed for masks).  Returns a list of
        (features, intramatch_size, mono_size, words_fil), one per frame."""
        Cn, F = self.ncams, len(tracks_per_frame)
        trs = [np.ascontiguousarray(t, np.int32).reshape(-1, Cn) for t in tracks_per_frame]
        nt = np.array([len(t) for t in trs], np.int32)
        tracks = np.ascontiguousarray(np.concatenate(trs) if nt.sum() else np.zeros((0, Cn), np.int32))
        cams = (_lib.Camera * Cn)()
        for c in range(Cn):
            K = np.asarray(K_mats[c], np.float64).reshape(3, 3)
            Rt = np.hstack([np.asarray(R_mats[c], np.float64).reshape(3, 3), np.asarray(t_mats[c], np.float64).reshape(3, 1)])
            cams[c].K[:] = K.ravel().tolist()
            cams[c].Rt[:] = Rt.ravel().tolist()
        keep, segp, stride = [], None, 0
        if seg_masks is not None:
            segp = (C.c_void_p * (F * Cn))()
            for i in range(F * Cn):
                if seg_masks[i] is not None:
                    a = np.ascontiguousarray(seg_masks[i], np.float32)
                    assert stride in (0, a.shape[1]), "all segmentation masks must share one row stride"
                    keep.append(a)
                    segp[i] = a.ctypes.data
                    stride = a.shape[1]
        undp = None
        if kps_undist is not None:
            undp = (C.c_void_p * (F * Cn))()
            for i in range(F * Cn):
                a = np.ascontiguousarray(kps_undist[i], _lib.KP_DTYPE)
                keep.append(a)
                undp[i] = a.ctypes.data
        wp = None
        if words_per_frame is not None:
            ws = [np.ascontiguousarray(w, np.uint32)[:n] for w, n in zip(words_per_frame, nt)]
            assert all(len(w) == n for w, n in zip(ws, nt)), "one word per track"
            words = np.ascontiguousarray(np.concatenate(ws) if nt.sum() else np.zeros(0, np.uint32))
            keep.append(words)
            wp = words.ctypes.data
        cap = max(total_feats, int(nt.max(initial=0))) + 1     # intramatch_size <= tracks, and the mono fill stops at total_feats
        capw = int(nt.max(initial=0)) + 1
        out = np.empty((F, cap), _lib.LF_DTYPE)
        wf = np.zeros((F, capw), np.uint32)
        n, ni, nm, nw = (np.zeros(F, np.int32) for _ in range(4))
        _lib.check(self.L.mcorb_rig_obtain_lf_features_frames(self.h_rig, slot, frame0, F, tracks.ctypes.data, nt.ctypes.data, wp, cams, segp,
                                                              stride, undp, total_feats, out.ctypes.data, cap, n.ctypes.data, ni.ctypes.data,
                                                              nm.ctypes.data, wf.ctypes.data, capw, nw.ctypes.data))
        return [(out[f, :n[f]].copy(), int(ni[f]), int(nm[f]), wf[f, :nw[f]].copy()) for f in range(F)]

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        _lib.check(self.L.mcorb_rig_level_size(self.h_rig, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def level(self, m, level, slot=0, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        fn = self.L.mcorb_rig_get_blurred if blurred else self.L.mcorb_rig_get_level
        _lib.check(fn(self.h_rig, slot, m, level, out.ctypes.data, w))
        return out

    def candidates(self, m, level, slot=0, cap=1 << 17):
        buf = np.zeros(cap, np.uint32)
        n = C.c_int()
        _lib.check(self.L.mcorb_rig_get_candidates(self.h_rig, slot, m, level, buf.ctypes.data, cap, C.byref(n)))
        p = buf[:n.value]
        return ((p >> 8) & 0xfff).astype(np.int32), (p >> 20).astype(np.int32), (p & 0xff).astype(np.int32)

    def info(self):
        o = np.zeros(8, np.int32)
        _lib.check(self.L.mcorb_rig_info(self.h_rig, o.ctypes.data))
        d = dict(zip(("kcap", "cells", "tiles", "cell_cap", "cand_cap", "bucket_total", "img_bytes", "nlevels"), (int(v) for v in o)))
        d["host_threads"] = int(self.L.mcorb_rig_host_threads(self.h_rig))
        return d

    def select_mode(self):
        """'gpu' (DistributeOctTree's list discipline in k_select: a job is one submission) or 'host' (worker pool between two GPU phases)"""
        return {1: "host", 2: "gpu"}[self.L.mcorb_rig_select_mode(self.h_rig)]

    def set_graph(self, every):
        """replay GPU-selected jobs from a captured HIP graph: 0 never, 1 always, K > 1 all but every K-th job of a slot"""
        _lib.check(self.L.mcorb_rig_set_graph(self.h_rig, every))

    def select_fallbacks(self, slot=0):
        """jobs of the slot the host stage had to redo (a level whose tree went below the GPU bucketing depth)"""
        return int(self.L.mcorb_rig_select_fallbacks(self.h_rig, slot))

    def early_reads_rejected(self, slot=0):
        """small batches: images whose early read did not match the signal word's checksum (records redone after the end event)"""
        return int(self.L.mcorb_rig_early_reads_rejected(self.h_rig, slot))

    def timing(self, slot=0):
        t = (C.c_float * 10)()
        _lib.check(self.L.mcorb_rig_last_timing(self.h_rig, slot, t))
        return dict(phase_a_us=t[0], select_us=t[1], phase_b_us=t[2], match_us=t[3], pyramid_us=t[4],
                    fast_us=t[5], compact_us=t[6], knn2_us=t[7], blur_us=t[8], describe_us=t[9])

    # -- multi-GPU plumbing ---------------------------------------------------
    def export_descriptors(self, dst_dev_ptr, nimg, slot=0):
        counts = np.zeros(nimg, np.int32)
        _lib.check(self.L.mcorb_rig_export_descriptors(self.h_rig, slot, dst_dev_ptr, counts.ctypes.data, nimg))
        return counts

    def match_external(self, desc_dev_ptr, counts, sets, slot=0, dist_thresh=75.0, ratio=0.85):
        self.match_external_submit(desc_dev_ptr, counts, sets, slot, dist_thresh, ratio)
        self.match_wait(slot)

    def match_external_submit(self, desc_dev_ptr, counts, sets, slot=0, dist_thresh=75.0, ratio=0.85):
        counts = np.ascontiguousarray(counts, np.int32)
        sets = np.ascontiguousarray(sets, np.int32).reshape(-1, self.ncams)
        self._keep = getattr(self, "_keep", {})
        self._keep[slot] = (counts, sets)          # must outlive the asynchronous job
        _lib.check(self.L.mcorb_rig_match_external_submit(self.h_rig, slot, desc_dev_ptr, counts.ctypes.data, len(counts),
                                                          sets.ctypes.data, len(sets), dist_thresh, ratio))

    def export_descriptors_dev(self, dst_dev_ptr, counts_dev_ptr, nimg, slot=0, then_stream=None):
        """Stream-ordered export: sets + int32 counts into caller device memory; `then_stream` (raw HIP stream handle)
        waits for the copies."""
        if then_stream:
            _lib.require_torch_first("export_descriptors_dev")
        _lib.check(self.L.mcorb_rig_export_descriptors_dev(self.h_rig, slot, dst_dev_ptr, counts_dev_ptr, nimg, then_stream))

    def match_external_dev_submit(self, desc_dev_ptr, counts_dev_ptr, ntotal, sets, slot=0, dist_thresh=75.0, ratio=0.85,
                                  after_stream=None):
        """External match with device-resident counts; the slot's stream waits for `after_stream` (the collective)."""
        if after_stream:
            _lib.require_torch_first("match_external_dev_submit")
        sets = np.ascontiguousarray(sets, np.int32).reshape(-1, self.ncams)
        self._keep = getattr(self, "_keep", {})
        self._keep[slot] = (sets,)                 # must outlive the asynchronous job
        _lib.check(self.L.mcorb_rig_match_external_dev_submit(self.h_rig, slot, desc_dev_ptr, counts_dev_ptr, int(ntotal),
                                                              sets.ctypes.data, len(sets), dist_thresh, ratio, after_stream))

    def match_wait(self, slot=0):
        _lib.check(self.L.mcorb_rig_match_wait(self.h_rig, slot))

    # pair-partitioned matching: explicit (query set, train set) pairs of an external block
    def match_pairs_external(self, desc_dev_ptr, counts, pair_sets, slot=0, dist_thresh=75.0, ratio=0.85):
        counts = np.ascontiguousarray(counts, np.int32)
        pair_sets = np.ascontiguousarray(pair_sets, np.int32).reshape(-1, 2)
        _lib.check(self.L.mcorb_rig_match_pairs_external(self.h_rig, slot, desc_dev_ptr, counts.ctypes.data, len(counts),
                                                         pair_sets.ctypes.data, len(pair_sets), dist_thresh, ratio))

    def match_pairs_external_dev_submit(self, desc_dev_ptr, counts_dev_ptr, ntotal, pair_sets, slot=0, dist_thresh=75.0, ratio=0.85,
                                        after_stream=None):
        if after_stream:
            _lib.require_torch_first("match_pairs_external_dev_submit")
        pair_sets = np.ascontiguousarray(pair_sets, np.int32).reshape(-1, 2)
        self._keep = getattr(self, "_keep", {})
        self._keep[slot] = (pair_sets,)            # must outlive the asynchronous job
        _lib.check(self.L.mcorb_rig_match_pairs_external_dev_submit(self.h_rig, slot, desc_dev_ptr, counts_dev_ptr, int(ntotal),
                                                                    pair_sets.ctypes.data, len(pair_sets), dist_thresh, ratio, after_stream))

    def match_sets(self, block, pair_sets, slot=0, dist_thresh=75.0, ratio=0.85):
        """knnMatch(k=2) + (dist_thresh, ratio) filter between sets of a DescriptorBlock; read with pairlist / pairknn2"""
        pair_sets = np.ascontiguousarray(pair_sets, np.int32).reshape(-1, 2)
        _lib.check(self.L.mcorb_rig_match_sets(self.h_rig, slot, block.h, pair_sets.ctypes.data, len(pair_sets), dist_thresh, ratio))

    def pairknn2(self, pair, slot=0):
        """raw knnMatch(k=2) table of pair `pair` of the last explicit-pair match -> (idx [nq][2], dist [nq][2]), -1 = absent"""
        idx, dist = np.zeros((self.kcap, 2), np.int32), np.zeros((self.kcap, 2), np.int32)
        n = C.c_int()
        _lib.check(self.L.mcorb_rig_get_pairknn2(self.h_rig, slot, pair, idx.ctypes.data, dist.ctypes.data, self.kcap, C.byref(n)))
        return idx[:n.value].copy(), dist[:n.value].copy()

    def pairlist(self, pair, slot=0):
        """accepted (query, train) indices of pair `pair` of the last explicit-pair match -> (idx1, idx2)"""
        i1, i2 = np.zeros(self.kcap, np.uint32), np.zeros(self.kcap, np.uint32)
        n = C.c_int()
        _lib.check(self.L.mcorb_rig_get_pairlist(self.h_rig, slot, pair, i1.ctypes.data, i2.ctypes.data, self.kcap, C.byref(n)))
        return i1[:n.value].copy(), i2[:n.value].copy()


class ORBextractor:
    """Mirror of the reference's ORBextractor (ORBextractor.h:43-116) over libmcorb."""

    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, orientation=ORIENT_NONE, device=0):
        self.L = _lib.load()
        self.params = default_params(nfeatures=nfeatures, scale_factor=scaleFactor, nlevels=nlevels,
                                     ini_th_fast=iniThFAST, min_th_fast=minThFAST, orientation=orientation,
                                     device_id=device)
        self._tables = get_tables(self.params)
        self.max_neighbor_ratio = 0.85   # ORBextractor.h:90
        h = C.c_void_p()
        _lib.check(self.L.mcorb_create(C.byref(self.params), 0, 0, C.byref(h)))
        self.h_ext = h

    def close(self):
        if getattr(self, "h_ext", None):
            self.L.mcorb_destroy(self.h_ext)
            self.h_ext = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """operator() (ORBextractor.cpp:1085-1171): returns (monoIndex, keypoints, descriptors);
        (-1, None, None) for an empty image, as the reference returns -1.  `mask` is ignored, as in the reference."""
        if image is None or np.asarray(image).size == 0:
            return -1, None, None
        img = np.asarray(image)
        cap = self.params.nfeatures + 8 * self.params.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n, mono = C.c_int(), C.c_int()
        if img.dtype == np.uint8:
            assert img.ndim == 2, "image.type() == CV_8UC1 (ORBextractor.cpp:1094)"
            img = _u8(img)
            st = self.L.mcorb_extract(self.h_ext, img.ctypes.data, img.shape[1], img.shape[0], img.strides[0],
                                      vLappingArea[0], vLappingArea[1], kps.ctypes.data, desc.ctypes.data, cap,
                                      C.byref(n), C.byref(mono))
        else:
            img = np.ascontiguousarray(img, np.float32)
            ch = 1 if img.ndim == 2 else img.shape[2]
            st = self.L.mcorb_extract_f32(self.h_ext, img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], ch,
                                          vLappingArea[0], vLappingArea[1], kps.ctypes.data, desc.ctypes.data, cap,
                                          C.byref(n), C.byref(mono))
        if st == E_EMPTY:
            return -1, None, None
        _lib.check(st)
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    def pyramid_level(self, level):
        """mvImagePyramid[level] (interior) of the last call (ORBextractor.h:89)."""
        w, h = C.c_int(), C.c_int()
        _lib.check(self.L.mcorb_get_pyramid_level(self.h_ext, level, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _lib.check(self.L.mcorb_get_pyramid_level(self.h_ext, level, out.ctypes.data, w.value, C.byref(w), C.byref(h)))
        return out

    def GetLevels(self):
        return self.params.nlevels

    def GetScaleFactor(self):
        return float(np.float32(self.params.scale_factor))

    def GetScaleFactors(self):
        return self._tables["scale"].copy()

    def GetInverseScaleFactors(self):
        return self._tables["inv_scale"].copy()

    def GetScaleSigmaSquares(self):
        return self._tables["sigma2"].copy()

    def GetInverseScaleSigmaSquares(self):
        return self._tables["inv_sigma2"].copy()

    def DescriptorDistance(self, a, b):
        return hamming256(a, b)

    def knnMatch2(self, q, t):
        """DescriptorMatcher("BruteForce-Hamming")->knnMatch(q, t, out, 2) on the GPU."""
        q, t = _u8(q).reshape(-1, 32), _u8(t).reshape(-1, 32)
        idx = np.zeros((len(q), 2), np.int32)
        dist = np.zeros((len(q), 2), np.int32)
        _lib.check(self.L.mcorb_knn2(self.h_ext, q.ctypes.data, len(q), t.ctypes.data, len(t), idx.ctypes.data,
                                     dist.ctypes.data))
        return idx, dist

    def matchRatio(self, q, t, dist_thresh=75.0, ratio=0.85):
        """BruteForceMatch's knn2 + ratio/threshold filter (MultiCameraFrame.cpp:1053-1078)."""
        q, t = _u8(q).reshape(-1, 32), _u8(t).reshape(-1, 32)
        i1 = np.zeros(len(q) + 1, np.uint32)
        i2 = np.zeros(len(q) + 1, np.uint32)
        n = C.c_int()
        _lib.check(self.L.mcorb_match_ratio(self.h_ext, q.ctypes.data, len(q), t.ctypes.data, len(t), dist_thresh,
                                            ratio, i1.ctypes.data, i2.ctypes.data, len(q), C.byref(n)))
        return i1[:n.value].copy(), i2[:n.value].copy()

    def getMatches_distRatio(self, A, i_A, B, i_B):
        """ORBextractor::getMatches_distRatio (ORBextractor.cpp:1228-1290): best / second-best search on the
        GPU k-NN kernel, the reference's one-to-one bookkeeping on the host.  Returns (i_match_A, i_match_B, BookK)."""
        A, B = _u8(A).reshape(-1, 32), _u8(B).reshape(-1, 32)
        i_A, i_B = np.asarray(i_A, np.int64), np.asarray(i_B, np.int64)
        mA, mB = [], []
        book = len(i_A) * len(i_B)
        if len(i_A) == 0 or len(i_B) == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32), book
        idx, dist = self.knnMatch2(A[i_A], B[i_B])
        for a in range(len(i_A)):
            d1 = float(dist[a, 0])
            d2 = float(dist[a, 1]) if idx[a, 1] >= 0 else 1e9
            # best_dist_1 / best_dist_2 is a double division in the reference (:1264): 0 / 0 (duplicate descriptors) is NaN
            # there, the comparison is false and the feature is skipped
            if d1 <= TH_LOW and d2 > 0 and d1 / d2 <= self.max_neighbor_ratio:
                idx_B = int(i_B[idx[a, 0]])
                if idx_B not in mB:
                    mB.append(idx_B)
                    mA.append(int(i_A[a]))
                else:
                    k = mB.index(idx_B)
                    d = hamming256(A[mA[k]], B[idx_B])
                    book += 1
                    if d1 < d:
                        mA[k] = int(i_A[a])
        return np.array(mA, np.uint32), np.array(mB, np.uint32), book


class ORBVocabulary:
    """DBoW2::ORBVocabulary (MCSlam/include/MCSlam/ORBVocabulary.h:21-30) as far as this path uses it:
    loadFromTextFile + transform(features, BowVector, FeatureVector, levelsup)."""

    def __init__(self, device=0):
        self.L_ = _lib.load()
        self.device = device
        self.h = None

    def loadFromTextFile(self, filename):
        h = C.c_void_p()
        st = self.L_.mcorb_vocab_load_text(filename.encode(), self.device, C.byref(h))
        if st != OK:
            return False          # the reference returns false and the caller exits (FrontEnd.h:139-143)
        self.close()
        self.h = h
        return True

    def create(self, k, L, scoring, weighting, parent, is_leaf, desc, weight):
        parent = np.ascontiguousarray(parent, np.int32)
        is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = _u8(desc).reshape(-1, 32)
        weight = np.ascontiguousarray(weight, np.float64)
        h = C.c_void_p()
        _lib.check(self.L_.mcorb_vocab_create(k, L, scoring, weighting, parent.ctypes.data, is_leaf.ctypes.data,
                                              desc.ctypes.data, weight.ctypes.data, len(parent), self.device, C.byref(h)))
        self.close()
        self.h = h
        return self

    def close(self):
        if getattr(self, "h", None):
            self.L_.mcorb_vocab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        k, L, nn, nw = (C.c_int() for _ in range(4))
        _lib.check(self.L_.mcorb_vocab_info(self.h, C.byref(k), C.byref(L), C.byref(nn), C.byref(nw)))
        return dict(k=k.value, L=L.value, nodes=nn.value, words=nw.value)

    def _call(self, fn, head, n, levelsup):
        cap = max(n, 1)
        ids, vals = np.zeros(cap, np.uint32), np.zeros(cap, np.float64)
        nodes, offs, feats = np.zeros(cap, np.uint32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
        nb, nf = C.c_int(), C.c_int()
        _lib.check(fn(*head, levelsup, ids.ctypes.data, vals.ctypes.data, cap, C.byref(nb), nodes.ctypes.data,
                      offs.ctypes.data, cap, C.byref(nf), feats.ctypes.data, cap))
        bow = (ids[:nb.value].copy(), vals[:nb.value].copy())
        fv = {int(nodes[i]): feats[offs[i]:offs[i + 1]].copy() for i in range(nf.value)}
        return bow, fv

    def transform(self, features, levelsup=4):
        """-> BowVector as (word ids ascending, values), FeatureVector as {node id: feature indices}."""
        d = _u8(features).reshape(-1, 32)
        return self._call(self.L_.mcorb_vocab_transform, (self.h, d.ctypes.data, len(d)), len(d), levelsup)

    def match_rig_frame(self, rig, frame=0, slot=0, levelsup=4, max_neighbor_ratio=0.85):
        """computeIntraMatches(matches, words_) (MultiCameraFrame.cpp:586-943): (tracks, n_rays, words)."""
        cap = rig.kcap * rig.ncams
        tr = np.full((cap, rig.ncams), -1, np.int32)
        nr = np.zeros(cap, np.int32)
        words = np.zeros(cap, np.uint32)
        nt, nw = C.c_int(), C.c_int()
        _lib.check(self.L_.mcorb_rig_match_bow(rig.h_rig, slot, frame, self.h, levelsup, max_neighbor_ratio, tr.ctypes.data,
                                               nr.ctypes.data, cap, C.byref(nt), words.ctypes.data, cap, C.byref(nw)))
        return tr[:nt.value].copy(), nr[:nt.value].copy(), words[:nw.value].copy()

    def match_rig_frames(self, rig, frame0, nframes, slot=0, levelsup=4, max_neighbor_ratio=0.85, y_undist=None):
        """computeIntraMatches(matches, words_) for frames [frame0, frame0 + nframes) at once -> list of (tracks, n_rays, words).
        y_undist: optional list, one float32 array per image of the SLOT (index frame * ncams + cam; None entries allowed for
        images outside the range) = image_kps_undist[cam][k].pt.y, the rows the reference's |dy| < 50 gate reads."""
        ptrs, keep = None, []
        if y_undist is not None:
            arr = (C.c_void_p * len(y_undist))()
            for i, y in enumerate(y_undist):
                if y is not None:
                    a = np.ascontiguousarray(y, np.float32)
                    keep.append(a)
                    arr[i] = a.ctypes.data
            ptrs = arr
        _lib.check(self.L_.mcorb_rig_match_bow_frames(rig.h_rig, slot, frame0, nframes, self.h, levelsup, max_neighbor_ratio, ptrs))
        out, cap = [], rig.kcap * rig.ncams
        for f in range(frame0, frame0 + nframes):
            tr = np.full((cap, rig.ncams), -1, np.int32)
            nr = np.zeros(cap, np.int32)
            words = np.zeros(cap, np.uint32)
            nt, nw = C.c_int(), C.c_int()
            _lib.check(self.L_.mcorb_rig_get_bow_tracks(rig.h_rig, slot, f, tr.ctypes.data, nr.ctypes.data, cap, C.byref(nt),
                                                        words.ctypes.data, cap, C.byref(nw)))
            out.append((tr[:nt.value].copy(), nr[:nt.value].copy(), words[:nw.value].copy()))
        return out

    def transform_rig_images(self, rig, img0, nimg, slot=0, levelsup=4):
        """transform() of images [img0, img0 + nimg) of a slot in one call -> list of (BowVector, FeatureVector)."""
        _lib.check(self.L_.mcorb_rig_transform_images(rig.h_rig, slot, img0, nimg, self.h, levelsup))
        out = []
        for m in range(img0, img0 + nimg):
            n = max(rig.L.mcorb_rig_num_keypoints(rig.h_rig, slot, m), 0)
            out.append(self._call(lambda *a, m=m: self.L_.mcorb_rig_get_transform(rig.h_rig, slot, m, *a[1:]), (), n, levelsup))   # (a[0] = levelsup)
        return out

    def transform_rig_image(self, rig, m, slot=0, levelsup=4):
        """transform() of image m's descriptors straight from the rig's HBM buffers (MultiCameraFrame.cpp:257)."""
        n = rig.L.mcorb_rig_num_keypoints(rig.h_rig, slot, m)
        return self._call(self.L_.mcorb_rig_transform_image, (rig.h_rig, slot, m, self.h), max(n, 0), levelsup)


class DescriptorBlock:
    """nsets descriptor sets resident in HBM (mcorb_descblock): upload a keyframe's LF descriptors once, match any two sets with
    Rig.match_sets (findInterMatches' knnMatch, FrontEnd.cpp:3344-3500)."""

    def __init__(self, nsets, kcap, device=0):
        self.L = _lib.load()
        self.h = C.c_void_p()
        _lib.check(self.L.mcorb_descblock_create(device, nsets, kcap, C.byref(self.h)))
        self.nsets, self.kcap = nsets, kcap

    def upload(self, set_index, desc):
        d = _u8(desc).reshape(-1, 32)
        _lib.check(self.L.mcorb_descblock_upload(self.h, set_index, d.ctypes.data, len(d)))

    def close(self):
        if getattr(self, "h", None):
            self.L.mcorb_descblock_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_tracks(ncams, counts, pair_lists):
    """computeIntraMatches' serial track merge (MultiCameraFrame.cpp:1167-1268) on host arrays, no device: counts[c] keypoints
    per camera, pair_lists[p] = (idx1, idx2) of camera pair p in (0,1), (0,2), .., (1,2), .. order -> (tracks [n][ncams], mergeable)."""
    L = _lib.load()
    counts = np.ascontiguousarray(counts, np.int32)
    npairs = ncams * (ncams - 1) // 2
    assert len(pair_lists) == npairs and len(counts) == ncams
    keep = [(np.ascontiguousarray(a, np.uint32), np.ascontiguousarray(b, np.uint32)) for a, b in pair_lists]
    p1, p2 = (C.c_void_p * npairs)(), (C.c_void_p * npairs)()
    for p, (a, b) in enumerate(keep):
        assert len(a) == len(b)
        p1[p], p2[p] = a.ctypes.data, b.ctypes.data
    npair = np.array([len(a) for a, _ in keep], np.int32)
    cap = int(npair.sum()) + 1
    tr = np.full((cap, ncams), -1, np.int32)
    n, mg = C.c_int(), C.c_int()
    _lib.check(L.mcorb_host_merge_tracks(ncams, counts.ctypes.data, p1, p2, npair.ctypes.data, tr.ctypes.data, cap, C.byref(n), C.byref(mg)))
    return tr[:n.value].copy(), mg.value


def fundamental_from_extrinsics(K_i, R_i, t_i, K_j, R_j, t_j):
    """F with x_j^T F x_i = 0 from the two cameras' extrinsics, the construction of MultiCameraFrame.cpp:1126-1142."""
    def T(R, t):
        M = np.eye(4)
        M[:3, :3] = np.asarray(R, np.float64)
        M[:3, 3] = np.asarray(t, np.float64).reshape(3)
        return M
    Tji = T(R_j, t_j) @ np.linalg.inv(T(R_i, t_i))
    tx, ty, tz = Tji[0, 3], Tji[1, 3], Tji[2, 3]
    skew = np.array([[0, -tz, ty], [tz, 0, -tx], [-ty, tx, 0]], np.float64)
    return np.linalg.inv(np.asarray(K_j, np.float64).T) @ skew @ Tji[:3, :3] @ np.linalg.inv(np.asarray(K_i, np.float64))


class IntraMatch:
    """MultiCameraFrame.h:42-57 (matchIndex widened from 5 to ncams entries)."""

    def __init__(self, matchIndex):
        self.matchIndex = list(matchIndex)
        self.mono = True
        self.n_rays = 0


class MultiCameraFrame:
    """Mirror of the extract + intra-rig-match members of MultiCameraFrame
    (MultiCameraFrame.h:70-90) for one rig frame."""

    def __init__(self, ncams, width, height, params=None, rig=None, **kw):
        self.num_cams_ = ncams
        self.rig = rig if rig is not None else Rig(ncams, width, height, 1, 1, params=params, **kw)
        self.imgs = []
        self.image_kps, self.image_kps_undist, self.image_descriptors = [], [], []
        self._matched = False

    def setData(self, img_set, segmap_set=None):
        """setData (MultiCameraFrame.cpp:95-152): accepts the reference's CV_32F [0,1] frames or u8."""
        if len(img_set) != self.num_cams_:
            print("ERROR:: number of images is wrong")   # the reference prints and returns (:98-101)
            return
        self.imgs = list(img_set)
        self.rig.upload(self.imgs)
        self._matched = False

    def extractFeaturesParallel(self):
        """extractFeaturesParallel (MultiCameraFrame.cpp:203-228)."""
        assert self.num_cams_ == len(self.imgs)
        self.rig.extract(self.num_cams_)
        self.image_kps, self.image_descriptors = [], []
        for c in range(self.num_cams_):
            _, k, d = self.rig.features(c)
            self.image_kps.append(k)
            self.image_descriptors.append(d)
        self.image_kps_undist = self.image_kps   # RECTIFY, or zero distortion: UndistortKeyPoints copies (:241-242, :302-305)
        self._matched = False

    extractFeatures = extractFeaturesParallel

    def setUndistorted(self, image_kps_undist):
        """image_kps_undist as UndistortKeyPoints (MultiCameraFrame.cpp:300-347) fills it for a distorted, unrectified rig
        (cv::undistortPoints is the caller's).  Read by BruteForceMatch's returned keypoints, the epipolar check of
        computeIntraMatches(old=True) and the BoW-guided matcher's |dy| < 50 gate.  Call after extractFeaturesParallel()."""
        if len(image_kps_undist) != self.num_cams_ or any(len(u) != len(k) for u, k in zip(image_kps_undist, self.image_kps)):
            raise ValueError("image_kps_undist must hold one entry per extracted keypoint")
        self.image_kps_undist = list(image_kps_undist)

    def _ensure_match(self, dist_thresh, ratio):
        key = (float(dist_thresh), float(ratio))
        if self._matched != key:
            self.rig.match(1, dist_thresh=dist_thresh, ratio=ratio)
            self._matched = key

    def BruteForceMatch(self, img1_ind, img2_ind, dist_thresh, neigh_ratio):
        """BruteForceMatch (MultiCameraFrame.cpp:1024-1086): returns indices_1, indices_2, kps1, kps2."""
        if not img1_ind < img2_ind:
            raise ValueError("the reference calls BruteForceMatch with cam1 < cam2 only")
        self._ensure_match(dist_thresh, neigh_ratio)
        i1, i2 = self.rig.pair_matches(0, img1_ind, img2_ind)
        return i1, i2, self.image_kps_undist[img1_ind][i1], self.image_kps_undist[img2_ind][i2]

    def computeIntraMatchesBoW(self, vocabulary, words_=None, levelsup=4):
        """computeIntraMatches(matches, words_) (MultiCameraFrame.cpp:586-943), the call FrontEnd.cpp:1009 makes.  The
        |dy| < 50 gate reads image_kps_undist (:708-716): set it with setUndistorted() when the rig is not rectified."""
        yu = [np.ascontiguousarray(k["y"], np.float32) for k in self.image_kps_undist]
        tr, nr, words = vocabulary.match_rig_frames(self.rig, 0, 1, levelsup=levelsup, y_undist=yu)[0]
        if words_ is not None:
            words_.extend(int(w) for w in words)
        out = [IntraMatch(row) for row in tr]
        for m, n in zip(out, nr):
            m.n_rays = int(n)
        return out

    def setCalibration(self, K_mats, R_mats, t_mats):
        """camconfig_.K_mats_/R_mats_/t_mats_: builds the per-pair fundamental matrices of
        MultiCameraFrame.cpp:1126-1142 (F = K_j^-T [t_ji]x R_ji K_i^-1 with T_ji = T_j0 T_i0^-1) in float64."""
        self.F_mats = np.stack([fundamental_from_extrinsics(K_mats[i], R_mats[i], t_mats[i], K_mats[j], R_mats[j], t_mats[j])
                                for i in range(self.num_cams_ - 1) for j in range(i + 1, self.num_cams_)])

    def computeIntraMatches(self, old=False, dist_thresh=75.0, ratio=0.85, kps_undist=None):
        """computeIntraMatches(matches, old) (MultiCameraFrame.cpp:1100-1288); old=True applies the epipolar
        check (:1178-1207) and needs setCalibration() or F_mats."""
        self._ensure_match(dist_thresh, ratio)
        if old:
            if getattr(self, "F_mats", None) is None:
                raise ValueError("computeIntraMatches(old=True) needs setCalibration(K, R, t) or F_mats")
            tr, mergeable = self.rig.tracks_epipolar(0, self.F_mats, kps_undist if kps_undist is not None else
                                                     (self.image_kps_undist if self.image_kps_undist is not self.image_kps else None))
        else:
            tr, mergeable = self.rig.tracks(0)
        self.cnt_mergable_matches = mergeable
        return [IntraMatch(row) for row in tr]
