"""ctypes declarations for libmcorb.so (include/mcorb.h).

The library is the product: HIP kernels + host engine behind a C ABI.  There is
no Python or CPU fallback -- if the shared object is missing this module raises,
and if no gfx950 device is usable the C calls return MCORB_E_NODEVICE.
"""
import ctypes as C
import os
import sys

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libmcorb.so")

OK, E_EMPTY, E_SIZE, E_CAP, E_ARG, E_HIP, E_NODEVICE, E_STATE, E_OVERFLOW = 0, -1, -2, -3, -4, -5, -6, -7, -8
ORIENT_NONE, ORIENT_IC_ANGLE = 0, 1
MAX_LEVELS, MAX_CAMS = 16, 16

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int), ("orientation", C.c_int),
                ("device_id", C.c_int), ("host_threads", C.c_int), ("cand_cap", C.c_int), ("selection", C.c_int),
                ("gpu_jobs", C.c_int), ("reserved", C.c_int * 5)]


class Camera(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("Rt", C.c_double * 12)]


LF_DTYPE = np.dtype([("match_index", "<i4", (MAX_CAMS,)), ("uv_ref", "<f4", (2,)), ("mono", "<i4"), ("n_rays", "<i4"),
                     ("point3d", "<f8", (3,)), ("desc", "u1", (32,))])


class McorbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mcorb error %d: %s" % (code, msg))
        self.code = code


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol include/mcorb.h declares
SIGNATURES = {
    "mcorb_default_params": (None, [C.POINTER(Params)]),
    "mcorb_last_error": (C.c_char_p, []),
    "mcorb_version": (C.c_char_p, []),
    "mcorb_device_count": (_i, []),
    "mcorb_rig_create": (_i, [C.POINTER(Params), _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "mcorb_rig_destroy": (None, [_vp]),
    "mcorb_rig_upload_u8": (_i, [_vp, _i, C.POINTER(_vp), _i, _i]),
    "mcorb_rig_upload_f32": (_i, [_vp, _i, C.POINTER(_vp), _i, _i, _i]),
    "mcorb_rig_extract_submit": (_i, [_vp, _i, _i, _i, _i]),
    "mcorb_rig_extract_wait": (_i, [_vp, _i]),
    "mcorb_rig_extract": (_i, [_vp, _i, _i, _i, _i]),
    "mcorb_rig_staging": (_i, [_vp, _i, _i, _vp, _ip]),
    "mcorb_rig_upload_staged": (_i, [_vp, _i, _i]),
    "mcorb_rig_process_submit": (_i, [_vp, _i, _i, _i, _i, _f, _f]),
    "mcorb_rig_process": (_i, [_vp, _i, _i, _i, _i, _f, _f]),
    "mcorb_rig_process_wait": (_i, [_vp, _i]),
    "mcorb_rig_num_keypoints": (_i, [_vp, _i, _i]),
    "mcorb_rig_get_features": (_i, [_vp, _i, _i, _vp, _vp, _i, _ip, _ip]),
    "mcorb_rig_match": (_i, [_vp, _i, _i, _f, _f]),
    "mcorb_rig_match_submit": (_i, [_vp, _i, _i, _f, _f]),
    "mcorb_rig_match_wait": (_i, [_vp, _i]),
    "mcorb_rig_get_pair_matches": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _ip]),
    "mcorb_rig_get_pair_knn2": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _ip]),
    "mcorb_rig_get_tracks": (_i, [_vp, _i, _i, _vp, _i, _ip, _ip]),
    "mcorb_rig_level_size": (_i, [_vp, _i, _ip, _ip]),
    "mcorb_rig_get_level": (_i, [_vp, _i, _i, _i, _vp, _i]),
    "mcorb_rig_get_blurred": (_i, [_vp, _i, _i, _i, _vp, _i]),
    "mcorb_rig_get_candidates": (_i, [_vp, _i, _i, _i, _vp, _i, _ip]),
    "mcorb_rig_last_timing": (_i, [_vp, _i, C.POINTER(_f)]),
    "mcorb_rig_select_mode": (_i, [_vp]),
    "mcorb_rig_select_fallbacks": (_i, [_vp, _i]),
    "mcorb_rig_early_reads_rejected": (_i, [_vp, _i]),
    "mcorb_rig_set_graph": (_i, [_vp, _i]),
    "mcorb_dev_sort_selftest": (_i, [_i, _vp, _i, _vp, _vp]),
    "mcorb_rig_match_pairs_external": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _f, _f]),
    "mcorb_rig_match_pairs_external_dev_submit": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _f, _f, _vp]),
    "mcorb_rig_get_pairlist": (_i, [_vp, _i, _i, _vp, _vp, _i, _ip]),
    "mcorb_host_merge_tracks": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _ip, _ip]),
    "mcorb_descblock_create": (_i, [_i, _i, _i, _vp]),
    "mcorb_descblock_destroy": (None, [_vp]),
    "mcorb_descblock_upload": (_i, [_vp, _i, _vp, _i]),
    "mcorb_descblock_desc_ptr": (_vp, [_vp]),
    "mcorb_descblock_counts_dev": (_vp, [_vp]),
    "mcorb_rig_match_sets": (_i, [_vp, _i, _vp, _vp, _i, _f, _f]),
    "mcorb_rig_get_pairknn2": (_i, [_vp, _i, _i, _vp, _vp, _i, _ip]),
    "mcorb_rig_kcap": (_i, [_vp]),
    "mcorb_rig_host_threads": (_i, [_vp]),
    "mcorb_rig_info": (_i, [_vp, _vp]),
    "mcorb_rig_desc_device_ptr": (_vp, [_vp, _i]),
    "mcorb_rig_stream": (_vp, [_vp, _i]),
    "mcorb_rig_export_descriptors": (_i, [_vp, _i, _vp, _vp, _i]),
    "mcorb_rig_match_external": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _f, _f]),
    "mcorb_rig_match_external_submit": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _f, _f]),
    "mcorb_rig_export_descriptors_dev": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "mcorb_rig_match_external_dev_submit": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _f, _f, _vp]),
    "mcorb_create": (_i, [C.POINTER(Params), _i, _i, C.POINTER(_vp)]),
    "mcorb_destroy": (None, [_vp]),
    "mcorb_extract": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _ip, _ip]),
    "mcorb_extract_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _ip, _ip]),
    "mcorb_get_tables": (_i, [C.POINTER(Params), _vp, _vp, _vp, _vp, _vp]),
    "mcorb_get_pyramid_level": (_i, [_vp, _i, _vp, _i, _ip, _ip]),
    "mcorb_hamming256": (_i, [_vp, _vp]),
    "mcorb_representative_desc": (_i, [_vp, _i]),
    "mcorb_knn2": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "mcorb_match_ratio": (_i, [_vp, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _i, _ip]),
    "mcorb_vocab_create": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(_vp)]),
    "mcorb_vocab_load_text": (_i, [C.c_char_p, _i, C.POINTER(_vp)]),
    "mcorb_vocab_destroy": (None, [_vp]),
    "mcorb_vocab_info": (_i, [_vp, _ip, _ip, _ip, _ip]),
    "mcorb_vocab_transform": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _ip, _vp, _vp, _i, _ip, _vp, _i]),
    "mcorb_rig_transform_image": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _ip, _vp, _vp, _i, _ip, _vp, _i]),
    "mcorb_rig_get_tracks_epipolar": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _ip, _ip]),
    "mcorb_rig_match_bow": (_i, [_vp, _i, _i, _vp, _i, C.c_double, _vp, _vp, _i, _ip, _vp, _i, _ip]),
    "mcorb_rig_transform_images": (_i, [_vp, _i, _i, _i, _vp, _i]),
    "mcorb_rig_get_transform": (_i, [_vp, _i, _i, _vp, _vp, _i, _ip, _vp, _vp, _i, _ip, _vp, _i]),
    "mcorb_rig_match_bow_frames": (_i, [_vp, _i, _i, _i, _vp, _i, C.c_double, _vp]),
    "mcorb_rig_get_bow_tracks": (_i, [_vp, _i, _i, _vp, _vp, _i, _ip, _vp, _i, _ip]),
    "mcorb_rig_obtain_lf_features": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _ip, _ip, _ip, _vp, _i, _ip]),
    "mcorb_rig_obtain_lf_features_frames": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "mcorb_host_select": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i]),
    "mcorb_host_resize_axis": (_i, [_i, _i, _i, _vp]),
    "mcorb_host_triangulate": (_i, [_vp, _vp, _i, _vp]),
    "mcorb_host_geometry": (_i, [C.POINTER(Params), _i, _i, _vp]),
    "mcorb_synth_rig_frame": (_i, [C.c_uint32, _i, _i, _i, _i, _vp, _i]),
}

_lib = None
LOADED_BEFORE_TORCH = False   # libmcorb.so (and with it /opt/rocm's HIP runtime) entered the process before torch did


def require_torch_first(what):
    """The stream-ordered multi-GPU calls exchange device pointers and HIP stream handles with torch (tensors of the collective,
    torch.cuda.Stream().cuda_stream).  That only works when both sides use ONE HIP runtime, which is the case when torch -- which
    ships its own libamdhip64 -- is imported BEFORE libmcorb.so is loaded: libmcorb then binds to the runtime
    already in the process.  The other order gives two runtimes whose handles mean nothing to each other; instead of letting
    that surface as a hang or an 'invalid resource handle' three calls later, refuse here."""
    if LOADED_BEFORE_TORCH and "torch" in sys.modules:
        raise RuntimeError("mc-slam_amd: %s was called with a torch-owned stream / tensor, but libmcorb.so was loaded before torch was "
                           "imported in this process: `import torch` first, then `import mcorb` (INTEGRATION.md section 5)" % what)


def load():
    """Load libmcorb.so; raises if it has not been built (no fallback path exists)."""
    global _lib, LOADED_BEFORE_TORCH
    if _lib is not None:
        return _lib
    LOADED_BEFORE_TORCH = "torch" not in sys.modules
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmcorb.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` or `make -C mc-slam_amd/csrc` (hipcc, gfx950). There is no CPU fallback."
                          % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)   # AttributeError here means the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(code):
    if code != OK:
        raise McorbError(code, load().mcorb_last_error().decode("utf-8", "replace"))
    return code


def default_params(**kw):
    p = Params()
    load().mcorb_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p
