"""Deterministic synthetic rig frames.

`synth_rig_frame` calls the C generator inside libmcorb.so (csrc/mcorb_synth.c);
`synth_rig_frame_numpy` is its NumPy mirror, used by the CPU test-suite to check
the generator itself and as a generator when the library has not been built.
"""
import numpy as np

DISPARITY = 24
_M64 = (1 << 64) - 1


def _xs64star(s):
    s ^= s >> 12
    s ^= (s << 25) & _M64
    s ^= s >> 27
    return s, (s * 0x2545F4914F6CDD1D) & _M64


def synth_rig_frame_numpy(frame, ncams, cam, w, h):
    cw = w + DISPARITY * (ncams - 1)
    canvas = np.full((h, cw), 128, np.uint8)
    seed = 0x4D435F53 ^ int(frame)
    s = seed
    nrect = (6000 * w * h) // 921600

    def uni(n):
        nonlocal s
        s, r = _xs64star(s)
        return (r >> 33) % n

    for _ in range(nrect):
        rw = 6 + uni(55)
        rh = 6 + uni(55)
        x0 = uni(cw)
        y0 = uni(h)
        g = uni(256)
        canvas[y0:min(y0 + rh, h), x0:min(x0 + rw, cw)] = g
    nseed = np.uint64((seed * 0x9E3779B97F4A7C15) & _M64)
    xoff = DISPARITY * cam
    yy, xx = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64) + np.uint64(xoff), indexing="ij")
    with np.errstate(over="ignore"):
        z = yy * np.uint64(cw) + xx + nseed
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    n = ((z >> np.uint64(33)) % np.uint64(13)).astype(np.int32) - 6
    v = canvas[:, xoff:xoff + w].astype(np.int32) + n
    return np.clip(v, 0, 255).astype(np.uint8)


def synth_rig_frame(frame, ncams, cam, w, h):
    from . import _lib
    out = np.zeros((h, w), np.uint8)
    rc = _lib.load().mcorb_synth_rig_frame(int(frame), ncams, cam, w, h, out.ctypes.data, w)
    if rc != 0:
        raise ValueError("mcorb_synth_rig_frame: bad arguments")
    return out
