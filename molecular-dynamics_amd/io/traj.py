"""`.xtc` and `.trr` trajectory files through the host codec library (csrc/trajio.cpp).

`PREFIX.xtc` is what `nstxout-compressed` asks for (/root/reference/mdp/production.mdp:12) and
what `gmx rms -f` reads (/root/reference/Snakefile:348); `PREFIX.trr` follows `nstxout/nstvout`
(/root/reference/mdp/nvt_complex.mdp:8-9). Encoding runs outside the interpreter lock, so a
writer thread compresses frames while the main thread keeps the GPU queue filled.
"""
from __future__ import annotations

import ctypes
import os
import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libmdxio.so")
SYMBOLS = ["trj_version", "trj_xtc_frame_bound", "trj_xtc_encode_frame", "trj_xtc_encode_frame_mt", "trj_xtc_encode_frame_q", "trj_xtc_respeculated", "trj_xtc_set_warmup", "trj_xtc_append", "trj_xtc_decode_frame",
           "trj_trr_frame_bound", "trj_trr_encode_frame", "trj_trr_append", "trj_trr_decode_frame"]
_LIB = None


class TrajError(RuntimeError):
    pass


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise TrajError(f"{LIB_PATH} not found: build it with `make -C molecular-dynamics_amd/csrc`")
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_ll, c_f = ctypes.c_int, ctypes.c_longlong, ctypes.c_float
    fp = ctypes.POINTER(c_f)
    ucp = ctypes.POINTER(ctypes.c_ubyte)
    L.trj_xtc_frame_bound.restype = c_ll; L.trj_xtc_frame_bound.argtypes = [c_int]
    L.trj_xtc_encode_frame.restype = c_ll
    L.trj_xtc_encode_frame.argtypes = [ucp, c_ll, c_int, c_ll, c_f, fp, fp, c_f]
    L.trj_xtc_encode_frame_mt.restype = c_ll
    L.trj_xtc_encode_frame_mt.argtypes = [ucp, c_ll, c_int, c_ll, c_f, fp, fp, c_f, c_int]
    L.trj_xtc_encode_frame_q.restype = c_ll
    L.trj_xtc_encode_frame_q.argtypes = [ucp, c_ll, c_int, c_ll, c_f, fp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_f, c_int]
    L.trj_xtc_respeculated.restype = c_ll
    L.trj_xtc_append.argtypes = [ctypes.c_char_p, c_int, c_int, c_ll, c_f, fp, fp, c_f]
    L.trj_xtc_decode_frame.argtypes = [ucp, c_ll, ctypes.POINTER(c_ll), ctypes.POINTER(c_int), ctypes.POINTER(c_ll),
                                       fp, fp, fp, fp]
    L.trj_trr_frame_bound.restype = c_ll; L.trj_trr_frame_bound.argtypes = [c_int]
    L.trj_trr_encode_frame.restype = c_ll
    L.trj_trr_encode_frame.argtypes = [ucp, c_ll, c_int, c_ll, c_f, c_f, fp, fp, fp, fp]
    L.trj_trr_append.argtypes = [ctypes.c_char_p, c_int, c_int, c_ll, c_f, c_f, fp, fp, fp, fp]
    L.trj_trr_decode_frame.argtypes = [ucp, c_ll, ctypes.POINTER(c_ll), ctypes.POINTER(c_int), ctypes.POINTER(c_ll),
                                       fp, fp, fp, fp, fp, fp, ctypes.POINTER(c_int)]
    for s in SYMBOLS:
        getattr(L, s)
    _LIB = L
    return L


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class XtcWriter:
    def __init__(self, path: str, natoms: int, precision: float = 1000.0, append: bool = False, threads: int = 1):
        """threads > 1: large frames are encoded in segments on that many threads (byte-identical output)."""
        self.lib = load_library()
        self.threads = max(1, int(threads))
        self.natoms = int(natoms)
        self.precision = float(precision)
        self.f = open(path, "ab" if append else "wb")
        import threading
        self._tls = threading.local()
        self.frames = 0

    def encode(self, step: int, time: float, box, x) -> bytes:
        """One frame as bytes; safe to call from several threads at once (per-thread buffers)."""
        x = _f32(x); box = _f32(np.asarray(box).reshape(9))
        if x.shape != (self.natoms, 3):
            raise TrajError(f"frame has shape {x.shape}, expected ({self.natoms}, 3)")
        tl = self._tls
        buf = getattr(tl, "buf", None)
        if buf is None:
            buf = tl.buf = (ctypes.c_ubyte * int(self.lib.trj_xtc_frame_bound(self.natoms)))()
        n = self.lib.trj_xtc_encode_frame_mt(buf, len(buf), self.natoms, int(step), float(time), _fptr(box), _fptr(x), self.precision,
                                             self.threads)
        if n < 0:
            raise TrajError(f"xtc encoding failed ({n}): coordinate out of range or not finite")
        return bytes(memoryview(buf)[:n])

    def encode_quantized(self, step: int, time: float, box, ix, stats) -> bytes:
        """The frame from coordinates rounded already (the engine's xtc capture): same bytes as `encode` of the floats."""
        ix = np.ascontiguousarray(ix, dtype=np.int32); stats = np.ascontiguousarray(stats, dtype=np.int32)
        box = _f32(np.asarray(box).reshape(9))
        if ix.shape != (self.natoms, 3):
            raise TrajError(f"frame has shape {ix.shape}, expected ({self.natoms}, 3)")
        tl = self._tls
        buf = getattr(tl, "buf", None)
        if buf is None:
            buf = tl.buf = (ctypes.c_ubyte * int(self.lib.trj_xtc_frame_bound(self.natoms)))()
        ipt = ctypes.POINTER(ctypes.c_int)
        n = self.lib.trj_xtc_encode_frame_q(buf, len(buf), self.natoms, int(step), float(time), _fptr(box), ix.ctypes.data_as(ipt),
                                            stats.ctypes.data_as(ipt), self.precision, self.threads)
        if n < 0:
            raise TrajError(f"xtc encoding failed ({n}): coordinate out of range or not finite")
        return bytes(memoryview(buf)[:n])

    def write_bytes(self, b: bytes):
        self.f.write(b)
        self.frames += 1

    def write(self, step: int, time: float, box, x):
        self.write_bytes(self.encode(step, time, box, x))

    def flush(self):
        self.f.flush()

    def close(self):
        if not self.f.closed:
            self.f.close()


class TrrWriter:
    def __init__(self, path: str, natoms: int, append: bool = False):
        self.lib = load_library()
        self.natoms = int(natoms)
        self.f = open(path, "ab" if append else "wb")
        self.buf = (ctypes.c_ubyte * int(self.lib.trj_trr_frame_bound(self.natoms)))()
        self.frames = 0

    def write(self, step: int, time: float, box, x=None, v=None, f=None, lam: float = 0.0):
        box = _f32(np.asarray(box).reshape(9))
        x = _f32(x) if x is not None else None
        v = _f32(v) if v is not None else None
        f = _f32(f) if f is not None else None
        n = self.lib.trj_trr_encode_frame(self.buf, len(self.buf), self.natoms, int(step), float(time), float(lam), _fptr(box),
                                          _fptr(x), _fptr(v), _fptr(f))
        if n < 0:
            raise TrajError(f"trr encoding failed ({n})")
        self.f.write(memoryview(self.buf)[:n])
        self.frames += 1

    def flush(self):
        self.f.flush()

    def close(self):
        if not self.f.closed:
            self.f.close()


def read_xtc(path: str):
    """-> list of dict(step, time, box (3,3), x (N,3), precision)."""
    L = load_library()
    data = np.fromfile(path, dtype=np.uint8)
    base = data.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))
    pos, out = 0, []
    while pos < len(data):
        used, nat, step = ctypes.c_longlong(0), ctypes.c_int(0), ctypes.c_longlong(0)
        t, prec = ctypes.c_float(0), ctypes.c_float(0)
        box = np.zeros(9, np.float32)
        p = ctypes.cast(ctypes.addressof(base.contents) + pos, ctypes.POINTER(ctypes.c_ubyte))
        rc = L.trj_xtc_decode_frame(p, len(data) - pos, ctypes.byref(used), ctypes.byref(nat), ctypes.byref(step),
                                    ctypes.byref(t), _fptr(box), None, ctypes.byref(prec))
        if rc != 0:
            raise TrajError(f"{path}: bad xtc frame at byte {pos} ({rc})")
        x = np.zeros((nat.value, 3), np.float32)
        rc = L.trj_xtc_decode_frame(p, len(data) - pos, ctypes.byref(used), ctypes.byref(nat), ctypes.byref(step),
                                    ctypes.byref(t), _fptr(box), _fptr(x), ctypes.byref(prec))
        if rc != 0:
            raise TrajError(f"{path}: bad xtc frame at byte {pos} ({rc})")
        out.append(dict(step=int(step.value), time=float(t.value), box=box.reshape(3, 3).copy(), x=x, precision=float(prec.value)))
        pos += used.value
    return out


def read_trr(path: str):
    """-> list of dict(step, time, lambda, box, x, v, f) (absent blocks are None)."""
    L = load_library()
    data = np.fromfile(path, dtype=np.uint8)
    base = data.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))
    pos, out = 0, []
    while pos < len(data):
        used, nat, step = ctypes.c_longlong(0), ctypes.c_int(0), ctypes.c_longlong(0)
        t, lam = ctypes.c_float(0), ctypes.c_float(0)
        box = np.zeros(9, np.float32)
        has = (ctypes.c_int * 3)()
        p = ctypes.cast(ctypes.addressof(base.contents) + pos, ctypes.POINTER(ctypes.c_ubyte))
        rc = L.trj_trr_decode_frame(p, len(data) - pos, ctypes.byref(used), ctypes.byref(nat), ctypes.byref(step), ctypes.byref(t),
                                    ctypes.byref(lam), _fptr(box), None, None, None, has)
        if rc != 0:
            raise TrajError(f"{path}: bad trr frame at byte {pos} ({rc})")
        arr = [np.zeros((nat.value, 3), np.float32) if has[k] else None for k in range(3)]
        rc = L.trj_trr_decode_frame(p, len(data) - pos, ctypes.byref(used), ctypes.byref(nat), ctypes.byref(step), ctypes.byref(t),
                                    ctypes.byref(lam), _fptr(box), _fptr(arr[0]), _fptr(arr[1]), _fptr(arr[2]), has)
        if rc != 0:
            raise TrajError(f"{path}: bad trr frame at byte {pos} ({rc})")
        out.append(dict(step=int(step.value), time=float(t.value), lam=float(lam.value), box=box.reshape(3, 3).copy(),
                        x=arr[0], v=arr[1], f=arr[2]))
        pos += used.value
    return out
