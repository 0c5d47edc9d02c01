"""ctypes loader for the CPU oracle (oracle/aadg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under aadg_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_OPS = 4

# must match `orc_unit` in aadg_oracle.c (and `aadg_unit` in include/aadg_hip.h)
UNIT_DTYPE = np.dtype([
    ("src", "<i4"), ("n_ops", "<i4"),
    ("op", "<i4", (MAX_OPS,)), ("iarg", "<i4", (MAX_OPS,)), ("farg", "<f4", (MAX_OPS,)),
    ("rect", "<i4", (MAX_OPS, 4)),
    ("scaled_w", "<i4"), ("scaled_h", "<i4"), ("pad", "<i4"), ("crop_x", "<i4"), ("crop_y", "<i4"),
], align=False)
assert UNIT_DTYPE.itemsize == 140


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "aadg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_sinkhorn_divergence.restype = ctypes.c_float
        _LIB.orc_sinkhorn_divergence_f64.restype = ctypes.c_double
        _LIB.orc_epsilon_schedule.restype = ctypes.c_int
    return _LIB


def _p(a, t=ctypes.c_void_p):
    return a.ctypes.data_as(t)


def op_u8(img, op, iarg=0, farg=0.0, rect=None):
    """One selectable op on an HWC uint8 RGB image."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, _ = img.shape
    out = np.empty_like(img)
    r = np.asarray(rect if rect is not None else [0, 0, -1, -1], dtype=np.int32)
    lib().orc_op_u8(_p(img), _p(out), H, W, int(op), int(iarg), ctypes.c_float(farg), _p(r))
    return out


def resize_bilinear(img, w, h):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, _ = img.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_resize_bilinear_u8(_p(img), H, W, _p(out), h, w)
    return out


def resize_nearest(mask, w, h):
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    H, W = mask.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_resize_nearest_u8(_p(mask), H, W, _p(out), h, w)
    return out


def aug_units(src, masks, units, crop, dataset_kind):
    """src [S,Hs,Ws,3] u8, masks [S,Hs,Ws] u8, units UNIT_DTYPE[N] -> (img [N,3,c,c] f32, lbl [N,K,c,c] f32)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    units = np.ascontiguousarray(units, dtype=UNIT_DTYPE)
    S, Hs, Ws, _ = src.shape
    N = units.shape[0]
    K = 2 if dataset_kind == 0 else 1
    oi = np.empty((N, 3, crop, crop), np.float32)
    ol = np.empty((N, K, crop, crop), np.float32)
    lib().orc_aug_units(_p(src), _p(masks), S, Hs, Ws, _p(units), N, crop, dataset_kind, _p(oi), _p(ol))
    return oi, ol


def epsilon_schedule(diameter, blur=0.05, scaling=0.5, p=2.0):
    buf = np.empty(128, np.float64)
    n = lib().orc_epsilon_schedule(ctypes.c_double(diameter), ctypes.c_double(blur), ctypes.c_double(scaling),
                                   ctypes.c_double(p), _p(buf), 128)
    return buf[:n].copy()


def sinkhorn_divergence(x, y, blur=0.05, scaling=0.5, f64=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    fn = lib().orc_sinkhorn_divergence_f64 if f64 else lib().orc_sinkhorn_divergence
    return float(fn(_p(x), x.shape[0], _p(y), y.shape[0], x.shape[1], ctypes.c_double(blur), ctypes.c_double(scaling)))


def sinkhorn_rewards(fe, D, B, M, blur=0.05, scaling=0.5, rewards=None):
    fe = np.ascontiguousarray(fe, dtype=np.float32)
    assert fe.shape[0] == D * B * M
    r = np.zeros(M, np.float32) if rewards is None else np.ascontiguousarray(rewards, dtype=np.float32)
    lib().orc_sinkhorn_rewards(_p(fe), D, B, M, fe.shape[1], ctypes.c_double(blur), ctypes.c_double(scaling), _p(r))
    return r


def normalize_rewards(r):
    r = np.ascontiguousarray(r, dtype=np.float32)
    out = np.empty_like(r)
    lib().orc_normalize_rewards(_p(r), r.shape[0], _p(out))
    return out


def policy_bce(logits, labels, M):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    labels = np.ascontiguousarray(labels, dtype=np.float32)
    N, K = logits.shape[:2]
    HW = int(np.prod(logits.shape[2:]))
    out = np.empty(M, np.float64)
    lib().orc_policy_bce(_p(logits), _p(labels), N, K, HW, M, _p(out))
    return out


def dice(logits, labels):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    labels = np.ascontiguousarray(labels, dtype=np.float32)
    N, K = logits.shape[:2]
    HW = int(np.prod(logits.shape[2:]))
    out = np.empty(K, np.float64)
    lib().orc_dice(_p(logits), _p(labels), N, K, HW, _p(out))
    return out
