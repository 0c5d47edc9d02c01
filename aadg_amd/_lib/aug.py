"""The fused uint8 augmentation call (csrc/aug_u8.hip): unit records, the host-side planner's Python statement, the launch wrapper and its profiling taps.
Reference: data/basic.py, data/policy.py:45-61, data/transform.py:97-236."""
import ctypes
import os

import numpy as np
import torch

from .binding import AadgError, DATASET_OPTIC, MAX_OPS, UNIT_DTYPE, _check, _require_cuda, _stream, _vp, load, workspace


# ------------------------------------------------------------------------------------------------
def units_to_device(units, device):
    """numpy UNIT_DTYPE[N] -> uint8 device tensor [N,140]."""
    units = np.ascontiguousarray(units, dtype=UNIT_DTYPE)
    host = torch.from_numpy(units.view(np.uint8).reshape(units.shape[0], UNIT_DTYPE.itemsize))
    return host.to(device, non_blocking=False)


def validate_units(units, P, Hs, Ws):
    """Host-side argument checks the kernels rely on (raises like the reference's asserts would)."""
    units = np.asarray(units)
    if units.dtype != UNIT_DTYPE:
        raise AadgError("units must have UNIT_DTYPE")
    if units.shape[0] == 0:
        raise AadgError("empty unit list")
    if (units["src"] < 0).any() or (units["src"] >= P).any():
        raise AadgError("unit.src out of range")
    if (units["n_ops"] < 0).any() or (units["n_ops"] > MAX_OPS).any():
        raise AadgError("unit.n_ops out of range (CONTROLLER.L <= %d)" % MAX_OPS)
    if (units["scaled_w"] * 3 < Ws).any() or (units["scaled_h"] * 3 < Hs).any() or \
            (units["scaled_w"] < 1).any() or (units["scaled_h"] < 1).any():
        raise AadgError("scale factor below 1/3 is not supported by the 8-tap resampler")
    for k in range(MAX_OPS):
        live = units["n_ops"] > k
        ops = units["op"][:, k][live]
        if ((ops < 0) | (ops > 9)).any():
            raise AadgError("unknown op id")
        r = units["rect"][:, k][live & (units["op"][:, k] == 9)]
        if r.size and ((r[:, 0] < 0).any() or (r[:, 1] < 0).any() or (r[:, 2] >= Ws).any() or (r[:, 3] >= Hs).any()):
            raise AadgError("cutout rectangle must be clipped to the image")
        b = units["iarg"][:, k][live & (units["op"][:, k] == 4)]
        if b.size and ((b < 0).any() or (b > 8).any()):
            raise AadgError("posterize bits out of range")
    return int(units["n_ops"].max())


def launch_plan(units, Hs, Ws, crop):
    """(classes, stats_mask, order, counts, stat_lists, late, n_stat_stencil, n_generic_wonly, (n_plain_early, n_sharp_early)) for aadg_aug_u8_forward_ex2 -- mirrors unit_flow() in
    csrc/aug_u8.hip.  n_generic_wonly (ABI 9): the first ones of the generic run shrink the width only and chain no stencil (one-pass tile).
    order: unit indices grouped by tile class (plain up-scaling, up-scaling with a Sharpness stencil, generic, staged);
    counts = (n_plain, n_sharp, n_generic, n_generic_sharp: the last ones of the generic run chain a Sharpness stencil); stat_lists[k]: the units whose k-th op needs a pixel pass for its image statistics;
    late: the units with such an op in a slot k >= 1."""
    n_ops = units["n_ops"]
    live = np.arange(MAX_OPS)[None, :] < n_ops[:, None]
    sharp = ((units["op"] == 8) & (units["farg"] != np.float32(1.0)) & live).sum(axis=1)
    ok = sharp <= 2
    if (Ws & 3) or (crop & 3):
        ok[:] = False
    up = ok & (units["scaled_w"] >= Ws) & (units["scaled_h"] >= Hs)
    generic = ok & ~up & (2 * units["scaled_w"] >= Ws) & (2 * units["scaled_h"] >= Hs)
    staged = ~(up | generic)
    classes = (1 if up.any() else 0) | (2 if staged.any() else 0) | (4 if generic.any() else 0)
    needs = np.isin(units["op"], (0, 2, 5)) & live
    # statistics by push-forward (csrc/aug_u8.hip: stats_by_pushforward): AutoContrast / Equalize in slot k >= 1 behind per-channel
    # byte maps only, fused-flow units -- k_lut derives that stage's histogram from the RAW image's, no pixel pass
    lut_class = np.isin(units["op"], (0, 1, 2, 3, 4, 5, 7))
    prefix_lut = np.ones_like(live)
    for k in range(1, MAX_OPS):
        prefix_lut[:, k] = prefix_lut[:, k - 1] & lut_class[:, k - 1]
    push = live & np.isin(units["op"], (0, 2)) & prefix_lut & (up | generic)[:, None]
    push[:, 0] = False
    pixel_pass = needs & ~push
    pixel_pass[:, 0] |= push.any(axis=1)                   # the raw histogram is the source of every push-forward
    stats_mask = 0
    for k in range(MAX_OPS):
        if pixel_pass[:, k].any():
            stats_mask |= 1 << k
    # tile classes in list order: up-scaling plain / with a stencil, down-scaling ("generic") plain / with a stencil, staged
    # (among the generic units without a stencil those that shrink the width only come first: k_fused3w's list, ABI 9)
    wonly = generic & (sharp == 0) & (units["scaled_h"] >= Hs) & (Ws >= 8)
    cls = np.where(up & (sharp == 0), 0, np.where(up, 1, np.where(wonly, 2, np.where(generic & (sharp == 0), 3, np.where(generic, 4, 5)))))
    # (ABI 12: inside the plain and the Sharpness up-scaling class the late units -- a slot k >= 1 needs a pixel pass -- come last)
    late_flag = pixel_pass[:, 1:].any(axis=1)
    order = np.argsort(cls * 2 + (late_flag & (cls < 2)), kind="stable").astype(np.int32)
    counts = (int((cls == 0).sum()), int((cls == 1).sum()), int(((cls == 2) | (cls == 3) | (cls == 4)).sum()), int((cls == 4).sum()))
    # work lists of the histogram kernels; slot k's list starts with the units that have a Sharpness stencil among ops [0, k) (ABI 7:
    # aadg_aug_lists.n_stat_stencil -- their tiles get a workgroup each), both parts in ascending unit order
    stencil = (units["op"] == 8) & (units["farg"] != np.float32(1.0)) & live
    stat_lists, n_stencil = [], []
    for k in range(MAX_OPS):
        idx = np.nonzero(pixel_pass[:, k])[0]
        before = stencil[idx, :k].any(axis=1) if k > 0 else np.zeros(idx.size, bool)
        stat_lists.append(np.concatenate([idx[before], idx[~before]]).astype(np.int32))
        n_stencil.append(int(before.sum()))
    # "late" units: a slot k >= 1 needs a pixel pass (include/aadg_hip.h: aadg_aug_lists.late_units)
    late = np.nonzero(pixel_pass[:, 1:].any(axis=1))[0].astype(np.int32)
    return (classes, stats_mask, order, counts, stat_lists, late, n_stencil, int((cls == 2).sum()),
            (int((~late_flag & (cls == 0)).sum()), int((~late_flag & (cls == 1)).sum())))


def launch_hints(units, Hs, Ws, crop):
    """launch_plan without the late list: (classes, stats_mask, order, counts, stat_lists)."""
    return launch_plan(units, Hs, Ws, crop)[:5]


class AugLists(ctypes.Structure):
    """mirror of `aadg_aug_lists` (include/aadg_hip.h): host struct of device index arrays"""
    _fields_ = [("order", ctypes.c_void_p), ("n_plain", ctypes.c_int32), ("n_sharp", ctypes.c_int32), ("n_generic", ctypes.c_int32),
                ("stat_units", ctypes.c_void_p * MAX_OPS), ("n_stat", ctypes.c_int32 * MAX_OPS), ("pool_hist", ctypes.c_void_p),
                ("late_units", ctypes.c_void_p), ("n_late", ctypes.c_int32), ("n_generic_sharp", ctypes.c_int32),
                ("n_stat_stencil", ctypes.c_int32 * MAX_OPS), ("gen_chunk", ctypes.c_int32), ("n_generic_wonly", ctypes.c_int32),
                ("n_plain_early", ctypes.c_int32), ("n_sharp_early", ctypes.c_int32)]


HIST_STRIDE = 772      # AADG_HIST_STRIDE


def pool_histograms(pool):
    """uint32 [P, HIST_STRIDE] statistics of the source pool (uint8 [P,H,W,3], device): per channel histogram + sum of L.
    The policy ops run on the raw source image, so these serve every unit / batch that draws the image: compute once per
    resident pool and pass as `pool_hist` to aug_u8_forward (recompute after writing to the pool)."""
    lib = load()
    _require_cuda(pool)
    if pool.dtype != torch.uint8 or pool.dim() != 4 or pool.shape[3] != 3 or not pool.is_contiguous():
        raise AadgError("pool must be contiguous uint8 [P,H,W,3]")
    P, Hs, Ws, _ = pool.shape
    hist = torch.empty((P, HIST_STRIDE), dtype=torch.int32, device=pool.device)
    _check(lib.aadg_pool_histograms_u8(pool.data_ptr(), P, Hs, Ws, hist.data_ptr(), _stream()), "aadg_pool_histograms_u8")
    return hist


# optional (start, stop) torch.cuda.Event pair recorded around the dominant kernel of the next
# aug_u8_forward call(s); used by bench.py to time that kernel live on the launch stream
# False: the library keeps the whole call on the caller's stream (A/B of the helper-stream fork, ABI 12)
AUG_FORK = True
PROFILE_EVENTS = None
# optional list: every aug_u8_forward call appends the op mix of its units (bench.py: the tile kernel's duration follows it)
PROFILE_MIX = None
# optional (start, stop) torch.cuda.Event pair recorded on the current stream around the WHOLE library call (all its kernels)
PROFILE_CALL_EVENTS = None
_pinned = {}


_REC = UNIT_DTYPE.itemsize + 4 * (2 + MAX_OPS)   # staging bytes per unit: the record + its slot in the class-order list and in each
                                                 # stage's statistics work list


def _pinned_units(n):
    buf = _pinned.get("units")
    if buf is None or buf.numel() < n * _REC:
        buf = torch.empty(max(n, 256) * _REC, dtype=torch.uint8).pin_memory()
        _pinned["units"] = buf
    return buf


def aug_u8_forward(pool, masks, units, crop, dataset, out_img=None, out_lbl=None, pool_hist=None, gen_chunk=0):
    """pool u8 [P,Hs,Ws,3], masks u8 [P,Hs,Ws] (device), units numpy UNIT_DTYPE[N]; pool_hist: pool_histograms(pool) or None
    (None: the statistics passes of the call read the source images themselves); gen_chunk: aadg_aug_lists.gen_chunk (0 = default).
    Returns (aug_images f32 [N,3,crop,crop], aug_labels f32 [N,K,crop,crop]) on the device."""
    lib = load()
    _require_cuda(pool, masks)
    if pool.dtype != torch.uint8 or masks.dtype != torch.uint8 or pool.dim() != 4 or pool.shape[3] != 3:
        raise AadgError("pool must be uint8 [P,H,W,3] and masks uint8 [P,H,W]")
    P, Hs, Ws, _ = pool.shape
    if tuple(masks.shape) != (P, Hs, Ws):
        raise AadgError("masks shape must match pool")
    units = np.asarray(units)
    if units.dtype != UNIT_DTYPE:
        raise AadgError("units must have UNIT_DTYPE")
    N = units.shape[0]
    if N == 0:
        raise AadgError("empty unit list")
    K = 2 if dataset == DATASET_OPTIC else 1
    dev = pool.device
    if out_img is None:
        out_img = torch.empty((N, 3, crop, crop), dtype=torch.float32, device=dev)
    if out_lbl is None:
        out_lbl = torch.empty((N, K, crop, crop), dtype=torch.float32, device=dev)
    _require_cuda(out_img, out_lbl)
    # units: host records -> pinned staging -> async H2D on the launch stream (no host sync)
    units = np.ascontiguousarray(units)
    stage = _pinned_units(N)
    ready = _pinned.get("units_ready")
    if ready is not None:
        ready.synchronize()          # previous copy out of the staging buffer has completed
    nb_units = N * UNIT_DTYPE.itemsize                       # a multiple of 4: the int32 lists behind it are aligned
    host = stage[:N * _REC].numpy()
    host[:nb_units] = units.view(np.uint8).reshape(-1)
    # validation + work lists (tile-class order, per-slot statistics lists, late list) by the library's host-side planner, written
    # straight into the staging buffer behind the records: [order N][stat_units MAX_OPS x N][late N] int32
    base = stage.data_ptr()
    summary = (ctypes.c_int32 * (11 + 2 * MAX_OPS))()
    rc = lib.aadg_aug_u8_plan(base, N, P, Hs, Ws, crop, base + nb_units, base + nb_units + 4 * N, base + nb_units + 4 * N * (1 + MAX_OPS), summary)
    if rc != 0:
        validate_units(units, P, Hs, Ws)                     # raises with the reason
        _check(rc, "aadg_aug_u8_plan")
    n_plain, n_sharp, n_generic, n_generic_sharp, n_late, classes, stats_mask, max_ops = summary[:8]
    d_units = torch.empty(N * _REC, dtype=torch.uint8, device=dev)
    lists = AugLists()
    lists.order = d_units.data_ptr() + nb_units
    lists.n_plain, lists.n_sharp, lists.n_generic, lists.n_generic_sharp = n_plain, n_sharp, n_generic, n_generic_sharp
    lists.n_generic_wonly = summary[8 + 2 * MAX_OPS]
    lists.n_plain_early, lists.n_sharp_early = (summary[9 + 2 * MAX_OPS], summary[10 + 2 * MAX_OPS]) if AUG_FORK else (0, 0)     # (0, 0): one stream
    lists.gen_chunk = int(gen_chunk)
    for k in range(MAX_OPS):
        lists.stat_units[k] = d_units.data_ptr() + nb_units + 4 * N * (1 + k)
        lists.n_stat[k] = summary[8 + k]
        lists.n_stat_stencil[k] = summary[8 + MAX_OPS + k]
    if pool_hist is not None:
        if pool_hist.dtype != torch.int32 or tuple(pool_hist.shape) != (P, HIST_STRIDE) or not pool_hist.is_cuda:
            raise AadgError("pool_hist must be pool_histograms(pool): int32 [P, %d] on the device" % HIST_STRIDE)
        lists.pool_hist = pool_hist.data_ptr()
        lists.late_units = d_units.data_ptr() + nb_units + 4 * N * (1 + MAX_OPS)
        lists.n_late = n_late
    d_units.copy_(stage[:N * _REC], non_blocking=True)          # records + work lists: one H2D copy
    ready = torch.cuda.Event()
    ready.record()
    _pinned["units_ready"] = ready
    nb = lib.aadg_aug_u8_workspace_bytes(N, Hs, Ws, crop)
    ws = workspace(nb, dev, "aug")
    ev0 = ev1 = 0
    if PROFILE_MIX is not None:
        live = np.arange(MAX_OPS)[None, :] < units["n_ops"][:, None]
        PROFILE_MIX.append({"units": int(N), "ops": int(live.sum()), "sharpness_ops": int(((units["op"] == 8) & live).sum()),
                            "sharpness_units": int(n_sharp), "stat_ops": int(sum(summary[8:8 + MAX_OPS])), "late_units": int(n_late),
                            "upscaled": int(((units["scaled_w"] != Ws) | (units["scaled_h"] != Hs)).sum()),
                            # units per tile kernel: k_fused3 (plain + Sharpness), k_fused3w (width-only down-scaling), the two passes
                            "tile_units": {"k_fused3": int(n_plain + n_sharp), "k_fused3w": int(lists.n_generic_wonly),
                                           "two_pass": int(n_generic - lists.n_generic_wonly), "two_pass_with_sharpness": int(n_generic_sharp)}})
    if PROFILE_EVENTS is not None:
        ev0, ev1 = PROFILE_EVENTS[0].cuda_event, PROFILE_EVENTS[1].cuda_event
    if PROFILE_CALL_EVENTS is not None:
        PROFILE_CALL_EVENTS[0].record()
    rc = lib.aadg_aug_u8_forward_ex2(pool.data_ptr(), masks.data_ptr(), P, Hs, Ws, d_units.data_ptr(), N, max_ops, crop,
                                     dataset, out_img.data_ptr(), out_lbl.data_ptr(), ws.data_ptr(), ws.numel(), _stream(),
                                     classes, stats_mask, ev0, ev1, ctypes.byref(lists))
    if PROFILE_CALL_EVENTS is not None:
        PROFILE_CALL_EVENTS[1].record()
    _check(rc, "aadg_aug_u8_forward")
    d_units.record_stream(torch.cuda.current_stream())
    return out_img, out_lbl


def op_u8(img, op, iarg=0, farg=0.0, rect=None):
    """One registry op on a uint8 HWC device image."""
    lib = load()
    _require_cuda(img)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise AadgError("img must be uint8 [H,W,3]")
    H, W, _ = img.shape
    out = torch.empty_like(img)
    r = (ctypes.c_int32 * 4)(*(rect if rect is not None else (0, 0, -1, -1)))
    nb = lib.aadg_aug_u8_workspace_bytes(1, H, W, 0)
    ws = workspace(nb, img.device, "op")
    rc = lib.aadg_op_u8(img.data_ptr(), out.data_ptr(), H, W, int(op), int(iarg), float(farg),
                        ctypes.cast(r, _vp), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_op_u8")
    return out
