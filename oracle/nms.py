"""ORACLE (test infrastructure only) -- ctypes front-end of oracle/nms_oracle.c plus the python-level ordering logic.

class_agnostic_nms follows /root/reference/pcdet/models/model_utils/model_nms_utils.py:6-25 and
nms_gpu /root/reference/pcdet/ops/iou3d_nms/iou3d_nms_utils.py:84-99.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    so = os.path.join(_HERE, '_build', 'liboracle_nms.so')
    src = os.path.join(_HERE, 'nms_oracle.c')
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '--no-print-directory'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int64)
        L.orc_iou_matrix.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, fp]
        L.orc_overlap_matrix.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, fp]
        L.orc_nms_sorted.argtypes = [fp, ctypes.c_int, ctypes.c_float, ip]
        L.orc_nms_sorted.restype = ctypes.c_int
        L.orc_nms_from_iou.argtypes = [fp, ctypes.c_int, ctypes.c_float, ip]
        L.orc_nms_from_iou.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def iou_matrix(a, b):
    a, pa = _f(a[:, :7])
    b, pb = _f(b[:, :7])
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().orc_iou_matrix(pa, a.shape[0], pb, b.shape[0], out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def overlap_matrix(a, b):
    a, pa = _f(a[:, :7])
    b, pb = _f(b[:, :7])
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().orc_overlap_matrix(pa, a.shape[0], pb, b.shape[0], out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def nms_sorted(boxes_sorted, thresh):
    """iou3d_nms.cpp:90-136 on boxes already in score order -> indices (int64) into that order."""
    b, pb = _f(boxes_sorted[:, :7])
    keep = np.zeros(max(b.shape[0], 1), dtype=np.int64)
    n = lib().orc_nms_sorted(pb, b.shape[0], ctypes.c_float(thresh), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return keep[:n].copy()


def nms_from_iou(iou, thresh):
    iou, pi = _f(iou)
    n0 = iou.shape[0]
    keep = np.zeros(max(n0, 1), dtype=np.int64)
    n = lib().orc_nms_from_iou(pi, n0, ctypes.c_float(thresh), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return keep[:n].copy()


def nms_gpu(boxes, scores, thresh, pre_maxsize=None):
    """iou3d_nms_utils.py:84-99: stable descending sort of the scores, optional pre-max cut, greedy NMS;
    returns indices into `boxes`."""
    order = np.argsort(-scores.astype(np.float32), kind='stable')
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = nms_sorted(boxes[order], thresh)
    return order[keep]


def iou_normal_matrix(a, b):
    """iou3d_nms_kernel.cu:314-325 (iou_normal: axis-aligned x / y extents, heading ignored), float32 operation for operation"""
    a = np.ascontiguousarray(a[:, :7], dtype=np.float32)
    b = np.ascontiguousarray(b[:, :7], dtype=np.float32)
    two, zero = np.float32(2.0), np.float32(0.0)
    ax, ay, adx, ady = (a[:, i][:, None] for i in (0, 1, 3, 4))
    bx, by, bdx, bdy = (b[:, i][None, :] for i in (0, 1, 3, 4))
    left = np.maximum(ax - adx / two, bx - bdx / two)
    right = np.minimum(ax + adx / two, bx + bdx / two)
    top = np.maximum(ay - ady / two, by - bdy / two)
    bottom = np.minimum(ay + ady / two, by + bdy / two)
    inter = np.maximum(right - left, zero) * np.maximum(bottom - top, zero)
    sa, sb = adx * ady, bdx * bdy
    return (inter / np.maximum(sa + sb - inter, np.float32(1e-8))).astype(np.float32)


def nms_normal_gpu(boxes, scores, thresh):
    """iou3d_nms_utils.py:102-117 -> iou3d_nms.cpp:139-188: stable descending score sort, axis-aligned IoU mask, greedy sweep;
    returns indices into `boxes`."""
    order = np.argsort(-scores.astype(np.float32), kind='stable')
    keep = nms_from_iou(iou_normal_matrix(boxes[order], boxes[order]), thresh)
    return order[keep]


def class_agnostic_nms(scores, boxes, thresh, pre_max, post_max, score_thresh=None):
    """model_nms_utils.py:6-25 -> (selected indices into the inputs, selected scores)."""
    src = scores
    idx0 = np.arange(scores.shape[0])
    if score_thresh is not None:
        m = scores >= score_thresh
        scores, boxes, idx0 = scores[m], boxes[m], idx0[m]
    if scores.shape[0] == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.float32)
    k = min(pre_max, scores.shape[0])
    top = np.argsort(-scores, kind='stable')[:k]
    keep = nms_gpu(boxes[top][:, :7], scores[top], thresh)
    sel = idx0[top[keep[:post_max]]]
    return sel.astype(np.int64), src[sel]
