"""Deterministic inputs shared by make_golden.py (reference side) and the tests (our side).

Values come from exact integer hashing mapped to dyadic rationals, so they are bit-identical on
every platform / numpy version and need not be stored next to the expected outputs.
"""
import numpy as np


def det_uniform(n, seed):
    """n float64 values in [0, 1) on a 2^-24 lattice; exact integer arithmetic only."""
    i = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15 & 0xFFFFFFFF)
    h = (i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return (h >> np.uint64(8)).astype(np.float64) / float(1 << 24)


def make_metric_inputs(T, ncol=384):
    """(N,124) normalised-looking inputs, (N,128) scaled targets and predictions, float32."""
    n = T * ncol
    x = (det_uniform(n * 124, 1).reshape(n, 124) - 0.5).astype(np.float32)
    y = ((det_uniform(n * 128, 2).reshape(n, 128) - 0.5) * 0.1).astype(np.float32)
    y[:, 120:] = np.abs(y[:, 120:])
    y[:, 60:72] = 0.0
    p = (y + (det_uniform(n * 128, 3).reshape(n, 128).astype(np.float32) - 0.5) * 0.02).astype(np.float32)
    p[:, 120:] = np.maximum(p[:, 120:], 0.0)
    return x, y, p


def make_crps_inputs(T=4, ncol=384, nlev=60, ns=8):
    sp3 = det_uniform(T * ncol * nlev * ns, 4).reshape(T, ncol, nlev, ns)
    t3 = det_uniform(T * ncol * nlev, 5).reshape(T, ncol, nlev)
    sp2 = det_uniform(T * ncol * ns, 6).reshape(T, ncol, ns)
    t2 = det_uniform(T * ncol, 7).reshape(T, ncol)
    return sp3, t3, sp2, t2


def make_cnn_inputs(n=96):
    xi = det_uniform(n * 124, 8).reshape(n, 124).astype(np.float32)
    yi = det_uniform(n * 128, 9).reshape(n, 128).astype(np.float32)
    return xi, yi


def subsample(arr):
    """Strided view used to pin large arrays: every 7th row (time*col flattened or axis 0/1) and
    every 3rd level -- coprime strides so all columns/levels classes are visited."""
    arr = np.asarray(arr)
    if arr.ndim == 3:
        return arr[:, ::7, ::3].copy()
    if arr.ndim == 2:
        return arr[::7, ::3].copy()
    return arr.copy()
