"""Shared helpers for the parity tests."""
import numpy as np


def rgba(word1):
    return [int(word1 & 0xFF), int((word1 >> 8) & 0xFF), int((word1 >> 16) & 0xFF), int(word1 >> 24)]


def same_bits_or_nan(a, b):
    """float arrays equal bit for bit where finite/inf, and NaN exactly where the other is NaN
    (NaN payload/sign differs between x86 and gfx950)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def describe_mismatch(a, b, limit=5):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        return "shape %s vs %s" % (a.shape, b.shape)
    idx = np.argwhere(a != b)
    return "%d mismatches, first: %s" % (len(idx), [(tuple(i), a[tuple(i)], b[tuple(i)]) for i in idx[:limit]])


def random_cloud(rng, n, lo=-0.95, hi=0.95, nan_every=0, dup_frac=0.0):
    pts = (rng.random((n, 3)) * (hi - lo) + lo).astype(np.float32)
    if dup_frac > 0 and n > 4:
        k = int(n * dup_frac)
        src = rng.integers(0, n, k)
        dst = rng.integers(0, n, k)
        pts[dst] = pts[src]
    if nan_every:
        pts[::nan_every, 0] = np.nan
        pts[1::nan_every * 2, 2] = np.inf
    col = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    return pts, col


def surface_cloud(rng, n, jitter=0.01):
    """points on a wavy sheet + a sphere: surface-like occupancy, as depth images produce"""
    u = rng.random(n) * 1.8 - 0.9
    v = rng.random(n) * 1.8 - 0.9
    z = 0.3 * np.sin(3 * u) * np.cos(2 * v) + 0.2
    pts = np.stack([u, v, z], 1)
    k = n // 3
    d = rng.normal(size=(k, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts[:k] = d * 0.4 + np.array([0.1, -0.2, -0.3])
    pts += rng.normal(scale=jitter, size=pts.shape)
    col = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    return pts.astype(np.float32), col
