"""Shared helpers for the test-suite: golden loaders and the synthetic input generator."""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def unhex(hs):
    return np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in hs], dtype=np.float64)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth(seed, chan, n0, n):
    """x[c][n] = (int64(splitmix64(seed ^ (c<<32 + n)) >> 11)) * 2^-52 - 1, in [-1, 1)."""
    with np.errstate(over="ignore"):
        idx = (np.uint64(chan) << np.uint64(32)) + np.arange(n0, n0 + n, dtype=np.uint64)
        h = _splitmix64(np.uint64(seed) ^ idx)
    return (h >> np.uint64(11)).astype(np.int64).astype(np.float64) * 2.0 ** -52 - 1.0


def design_coefs(name):
    """(kind, b, a, init) of a golden design; kind in {"iir", "fir", "fir_partial"}."""
    d = load_json("designs.json")[name]
    if d["kind"] in ("fir", "fir_partial"):
        return d["kind"], unhex(d["b"]), None
    return "iir", unhex(d["b"]), unhex(d["a"])


def bits_equal(p, q):
    p = np.ascontiguousarray(p, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    return p.shape == q.shape and p.tobytes() == q.tobytes()


def rel_to_peak(y, yref):
    """max|y - yref| / max|yref| along the last axis (the parity metric of BASELINE.md 3)."""
    y = np.asarray(y, dtype=np.float64)
    yref = np.asarray(yref, dtype=np.float64)
    peak = np.max(np.abs(yref), axis=-1)
    peak = np.where(peak == 0, 1.0, peak)
    return np.max(np.abs(y - yref), axis=-1) / peak
