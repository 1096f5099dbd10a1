#!/usr/bin/env python3
"""Diagnostic: a bank of C channels on a 64K block as ONE launch, against the same channels as banks of <= 1024
channels launched back to back (what splitting a launch into resident rounds by hand would give)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from tests import util
_, b, a = util.design_coefs(sys.argv[1] if len(sys.argv) > 1 else "bw4_bp_0.3_0.6")     # e.g. bw8_bp_0.25_0.75: order 16
s = torch.cuda.current_stream()
N, RING = 65536, 4
d = gbank.iir_design(b, a)
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for k in range(reps):
        fn(k)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for C in (1024, 1280, 1536, 2048, 2560, 3072, 4096, 6144):
    x = torch.empty(RING * C * N, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    gbank.synth_fill(x.view(RING * C, N), 1)
    one = gbank.FilterBank([d], [0] * C)
    parts, c0 = [], 0
    while c0 < C:
        n = min(1024, C - c0)
        parts.append((c0, gbank.FilterBank([d], [0] * n)))
        c0 += n
    def f_one(k):
        base = (k % RING) * C * N * 8
        one.updaten_ptr(y.data_ptr() + base, N, x.data_ptr() + base, N, N, s)
    def f_split(k):
        base = (k % RING) * C * N * 8
        for c0, bk in parts:
            bk.updaten_ptr(y.data_ptr() + base + c0 * N * 8, N, x.data_ptr() + base + c0 * N * 8, N, N, s)
    timed(f_one, 100); t1 = timed(f_one, 100)
    timed(f_split, 100); t2 = timed(f_split, 100)
    print("%5d channels: one launch %7.1f us = %.0f GB/s   launches of <= 1024: %7.1f us = %.0f GB/s   ratio %.3f"
          % (C, t1, C * N * 16 / t1 / 1e3, t2, C * N * 16 / t2 / 1e3, t2 / t1), flush=True)
    one.close()
    for _, bk in parts: bk.close()
    del x, y
