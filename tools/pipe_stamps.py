#!/usr/bin/env python3
"""Diagnostic: timeline of the pipelined IIR kernel's workgroups.  Needs the stamped library build:
  tools/variant.sh stamps -DDSP_PIPE_STAMPS && DSP_AMD_LIBRARY=$PWD/tools/_build/libdsp_stamps.so python tools/pipe_stamps.py

Stamps are 100 MHz s_memrealtime: [0] entry, [1] set-up done (front), [2] segment 0 published,
[3] back starts segment 0, [5] back done.  usage: pipe_stamps.py [channels] [n]"""
import ctypes as C
import os
import sys
os.environ["DSP_AMD_SCAN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from dsp_amd.cdsp import dspdll
from tests import util

_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dspdll.dsp_debug_scan_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for n in [int(v) for v in sys.argv[2:]] or [2048, 16384, 65536]:
    LD = n + 128
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_)
    x = torch.empty((C_, LD), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    gbank.synth_fill(x, 1)
    for rep in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        bank.updaten_ptr(y.data_ptr(), LD, x.data_ptr(), LD, n)
        e1.record()
        torch.cuda.synchronize()
    buf = np.zeros(C_ * 32, dtype=np.uint64)
    assert dspdll.dsp_debug_scan_stamps(bank._handle, buf.ctypes.data, buf.size) == 0
    st = buf.reshape(C_, 32).astype(np.float64) / 100.0    # us
    t0 = st[:, 0].min()
    print("n=%d channels=%d: event time %.1f us; stamped span (first entry -> last back done) %.1f us" % (
        n, C_, e0.elapsed_time(e1) * 1e3, st[:, 5].max() - t0))
    def q(name, v):
        print("   %-34s min %6.1f  p50 %6.1f  p90 %6.1f  max %6.1f" % (name, v.min(), np.percentile(v, 50), np.percentile(v, 90), v.max()))
    q("WG entry after first entry", st[:, 0] - t0)
    q("entry -> set-up done", st[:, 1] - st[:, 0])
    q("set-up done -> seg 0 published", st[:, 2] - st[:, 1])
    q("published -> back starts seg 0", st[:, 3] - st[:, 2])
    q("back starts seg 0 -> back done", st[:, 5] - st[:, 3])
    q("  front: set-up done -> barrier 1", st[:, 7] - st[:, 1])
    q("  front: zero-state pass", st[:, 8] - st[:, 7])
    q("  front: prefix + publish", st[:, 2] - st[:, 8])
    q("  back: entry -> set-up done", st[:, 9] - st[:, 0])
    q("  back: seg 0 recurrences", st[:, 10] - st[:, 3])
    if n >= 10 * 2048:
        q("  steady it=8 front: barriers+tile", st[:, 12] - st[:, 11])
        q("  steady it=8 front: zero-state", st[:, 13] - st[:, 12])
        q("  steady it=8 front: prefix+publish", st[:, 14] - st[:, 13])
        q("  steady: back seg 8 start - front it=8 b1", st[:, 15] - st[:, 12])
        q("  steady front: it=8 b1 -> it=9 top", st[:, 22] - st[:, 12])
        q("  steady front: it=9 top -> it=9 b1 (wait+park)", st[:, 23] - st[:, 22])
        q("  steady front period (b1 -> b1)", st[:, 23] - st[:, 12])
        q("  steady back: seg 8 b1 -> recurrences done", st[:, 18] - st[:, 15])
        q("  steady back: -> stores issued, next top", st[:, 19] - st[:, 18])
        q("  steady back: next top -> seg 9 b1 (wait)", st[:, 21] - st[:, 19])
        q("  steady back period (b1 -> b1)", st[:, 21] - st[:, 15])
    if n >= 10 * 2048:
        clk = (st[:, 17] - st[:, 16]) / (st[:, 14] - st[:, 0])   # (cycles/100) per us = 1e-2 cycles/us
        q("  shader clock GHz (entry -> it=8)", clk * 0.1)
    q("WG life", st[:, 5] - st[:, 0])
    q("WG end after first entry", st[:, 5] - t0)
    raw = buf.reshape(C_, 32)
    hw = raw[:, 4].astype(np.int64); xcc = (raw[:, 6].astype(np.int64)) & 15
    cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
    life = st[:, 5] - st[:, 0]
    print("   life by XCC:", {int(k): round(float(life[xcc == k].mean()), 1) for k in np.unique(xcc)})
    print("   WGs by XCC :", {int(k): int((xcc == k).sum()) for k in np.unique(xcc)})
    print("   life by SE :", {int(k): round(float(life[se == k].mean()), 1) for k in np.unique(se)})
    print("   life by CU id:", {int(k): round(float(life[cu == k].mean()), 1) for k in np.unique(cu)})
    key = xcc * 4096 + se * 64 + sh * 16 + cu
    cnt = np.array([(key == k).sum() for k in np.unique(key)])
    print("   distinct CUs %d; WGs per CU: min %d max %d" % (len(cnt), cnt.min(), cnt.max()))
    spread = np.array([life[key == k].max() - life[key == k].min() for k in np.unique(key)])
    print("   life spread within one CU: mean %.1f max %.1f; life by WGs-per-CU:" % (spread.mean(), spread.max()),
          {int(c): round(float(np.mean([life[key == k].mean() for k in np.unique(key) if (key == k).sum() == c])), 1) for c in np.unique(cnt)})
    print("   life by blockIdx quartile:", [round(float(life[i * C_ // 4:(i + 1) * C_ // 4].mean()), 1) for i in range(4)])
    del bank
