#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the one-wave scan kernel (orders 9..16 and MODE_SCAN), DSP_AMD_SCAN_STAMPS=1.

usage: scan_stamps.py [design key] [channels] [samples per channel]"""
import ctypes as C
import os
import sys
os.environ["DSP_AMD_SCAN_STAMPS"] = "1"
os.environ["DSP_AMD_SCAN_VARIANT"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from dsp_amd.cdsp import dspdll
from tests import util

key = sys.argv[1] if len(sys.argv) > 1 else "bw8_bp_0.25_0.75"
C_ = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 19
_, b, a = util.design_coefs(key)
bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_, mode=gbank.MODE_SCAN)
x = torch.empty((C_, L), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
gbank.synth_fill(x, 1)
for rep in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    bank.updaten_ptr(y.data_ptr(), L, x.data_ptr(), L, L)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print("%s order %d, %d channels x %d: %.3f ms = %.1f Gsamples/s = %.2f TB/s" % (key, len(a) - 1, C_, L, ms, C_ * L / ms / 1e6, C_ * L * 16 / ms / 1e9))
buf = np.zeros(C_ * 32, dtype=np.uint64)
dspdll.dsp_debug_scan_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert dspdll.dsp_debug_scan_stamps(bank._handle, buf.ctypes.data, buf.size) == 0
st = buf[:C_ * 8].reshape(C_, 8).astype(np.float64)[:, :6]
tot = st.sum(axis=1)
nseg = L // (64 * 64)
names = ["park (vmcnt wait + ds_write)", "prefetch issue", "pass 1 (zero-state)", "combine (prefix)", "pass 2 (recurrence)", "store"]
print("per workgroup: %.0f cycles per segment (%d segments)" % (tot.mean() / nseg, nseg))
for i, nme in enumerate(names):
    print("  %-30s %9.0f cycles/segment (%.1f%%)" % (nme, st[:, i].mean() / nseg, 100 * st[:, i].mean() / tot.mean()))
