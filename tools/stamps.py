#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the scan kernel (DSP_AMD_SCAN_STAMPS build path)."""
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
L, B = 1 << 19, 1 << 16
bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_, mode=gbank.MODE_SCAN)
x = torch.empty((C_, L), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
gbank.synth_fill(x, 1)
for j in range(L // B):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    bank.updaten_ptr(y.data_ptr() + j * B * 8, L, x.data_ptr() + j * B * 8, L, B)
    e1.record()
    torch.cuda.synchronize()
print("channels %d: last stamped kernel %.1f us" % (C_, e0.elapsed_time(e1) * 1e3))
buf = np.zeros(C_ * 8, dtype=np.uint64)
dspdll.dsp_debug_scan_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert dspdll.dsp_debug_scan_stamps(bank._handle, buf.ctypes.data, buf.size) == 0
st = buf.reshape(C_, 8).astype(np.float64)
rt = st[:, 7].copy()
beg, end = st[:, 5].copy(), st[:, 6].copy()
t0 = beg.min()
print("WG start (us after first): p50 %.1f p90 %.1f max %.1f ; WG end: min %.1f p50 %.1f max %.1f ; WG life mean %.1f us" % (
    np.percentile(beg - t0, 50) / 100, np.percentile(beg - t0, 90) / 100, (beg - t0).max() / 100,
    (end - t0).min() / 100, np.percentile(end - t0, 50) / 100, (end - t0).max() / 100, (end - beg).mean() / 100))
life = (end - beg) / 100
print("WG life by blockIdx%8 (XCD group):", [round(float(life[i::8].mean()), 1) for i in range(8)])
print("WG life spread inside one XCD group: min %.1f max %.1f std %.1f" % (life[0::8].min(), life[0::8].max(), life[0::8].std()))
print("WG life by (blockIdx//8)%32:", [round(float(life.reshape(-1, 8)[i::32].mean()), 1) for i in range(32)])
order = np.argsort(beg)
print("start times of WGs by channel id (first 16):", ((beg - t0)[:16] / 100).round(1).tolist())
print("start-time histogram (10 us bins):", np.histogram((beg - t0) / 100, bins=np.arange(0, 330, 10))[0].tolist())
st[:, 5:] = 0
tot = st.sum(axis=1)
print("in-kernel clock: %.3f GHz (s_memtime / s_memrealtime x 100 MHz), span %.1f us" % ((tot / rt).mean() * 0.1, rt.mean() / 100))
names = ["barrier after park", "prefetch issue", "pass1", "combine", "pass2(+start exch)", "carry+store issue",
         "park (vmcnt wait + ds_write)", "barrier after store"]
print("last block, per workgroup: mean total cycles %.0f" % tot.mean())
for i, nme in enumerate(names):
    print("  %-30s mean %9.0f  (%.1f%%)   min %9.0f max %9.0f" % (nme, st[:, i].mean(), 100 * st[:, i].mean() / tot.mean(), st[:, i].min(), st[:, i].max()))
