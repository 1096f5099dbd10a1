#!/usr/bin/env python3
"""Randomized sweep of RTFilterStream_updaten against the oracle: Butterworth designs of orders 4..16 (band-pass and
low-pass), sequences of calls of random lengths (2^16 .. 2^27 + ragged tails) on one stream, out of place and in
place.  Prints the split the library chose.  usage: fuzz_stream.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import dsp, bank as gbank, stream as gstream
from oracle import oracle as orc
from tests import util

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    order = int(rng.integers(4, 17))
    if order % 2 == 0 and rng.random() < 0.6:
        lo = 0.15 + 0.2 * rng.random()
        f = dsp.butterworth(order // 2, lo, lo + 0.25 + 0.25 * rng.random())
    else:
        f = dsp.butterworth(order, 0.0, 0.25 + 0.45 * rng.random())
    b, a = np.array(f.b()), np.array(f.a())
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    ref = orc.iir(b, a)
    pos, worst, splits = 0, 0.0, []
    for k in range(int(rng.integers(1, 4))):
        n = int(2 ** rng.uniform(16, 27.3)) + int(rng.integers(0, 5000))
        x = util.synth(500 + c, 0, pos, n)
        want = ref.updaten(x)
        xd = torch.from_numpy(x).cuda()
        if rng.random() < 0.25:
            fs.updaten_ptr(xd.data_ptr(), xd.data_ptr(), n)
            got = xd.cpu().numpy()
        else:
            got = fs(xd).cpu().numpy()
        splits.append((n, fs.last_split))
        worst = max(worst, float(np.max(np.abs(got - want)) / np.max(np.abs(want))))
        pos += n
    st, fv, ini = fs.get_state()
    serr = float(np.max(np.abs(st[:len(ref.state)] - ref.state)) / max(1.0, float(np.max(np.abs(ref.state)))))
    ok = worst <= 1e-12 and serr <= 1e-12 and abs(fv - ref.filtered_value) <= 1e-12 * max(1.0, abs(ref.filtered_value))
    bad += not ok
    print("case %2d order %2d decay %5d calls %s: err %.2e state %.2e%s" % (
        c, order, fs.decay_length, splits, worst, serr, "" if ok else "   <-- FAIL"), flush=True)
    fs.close()
print("failures", bad)
