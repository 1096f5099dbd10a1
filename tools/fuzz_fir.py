#!/usr/bin/env python3
"""Ad-hoc robustness sweep of the FIR kernels: random tap counts, channel counts, block-length sequences and
stable / partial initialisation under AUTO (matrix cores from 17 taps) and SEQUENTIAL (bit-exact) against the oracle.
usage: fuzz_fir.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dsp_amd import bank as gbank
from tests import util, gpu_util as G

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    nd = int(rng.choice([1, 1, 2, 3]))                       # designs of different lengths share one launch group
    Ks = [int(rng.choice([1, 2, 3, 8, 16, 17, 31, 32, 33, 64, 100, 255, 256, 257, 512, 700, 754, 1024, 2000, 4096]))
          for _ in range(nd)]
    K = max(Ks)
    partial = bool(rng.random() < 0.3)
    nch = int(rng.choice([1, 3, 64, 130])) if K <= 1024 else int(rng.choice([1, 3, 9]))
    if K <= 255 and rng.random() < 0.5:
        nch = int(rng.choice([512, 600, 1100]))              # wide enough for the streaming kernel (fir_tile_kernel)
    blocks = [int(rng.choice([1, 7, 100, 2048, 4095, 4096, 4097, 4096 + 5, 8191, 8192, 8192 + 16, 12288 + 2047, 16384, 20000]))
              for _ in range(int(rng.integers(2, 5)))]
    mode = gbank.MODE_SEQUENTIAL if rng.random() < 0.3 else gbank.MODE_AUTO
    ds = [gbank.fir_design(rng.standard_normal(k) / max(1, k) ** 0.5, partial=partial and i == 0) for i, k in enumerate(Ks)]
    cmap = [int(rng.integers(0, nd)) for _ in range(nch)]
    ys_g, ys_o, gb, ob = G.run_pair(ds, cmap, blocks, mode, seed=300 + c)
    yg, yo = np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1)
    if mode == gbank.MODE_SEQUENTIAL:
        ok = util.bits_equal(yg, yo); err = 0.0 if ok else float(np.max(util.rel_to_peak(yg, yo)))
    else:
        err = float(np.max(util.rel_to_peak(yg, yo))); ok = err <= 1e-12
    st, fv, ini = gb.get_state()
    sok = util.bits_equal(st[:, :ob.state.shape[1]], ob.state) and np.array_equal(ini, ob.init)
    if not (ok and sok): bad += 1
    print("case %2d K %4d partial %d nch %3d blocks %s mode %d: err %.2e state_bits %s%s" % (
        c, K, partial, nch, blocks, mode, err, sok, "" if ok and sok else "   <-- FAIL"), flush=True)
    gb.close()
print("failures", bad)
