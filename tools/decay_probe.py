"""Decay length and scan-safety of butterworth(order, wl, wu) designs as RTFilterStream sees them.  usage: decay_probe.py 9,0,0.1 5,0.3,0.5 ..."""
import sys
sys.path.insert(0, '.')
import numpy as np
from dsp_amd import dsp, bank as gbank, stream as gstream
cands = [tuple(float(v) if i else int(v) for i, v in enumerate(a.split(','))) for a in sys.argv[1:]] or [(8, 0.25, 0.75)]
for args in cands:
    f = dsp.butterworth(*args)
    b, a = np.array(f.b()), np.array(f.a())
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    print(args, 'order', len(a) - 1, 'decay', fs.decay_length, 'exact', fs.exact_split)
    fs.close()
