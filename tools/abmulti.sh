#!/bin/bash
# usage: tools/abmulti.sh OUT ROUNDS variant...   -- base + variants (tools/_build/libdsp_<v>.so), alternating on the
# same box; prints per library min / median / max Msamples/s and the copy ceiling measured beside each run
OUT=$1; R=$2; shift 2
: > $OUT.raw
for r in $(seq $R); do for v in base "$@"; do
  if [ $v = base ]; then L=$PWD/dsp_amd/lib/libdsp.so; else L=$PWD/tools/_build/libdsp_$v.so; fi
  DSP_AMD_LIBRARY=$L timeout -k 10 200 python bench.py --no-cpu $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['roofline'].get('copy_ceiling_measured') or 0))" >> $OUT.raw || exit 1
done; done
python - $OUT.raw > $OUT <<'PY'
import sys, statistics as st
rows = {}
for line in open(sys.argv[1]):
    v, val, ceil = line.split()
    rows.setdefault(v, []).append((int(val), int(ceil)))
for v, r in rows.items():
    vals = sorted(x[0] for x in r)
    print("%-12s min %7d  median %7d  max %7d   (copy ceiling GB/s: %s)" % (v, vals[0], st.median(vals), vals[-1], " ".join(str(x[1]) for x in r)))
PY
cat $OUT
