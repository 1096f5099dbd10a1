#!/bin/bash
# usage: tools/ab.sh <variant .so> <out log> [bench args...]  -- base vs variant library, alternating, same box
V=$1; OUT=$2; shift 2
for r in 1 2; do for v in base var; do
  if [ $v = base ]; then L=$PWD/dsp_amd/lib/libdsp.so; else L=$PWD/$V; fi
  echo "== $v $*" >> $OUT
  DSP_AMD_LIBRARY=$L timeout -k 10 200 python bench.py --no-cpu "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])" >> $OUT || exit 1
done; done
