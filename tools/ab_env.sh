#!/bin/bash
# usage: ab_env.sh OUT ROUNDS ENVVAR  -- alternate bench.py with and without ENVVAR=1
OUT=$1; R=$2; V=$3
: > $OUT
for r in $(seq $R); do for m in on off; do
  if [ $m = off ]; then export $V=1; else unset $V; fi
  timeout -k 10 200 python bench.py --no-cpu $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['value']), round(d['roofline'].get('copy_ceiling_measured') or 0), round(d['roofline'].get('kernel_ms_avg',0)*1000,1))" >> $OUT || exit 1
done; done
cat $OUT
