#!/bin/bash
# usage: scripts_prof.sh <tag> [bench args...]   -- rocprofv3 kernel stats + PMC passes for bench.py
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu "$@" > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --no-cpu "$@" > $OUT/pmc_sq.log 2>&1 || { echo "sq pass failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu "$@" > $OUT/pmc_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu "$@" > $OUT/pmc_write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/pmc_write.log; exit 1; }
cd $R
python3 $R/scripts_prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merged-back directory small: drop the raw per-dispatch CSVs of the PMC passes
find $OUT -name "*.csv" -size +4M -delete
