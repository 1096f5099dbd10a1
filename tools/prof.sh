#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   -- rocprofv3 kernel stats + PMC passes for bench.py
# (each PMC pass is a run of its own, without trace domains; FETCH_SIZE and WRITE_SIZE in separate passes)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {   # pass <dir> <rocprofv3 options...> : bench args appended
    local d=$1; shift
    timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d $OUT/$d -- python3 $R/bench.py --no-cpu "${BENCH_ARGS[@]}" > $OUT/$d.log 2>&1 \
        || { echo "$d pass failed"; tail -5 $OUT/$d.log; exit 1; }
}
BENCH_ARGS=("$@")
pass stats --kernel-trace --stats
pass pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
pass pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
pass pmc_write --pmc WRITE_SIZE
if [ "${PROF_MFMA:-0}" = "1" ]; then
    pass pmc_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
fi
cd $R
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merged-back directory small: drop the raw per-dispatch CSVs of the PMC passes
find $OUT -name "*.csv" -size +4M -delete
