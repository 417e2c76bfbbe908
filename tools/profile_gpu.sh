#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + PMC passes of bench.py, one rocprofv3 run per pass
# (counters are not split over passes; --pmc is never combined with other trace domains).
#   tools/profile_gpu.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"

run() {  # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $BENCH > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc" >> "$OUT/status.txt"
    return $rc
}

: > "$OUT/status.txt"
run trace --kernel-trace --stats || exit 1
run pmc_a --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
run pmc_b --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
run pmc_c --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_LEVEL_WAVES || exit 1
run pmc_d --pmc FETCH_SIZE || exit 1
run pmc_e --pmc WRITE_SIZE || exit 1
find "$OUT" -name "*.csv" | head -40
# per-frame totals for bench.py's roofline object (profiles/pmc_<workload>.json): tools/pmc_frame.py <tag> <workload>
