#!/bin/bash
# kernel trace of a sequence of frames in flight (experiment): tools/r03_inflight_trace.sh <workload> <frames> <in flight>
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_inflight_$1 -o t -- python3 $GRAFT_REPO_ROOT/tools/pipeline_time.py $1 $2 $3 > $GRAFT_REPO_ROOT/gpurun_out/trace_inflight_$1.log 2>&1 || { tail -n 20 $GRAFT_REPO_ROOT/gpurun_out/trace_inflight_$1.log; exit 1; }
grep "ms/frame" $GRAFT_REPO_ROOT/gpurun_out/trace_inflight_$1.log
ls -la $GRAFT_REPO_ROOT/gpurun_out/trace_inflight_$1
