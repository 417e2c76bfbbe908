#!/bin/bash
# one PMC pass (occupancy / VALU counters) over tools/time_workload.py: tools/pmc_quick.sh <tag> <workload> [flags]
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmcq_$TAG; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python3 tools/time_workload.py "$@" > $OUT/a.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open("$OUT/a/a_counter_collection.csv")):
    if "rt::" not in r["Kernel_Name"] and "rt_relaxed::" not in r["Kernel_Name"]: continue
    k=r["Kernel_Name"].replace("void ","")[:70]+"|grid"+r["Grid_Size"]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    if r["Counter_Name"]=="GRBM_GUI_ACTIVE": agg[k]["_ms"]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for k,v in sorted(agg.items(), key=lambda kv:-kv[1]["_ms"])[:8]:
    c=len(n[k]); cyc=v["GRBM_GUI_ACTIVE"]/c/8
    print("%s calls %d ms %.2f waves %.0f waves/SIMD %.2f VALUutil %.1f%% laneutil %.1f%% wait %.0f%%" % (k, c, v["_ms"]/c, v["SQ_WAVES"]/c, 4*v["SQ_WAVE_CYCLES"]/c/cyc/1024, 100*v["SQ_INSTS_VALU"]/c/(1024*cyc/2), 100*v["SQ_THREAD_CYCLES_VALU"]/max(1.0,64*v["SQ_ACTIVE_INST_VALU"]), 100*v["SQ_WAIT_ANY"]/max(1.0,v["SQ_WAVE_CYCLES"])), "valu_winstr %.3e" % (v["SQ_INSTS_VALU"]/c))
PY
