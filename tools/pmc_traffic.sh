#!/bin/bash
# HBM traffic per kernel launch (two PMC passes: FETCH_SIZE, WRITE_SIZE) over tools/time_workload.py:
#   tools/pmc_traffic.sh <tag> <workload> [flags]     bytes = 2 x FETCH_SIZE KB (gfx950 correction) + WRITE_SIZE KB
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmct_$TAG; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o a -- python3 tools/time_workload.py "$@" > $OUT/$c.log 2>&1 || exit 1
done
python3 - <<PY
import csv,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(set))
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for r in csv.DictReader(open("$OUT/%s/a_counter_collection.csv" % c)):
        if "rt::" not in r["Kernel_Name"] and "rt_relaxed::" not in r["Kernel_Name"]: continue
        k=r["Kernel_Name"].replace("void ","")[:70]+"|grid"+r["Grid_Size"]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k,v in sorted(agg.items(), key=lambda kv:-(2*kv[1]["FETCH_SIZE"]+kv[1]["WRITE_SIZE"]))[:8]:
    f=v["FETCH_SIZE"]/max(1,len(n[k]["FETCH_SIZE"])); w=v["WRITE_SIZE"]/max(1,len(n[k]["WRITE_SIZE"]))
    print("%s launches %d  fetch %.1f MB (x2 corrected)  write %.1f MB  per launch" % (k, len(n[k]["FETCH_SIZE"]), 2*f*1024/1e6, w*1024/1e6))
PY
