#!/bin/bash
# two PMC passes over the C4 frame (tools/time_workload.py), coop_kernel and direct_kernel rows: tools/c4_pmc.sh <tag>
export TMPDIR=/tmp
OUT=gpurun_out/c4pmc_${1:-x}; mkdir -p $OUT
W=many2048_1080p_32spp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python3 tools/time_workload.py $W > $OUT/a.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- python3 tools/time_workload.py $W > $OUT/b.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -o c -- python3 tools/time_workload.py $W > $OUT/c.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "coop_kernel" not in r["Kernel_Name"] and "direct_kernel" not in r["Kernel_Name"]: continue
        k=r["Kernel_Name"][9:50]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE": agg[k]["_ms"]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
        agg[k]["_vgpr"]=float(r["VGPR_Count"])*len(n[k]); agg[k]["_scratch"]=float(r["Scratch_Size"])*len(n[k])
    for k,v in agg.items():
        print(k, {c: "%.4g"%(x/len(n[k])) for c,x in sorted(v.items())})
PY
