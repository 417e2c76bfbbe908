#!/bin/bash
# PMC counters of the cooperative kernel on the mixed and the sphere-only large scene: tools/coop_pmc.sh <tag>
export TMPDIR=/tmp RT_NO_TALLY=1
OUT=gpurun_out/coop_pmc_${1:-x}; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python3 tools/coop_probe.py > $OUT/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- python3 tools/coop_probe.py > $OUT/b.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "coop_kernel" not in r["Kernel_Name"]: continue
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        agg[k]["_ms"]+= (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 if r["Counter_Name"]=="GRBM_GUI_ACTIVE" else 0
        agg[k]["_vgpr"]=float(r["VGPR_Count"]); agg[k]["_scratch"]=float(r["Scratch_Size"]); agg[k]["_lds"]=float(r["LDS_Block_Size"])
    for k,v in agg.items():
        print(k, {c: "%.4g"%(x/len(n[k]) if not c.startswith("_v") and not c.startswith("_s") and not c.startswith("_l") else x) for c,x in sorted(v.items())})
PY
