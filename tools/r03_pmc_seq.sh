#!/bin/bash
# counters of the listed kernel as frames in flight launch it (RT_FLAG_SEQUENCE), per setting of RT_WAVE_GROUPS_MS:
#   tools/r03_pmc_seq.sh <workload> <ms>...
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
wl=$1; shift
for ms in "$@"; do
  export RT_WAVE_GROUPS_MS=$ms
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_seq_${wl}_$ms
  rm -rf $out
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/pipeline_time.py $wl 4 2 > $out.log 2>&1 || { tail -n 5 $out.log; exit 1; }
  python3 - $out $ms <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if "render_kernel" not in r["Kernel_Name"]: continue
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[r["Kernel_Name"][:70]] += 1
for k, v in agg.items():
    c = n[k]
    print("RT_WAVE_GROUPS_MS=%s %s: %d launches; per launch VALU %.3e, lane utilisation %.1f %%, %.2f ms, waves %.0f" % (
        sys.argv[2], k, c, v["SQ_INSTS_VALU"] / c, 100 * v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]), v["GRBM_GUI_ACTIVE"] / 8 / c / 2.4e6, v["SQ_WAVES"] / c))
PY
done
