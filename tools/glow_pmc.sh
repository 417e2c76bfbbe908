#!/bin/bash
# HBM counters of rt_apply_glow's kernels (tools/glow_time.py, 4K): -> gpurun_out/glow_pmc/, profiles/pmc_glow_4k.json
export TMPDIR=/tmp
OUT=gpurun_out/glow_pmc; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/glow_time.py > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/d -o d -- python3 tools/glow_time.py > $OUT/d.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/e -o e -- python3 tools/glow_time.py > $OUT/e.log 2>&1 || exit 1
python3 - <<PY
import csv, json, collections, sys
sys.path.insert(0, ".")
from bench import kernel_source_hash
tot = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float))
calls = 4.0                                   # glow_time.py applies the glow four times
for f in ("$OUT/d/d_counter_collection.csv", "$OUT/e/e_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "glow_" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]) / calls
        per[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"]) / calls
out = {"kernel_source_sha256": kernel_source_hash(), "fetch_kb": tot["FETCH_SIZE"], "write_kb": tot["WRITE_SIZE"], "hbm_bytes": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024,
       "per_kernel_kb": {k: dict(v) for k, v in per.items()}, "note": "per rt_apply_glow call on a 4K frame; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB"}
json.dump(out, open("profiles/pmc_glow_4k.json", "w"), indent=1)
print(json.dumps(out)[:600])
for r in list(csv.DictReader(open("$OUT/trace/trace_kernel_stats.csv")))[:8]:
    print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, "us")
PY
