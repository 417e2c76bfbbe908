#!/bin/bash
# the driver's bench command on the final tree (after profiles/pmc_*.json were regenerated for it), and the C4 / glow lines
set -o pipefail
timeout -k 10 600 python -u bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || { tail gpurun_out/r03_bench_default.err; exit 1; }
tail -n 2 gpurun_out/r03_bench_default.err
timeout -k 10 600 python -u bench.py --workload many2048_1080p_32spp --steps 3 --warmup 1 --no-extras --estimate-tally --no-verify --no-cpu-baseline > gpurun_out/r03_bench_c4.json 2> gpurun_out/r03_bench_c4.err || { tail gpurun_out/r03_bench_c4.err; exit 1; }
timeout -k 10 600 python -u bench.py --workload glass_1080p_64spp --steps 32 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03_bench_glass.json 2> gpurun_out/r03_bench_glass.err || { tail gpurun_out/r03_bench_glass.err; exit 1; }
timeout -k 10 300 python -u bench.py --workload glow_4k > gpurun_out/r03_bench_glow_4k.json 2> gpurun_out/r03_bench_glow_4k.err || { tail gpurun_out/r03_bench_glow_4k.err; exit 1; }
python3 - <<PY
import json
for n in ("default", "c4", "glass", "glow_4k"):
    d = json.load(open("gpurun_out/r03_bench_%s.json" % n))
    r = d.get("roofline") or {}
    print(n, d["ms_per_step"], d["value"], d["unit"], "latency", d.get("frame_latency_ms"), "kernel_ms", d.get("kernel_ms"),
          {k: r.get(k) for k in ("frac", "pmc_stale", "executed_frac", "traffic", "lane_utilisation")}, (d.get("verified") or {}).get("identical"))
PY
