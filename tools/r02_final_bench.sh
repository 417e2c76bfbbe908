#!/bin/bash
# the round's bench lines: default (4K headline), glass in flight, C4 one at a time with the whole-frame tally, glow
set -o pipefail
timeout -k 10 600 python -u bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail gpurun_out/bench_default.err; exit 1; }
echo default done
timeout -k 10 600 python -u bench.py --workload glass_1080p_64spp --steps 32 > gpurun_out/bench_glass.json 2> gpurun_out/bench_glass.err || { tail gpurun_out/bench_glass.err; exit 1; }
echo glass done
timeout -k 10 900 python -u bench.py --workload many2048_1080p_32spp --steps 3 --warmup 1 --frames-in-flight 1 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err || { tail gpurun_out/bench_c4.err; exit 1; }
echo c4 done
timeout -k 10 300 python -u bench.py --workload glow_4k > gpurun_out/bench_glow.json 2> gpurun_out/bench_glow.err || { tail gpurun_out/bench_glow.err; exit 1; }
python3 - <<PY
import json
for n in ("default","glass","c4","glow"):
    d=json.load(open("gpurun_out/bench_%s.json"%n))
    print(n, d["ms_per_step"], d["value"], d.get("frame_latency_ms"), d.get("kernel_ms"), d["roofline"].get("frac"), d["roofline"].get("timed_region_valu_issue_frac"), d.get("verified",{}).get("identical"), (d.get("cpu_baseline") or {}).get("value"))
PY
