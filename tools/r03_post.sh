#!/bin/bash
# here, after tools/r03_final.sh on the GPU box: per-frame counter totals (with the kernel-source hash) and the summaries
set -e
cd "$(dirname "$0")/.."
python3 tools/pmc_frame.py r03_default default4_4k_64spp > /dev/null
python3 tools/pmc_frame.py r03_relaxed default4_4k_64spp_relaxed > /dev/null
python3 tools/pmc_frame.py r03_c5 glass_1080p_64spp > /dev/null
python3 tools/pmc_frame.py r03_c4 many2048_1080p_32spp > /dev/null
python3 tools/summarize_profile.py r03_default "Round 3, final tree: DEFAULT-4, 3840x2160 / 64 spp / depth 10 on one MI355X, frames one at a time (the headline workload of bench.py)." > /dev/null
RT_PROFILE_ARGS="--frames-in-flight 1 --no-verify --flags 128" python3 tools/summarize_profile.py r03_relaxed "Round 3, final tree: the same frame with RT_FLAG_RELAXED (the direct kernel of rt_kernels_relaxed.hip: fused multiply-adds, 1-ulp sqrt / reciprocal)." > /dev/null
RT_PROFILE_ARGS="--frames-in-flight 1 --no-verify --workload glass_1080p_64spp" python3 tools/summarize_profile.py r03_c5 "Round 3, final tree: C5 (glass / mirror / mixed materials, 1920x1080 / 64 spp / depth 12), frames one at a time." > /dev/null
RT_PROFILE_ARGS="--frames-in-flight 1 --no-verify --estimate-tally --workload many2048_1080p_32spp" python3 tools/summarize_profile.py r03_c4 "Round 3, final tree: C4 (2049 hittables, 8 point lights, 1920x1080 / 32 spp / depth 10), frames one at a time." > /dev/null
cp gpurun_out/r03_relaxed_margins.json profiles/ 2>/dev/null || true
git status --short profiles | head -20
