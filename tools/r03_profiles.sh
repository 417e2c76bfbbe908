#!/bin/bash
# Round 3's profiles (on the GPU box): kernel trace + PMC passes of bench.py for the headline, its relaxed variant, C5 and C4.
# Afterwards, here: tools/pmc_frame.py <tag> <workload> and tools/summarize_profile.py <tag> <note> per tag.
set -o pipefail
tools/profile_gpu.sh r03_default --frames-in-flight 1 --no-verify || exit 1
tools/profile_gpu.sh r03_relaxed --frames-in-flight 1 --no-verify --flags 128 || exit 1
tools/profile_gpu.sh r03_c5 --frames-in-flight 1 --no-verify --workload glass_1080p_64spp || exit 1
tools/profile_gpu.sh r03_c4 --frames-in-flight 1 --no-verify --estimate-tally --workload many2048_1080p_32spp || exit 1
cat gpurun_out/prof_r03_*/status.txt
