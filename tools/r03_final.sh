#!/bin/bash
# Round 3, final tree: GPU suite, the round's profiles, the relaxed build's margins.  Then, here: tools/pmc_frame.py and
# tools/summarize_profile.py per tag, and bench.py once more (tools/r03_bench.sh) so that its line prices this tree.
set -o pipefail
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/r03f_tests.log 2>&1 || { tail -n 30 gpurun_out/r03f_tests.log; exit 1; }
tail -n 2 gpurun_out/r03f_tests.log
tools/r03_profiles.sh || exit 1
timeout -k 10 600 python -u tools/relaxed_margins.py gpurun_out/r03_relaxed_margins.json | tail -n 8
