#!/bin/bash
# rank 0's share of an N-GPU frame on one GPU, for several builds: tools/emulate_ranks.sh "<ranks...>" "<lib suffixes...>"
for n in $1; do for v in $2; do lib=raytracer_amd/librt_amd_$v.so; [ $v = main ] && lib=raytracer_amd/librt_amd.so; echo "ranks $n lib $v"; RT_EMULATE_RANKS=$n RT_AMD_LIB=$PWD/$lib timeout -k 10 100 python tools/time_workload.py default4_4k_64spp 0 2>/dev/null | tail -2; done; done
