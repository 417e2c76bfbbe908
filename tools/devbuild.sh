#!/bin/bash
# quick experiment build of the C-ABI library: tools/devbuild.sh <suffix> [extra -D flags...]  -> raytracer_amd/librt_amd_<suffix>.so
# (only the kernel variants the bench workloads use: RT_DEV_FAST)
set -e
S=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
B=/tmp/devbuild_$S; mkdir -p $B
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DRT_SCALAR_SCENE -DRT_DEV_FAST --offload-arch=gfx950 -I../../include -I. $*"
/opt/rocm/bin/hipcc $F -c rt_kernels.hip -o $B/rt_kernels.o &
/opt/rocm/bin/hipcc $F -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -c rt_kernels_relaxed.hip -o $B/rt_kernels_relaxed.o &
/opt/rocm/bin/hipcc $F -x hip -c rt_abi.cpp -o $B/rt_abi.o &
/opt/rocm/bin/hipcc $F -c rt_glow.hip -o $B/rt_glow.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/rt_kernels.o $B/rt_kernels_relaxed.o $B/rt_glow.o $B/rt_abi.o -o ../librt_amd_$S.so
ls -la ../librt_amd_$S.so
