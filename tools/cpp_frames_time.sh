#!/bin/bash
# times the C++ driver's RenderFrames on the glass scene: tools/cpp_frames_time.sh
set -e
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/cppseq
python3 -c "
import raytracer_amd as ra
ra.scenes.glass_mirror().save('gpurun_out/cppseq/glass.rtscene')"
for f in 1 6; do
  timeout -k 10 200 raytracer_amd/host/rt_render --scene gpurun_out/cppseq/glass.rtscene --w 1920 --h 1080 --samples 64 --depth 12 --frames 12 --in-flight $f --out gpurun_out/cppseq/g | tail -1
done
for f in 1 6; do
  timeout -k 10 200 raytracer_amd/host/rt_render --w 3840 --h 2160 --samples 64 --depth 10 --frames 24 --in-flight $f --out gpurun_out/cppseq/d | tail -1
done
rm -f gpurun_out/cppseq/*.ppm
