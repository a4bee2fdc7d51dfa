#!/bin/bash
# rebuild lsd_kernels with -DPLF_LSD_TIMING into a scratch library and print the per-section cycle counts (frame 0)
set -e
cd "$(dirname "$0")/../rgbd_pl_slam_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fgpu-rdc -w"
mkdir -p /tmp/plft && for f in orb_kernels orb_front orb_octree orb_host line_kernels line_host match_kernels match_host frame_kernels batch_host api_misc; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/plft/$f.o & done; /opt/rocm/bin/hipcc $FLAGS -DPLF_LSD_TIMING $PLF_TIMING_EXTRA -c lsd_kernels.hip -o /tmp/plft/lsd_kernels.o; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fgpu-rdc --hip-link -shared -fPIC -o /tmp/plft/libplf_hip.so /tmp/plft/*.o
