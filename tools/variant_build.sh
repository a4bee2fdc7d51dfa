#!/bin/bash
# tools/variant_build.sh <name> [file.hip="extra hipcc flags" ...]: scratch library tools/scratch/libplf_<name>.so = the in-tree sources with the
# named files compiled with extra flags (e.g. lsd_kernels.hip="-DPLF_REGIONS_WPE=8").  A/B on one GPU box:
#   PLF_LIB_PATH=tools/scratch/libplf_<name>.so python bench.py ...
set -e
name=$1; shift
cd "$(dirname "$0")/../rgbd_pl_slam_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fgpu-rdc -w"
O=/tmp/plf_variant_$name; C=/tmp/plf_variant_objs; mkdir -p $O $C ../../tools/scratch
for f in *.hip; do
  b=${f%.hip}; extra=""
  for kv in "$@"; do [ "${kv%%=*}" == "$f" ] && extra="${kv#*=}"; done
  if [ -n "$extra" ]; then /opt/rocm/bin/hipcc $FLAGS $extra -c $f -o $O/$b.o &
  elif [ -f $C/$b.o ] && [ $C/$b.o -nt $f ] && [ -z "$(find . -name '*.h' -newer $C/$b.o)" ]; then cp $C/$b.o $O/$b.o
  else ( /opt/rocm/bin/hipcc $FLAGS -c $f -o $C/$b.o && cp $C/$b.o $O/$b.o ) & fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fgpu-rdc --hip-link -shared -fPIC -o ../../tools/scratch/libplf_$name.so $O/*.o
echo built tools/scratch/libplf_$name.so
