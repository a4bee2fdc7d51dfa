#!/bin/bash
# VGPR / SGPR / spill / LDS usage of every k_* kernel in a built library (default: the in-tree one)
so=${1:-$(dirname "$0")/../rgbd_pl_slam_amd/libplf_hip.so}
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=/tmp/kr_fb $so
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/kr_fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/kr_dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kr_dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('  - .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', blk) or [None,'?'])[1]
    name=g('name')
    m=re.match(r'_Z\d+(k_[a-z0-9_]+)', name)
    if m: print('%-28s vgpr %3s spill %3s sgpr %3s sgpr_spill %3s lds %6s scratch %4s' % (m.group(1), g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('sgpr_spill_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
" | sort
