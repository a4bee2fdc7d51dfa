#!/bin/bash
# tools/disasm_kernel.sh <library.so> <kernel name substring> [out.s]: gfx950 machine code of one kernel of a built library (llvm-objdump of the code object)
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fb "$1"
$L/clang-offload-bundler --unbundle --type=o --input=$T/fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$L/llvm-objdump -d --no-show-raw-insn $T/dev.co | awk -v k="$2" '/^[0-9a-f]+ <.*>:/ { on = index($0, k) > 0 } on' > "${3:-/dev/stdout}"
rm -rf $T
