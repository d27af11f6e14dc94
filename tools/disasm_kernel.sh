#!/bin/bash
# disassemble the gfx950 code object of one built object file:  tools/disasm_kernel.sh oryon_amd/csrc/pdsc_encoder.o > /tmp/x.s
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section=.hip_fatbin=$T/fat.bin "$1" $T/host.o
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
$LLVM/llvm-objdump -d $T/dev.co
rm -rf $T
