#!/bin/bash
# Development: waits, barriers, DMA and stores of one kernel of encoder.o in address order (is a compiler-inserted vmcnt(0) in the way?).
#   tools/kernel_flow.sh 'gemm8_kernelILi2ELb1EEEvNS0_9Gemm8ArgsE' [object]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
obj=${2:-$R/haconvdr_amd/csrc/encoder.o}
T=$(mktemp -d)
cp $obj $T/o.o && cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null
/opt/rocm/lib/llvm/bin/llvm-objdump -d o.o.*gfx950* > all.s
awk -v k="$1" '/^[0-9a-f]+ </{p = index($0, k) > 0} p' all.s > k.s
grep -n "s_waitcnt\|s_barrier\|global_load_lds\|global_store\|global_load_dword\|buffer_\|scratch_" k.s | sed 's#//.*##' | awk '{$1=$1};1' | cut -c1-100
rm -rf $T
