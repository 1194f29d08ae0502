#!/bin/bash
# Development: disassemble one kernel of encoder.o (or another object) into scratch/dis/<tag>.s and print instruction counts.
#   tools/kdis.sh 'attention_pipe_kernelILi8E' k8 [object]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
obj=${3:-$R/haconvdr_amd/csrc/encoder.o}
D=$R/scratch/dis
mkdir -p $D/tmp_$2
cp $obj $D/tmp_$2/o.o
(cd $D/tmp_$2 && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null && /opt/rocm/lib/llvm/bin/llvm-objdump -d o.o.*gfx950* > all.s)
awk -v k="$1" '/^[0-9a-f]+ </{p = index($0, k) > 0} p' $D/tmp_$2/all.s | sed 's#//.*##' > $D/$2.s
rm -r $D/tmp_$2
echo "$D/$2.s: $(wc -l < $D/$2.s) lines"
for pat in v_mfma scratch_ v_writelane v_readlane v_accvgpr s_waitcnt s_barrier global_load_lds ds_read v_exp_f32 s_nop s_cbranch; do
    printf "  %-18s %s\n" $pat $(grep -c "$pat" $D/$2.s || true)
done
