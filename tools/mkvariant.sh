#!/bin/bash
# Development: build a variant of the search kernels beside the in-tree library.
#   tools/mkvariant.sh NAME [-DSH_...=...]   ->  scratch/v/libhaconvdr_NAME.so   (scratch/ is not tracked)
# Run it with  HAC_LIBRARY_PATH=$PWD/scratch/v/libhaconvdr_NAME.so python tools/ab_search.py ...
# Switches of scan_split.inc: SH_M16, SH_QFAST, SH_LATE, SH_STAGE, SH_REC_CAP=n (small staging region: overflow route),
# timing-only ablations SH_NOEPI=1|2|3 and SH_NODMA=1|3, stamps SH_STAMP (+ SH_ST_ALL | SH_ST_EPI, SH_ST_R=round, SH_ST_T=step).
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/haconvdr_amd/csrc
V=$R/scratch/v
mkdir -p $V
[ -f $C/encoder.o ] || make -C $C encoder.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" -I$C -c $C/flat_ip.hip -o $V/f_$name.o 2> $V/f_$name.err || { grep -E "error" $V/f_$name.err | head; exit 1; }
hipcc -shared -fPIC --offload-arch=gfx950 -o $V/libhaconvdr_$name.so $V/f_$name.o $C/encoder.o
grep -A12 "Function Name: .*scanh_kernelILi1ELb0" $V/f_$name.err | grep -E "VGPRs:|ScratchSize" | sed 's/\[-Rpass.*//' | awk '{$1="";print}' | tr '\n' ' '; echo " -> $name"
