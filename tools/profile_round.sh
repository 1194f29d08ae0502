#!/bin/bash
# One profiling campaign of bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02
# Four rocprofv3 runs, the program directly after `--` (no wrapper between the profiler and python):
#   <tag>_trace  --kernel-trace --stats              the default bench (headline + extras), no CPU baseline
#   <tag>_fetch  --pmc FETCH_SIZE                     headline step only
#   <tag>_write  --pmc WRITE_SIZE                     headline step only
#   <tag>_sq     --pmc SQ_* GRBM_GUI_ACTIVE           headline step only: matrix-pipe busy cycles, wave-time split, clock
# Counters are collected in their own runs, with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section).
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
RAW=/tmp/hac_prof                      # raw rocprofv3 output stays on the box (it exceeds what gpurun copies back)
OUT=$R/gpurun_out/${TAG}_profiles      # summaries + bench lines: copied back, then `cp` into profiles/
rm -rf $RAW && mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/${TAG}_trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err || { echo "trace run failed"; tail -5 $OUT/${TAG}_trace.err; exit 1; }
echo "trace done"
for pass in fetch write sq; do
  case $pass in
    fetch) PMC="FETCH_SIZE";;
    write) PMC="WRITE_SIZE";;
    sq) PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE";;
  esac
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $RAW/${TAG}_$pass -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_${pass}_bench.json 2> $OUT/${TAG}_$pass.err || { echo "$pass run failed"; tail -5 $OUT/${TAG}_$pass.err; exit 1; }
  echo "$pass done"
done
python3 $R/tools/refresh_profiles.py $TAG --raw $RAW --out $OUT > $OUT/${TAG}_refresh.log 2>&1 || { echo "summary failed"; tail -5 $OUT/${TAG}_refresh.log; exit 1; }
du -sh $RAW $OUT
