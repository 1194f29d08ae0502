#!/bin/bash
# Development (GPU box): PMC counters of scanh_kernel<1,false,16> over tools/ab_search.py, one rocprofv3 --pmc pass per counter set.
#   bash tools/pmc_scan.sh <variant name under scratch/v | in-tree> "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQC_ICACHE_REQ SQC_ICACHE_MISSES" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
v=$1; shift
[ "$v" = in-tree ] || export HAC_LIBRARY_PATH=$R/scratch/v/libhaconvdr_$v.so
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcscan_${v}_$i -- python3 $R/tools/ab_search.py ${ROWS:-10000000} 2 > $R/gpurun_out/pmcscan_${v}_$i.log 2>&1 || tail -5 $R/gpurun_out/pmcscan_${v}_$i.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmcscan_${v}_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "scanh_kernel<1, false, 16>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, sum(v) / len(v), "n", len(v))
PY
