#!/bin/bash
# usage (on the GPU box): [PROBE="tools/pmc_dense.py solve" KERNEL=dense_kernel] tools/pmc_probe.sh TAG "COUNTERS..." ["COUNTERS2..."]
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/p$i -- python3 $R/${PROBE:-tools/pmc_probe.py} > $O/p$i.log 2>&1 || exit $i
done
python3 - <<PY
import csv, glob, collections
out = collections.defaultdict(list)
for f in glob.glob("$O/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "${KERNEL:-fused_iterate}" in r["Kernel_Name"]:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(out):
    print("$TAG", k, sum(out[k]) / len(out[k]))
PY
