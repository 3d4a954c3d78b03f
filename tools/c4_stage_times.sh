#!/bin/bash
# On the GPU box: kernel trace of tools/time_config4.py (benchmark config 4, one solve_ik batch of
# 262144 problems) for one or more builds of the library, mean duration of each stage's kernel.
#   bash tools/c4_stage_times.sh main [variant ...]     (variant = tools/tmp/libmink_<variant>.so,
#   a copy of build_ab/libmink_<variant>.so from tools/build_variant.sh; build_ab/ does not travel)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
for v in "$@"; do
  if [ "$v" != "main" ]; then export MINK_AMD_LIB=$R/tools/tmp/libmink_$v.so; else unset MINK_AMD_LIB; fi
  rm -rf $R/gpurun_out/c4kt_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c4kt_$v -- python3 $R/tools/time_config4.py > $R/gpurun_out/c4kt_$v.log 2>&1 || exit 1
  echo "== $v"; grep "config 4" $R/gpurun_out/c4kt_$v.log
  python3 - "$R/gpurun_out/c4kt_$v" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_stats.csv', recursive=True))[-1]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'fused_' in n or 'dense' in n:
        print('  ', n[:70], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
done
