#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + two PMC passes of the wave-per-problem
# kernel on the config-4 workload (tools/pmc_dense.py, 32768 problems).  Raw outputs go to
# gpurun_out/<tag>/.
set -u
TAG=${1:-prof_dense}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RUN="python3 $R/tools/pmc_dense.py solve"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $RUN > $O/kt.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- $RUN > $O/pmc1.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH --output-format csv -d $O/pmc2 -- $RUN > $O/pmc2.log 2>&1 || exit 3
echo profile_done
