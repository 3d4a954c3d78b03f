#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + separate PMC passes of the
# headline bench workload.  Raw outputs go to gpurun_out/<tag>/; tools/summarize_profile.py
# turns them into the committed summaries under profiles/.
set -u
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $BENCH > $O/kt.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_inst -- $BENCH > $O/pmc_inst.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $O/pmc_act -- $BENCH > $O/pmc_act.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH > $O/pmc_fetch.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH > $O/pmc_write.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_grbm -- $BENCH > $O/pmc_grbm.log 2>&1 || exit 6
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_all -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt_all.log 2>&1 || exit 7
echo profile_done
