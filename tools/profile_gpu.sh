#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes over a
# short bench.py run.  Summaries land under gpurun_out/<tag>/ ; copy what is to be judged to profiles/.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-prof}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (no --no-configs: the same command also runs BASELINE config 2 and 5, so ita_stream_kernel<128, false, ...> and
#  ita_tail_big_kernel get their kernel stats and counters from this run too)
ARGS="--steps 24 --warmup 8 --no-cpu-baseline --no-latency $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py $ARGS > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $OUT/errors.log
done
python3 $R/tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
