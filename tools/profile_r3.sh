#!/bin/bash
# Round-3 profile set (GPU box): rocprofv3 kernel stats + PMC passes (separate passes, as the MI355X guide prescribes) of
#   main   : bench.py's 1024-frame step alone (--no-configs, so that every launch of a kernel name has the same batch)
#   c2     : tools/bench_mha_block.py      c5 : tools/bench_tail_large.py      c5mha : tools/bench_c5_mha.py 8 8192
# Keeps only the condensed files (raw traces exceed what gpurun merges back): gpurun_out/<tag>/<set>_{summary.txt,kernel_stats.csv}
# usage: tools/profile_r3.sh <tag> [sets...]
set -o pipefail
TAG=${1:-r03}; shift
SETS=${@:-main c2 c5 c5mha}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for S in $SETS; do
  case $S in
    main)  SCRIPT=bench.py; ARGS="--steps 24 --warmup 8 --no-cpu-baseline --no-latency --no-configs" ;;
    c2)    SCRIPT=tools/bench_mha_block.py; ARGS="" ;;
    c5)    SCRIPT=tools/bench_tail_large.py; ARGS="" ;;
    c5mha) SCRIPT=tools/bench_c5_mha.py; ARGS="8 8192" ;;
  esac
  rm -rf $R/gpurun_out/_raw_$S
  $R/tools/profile_cmd.sh _raw_$S $SCRIPT $ARGS
  cp $R/gpurun_out/_raw_$S/summary.txt $OUT/${S}_pmc_summary.txt
  find $R/gpurun_out/_raw_$S/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${S}_kernel_stats.csv
  [ $S = main ] && python3 $R/tools/make_kernel_traffic.py $R/gpurun_out/_raw_$S "tools/profile_r3.sh $TAG (bench.py --no-configs, 1024 frames)" > $OUT/kernel_traffic.json
  [ $S = main ] && grep '^{' $R/gpurun_out/_raw_$S/stats.log | tail -1 > $OUT/main_bench.json
  rm -rf $R/gpurun_out/_raw_$S
  echo "$S done"
done
