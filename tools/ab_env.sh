#!/bin/bash
# A/B on one GPU box of library builds AND environment switches, alternated for ROUNDS rounds:
#   tools/ab_env.sh "tag[:VAR=value[,VAR=value...]]" ...      tag "base" = the current libita_mi355x.so, others = libita_mi355x_<tag>.so
# e.g. tools/ab_env.sh base base:ITA_FAST_SITES=0 r2        Prints frames/s, ms/step, encoder launch ms per run.
cd "$(dirname "$0")/.."
D=drone-oa-iree-vit-accelerator_amd/csrc
cp $D/libita_mi355x.so $D/libita_mi355x_base.so
for i in $(seq 1 ${ROUNDS:-2}); do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=$(echo "${spec#*:}" | tr ',' ' ')
    cp $D/libita_mi355x_$tag.so $D/libita_mi355x.so
    env $envs python bench.py --steps ${STEPS:-300} --warmup 50 --no-cpu-baseline --no-latency --no-configs $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$spec', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
cp $D/libita_mi355x_base.so $D/libita_mi355x.so
