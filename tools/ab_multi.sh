#!/bin/bash
# A/B/C... builds of the library on the same GPU box (tools/build_variant.sh), alternated for ROUNDS rounds so that
# box-to-box and clock drift cancel: tools/ab_multi.sh tagA tagB ...   (tag "base" = the current libita_mi355x.so)
# Prints frames/s, ms/step and the dominant kernel's launch time per run.  BENCH_ARGS: extra bench.py arguments.
cd "$(dirname "$0")/.."
D=drone-oa-iree-vit-accelerator_amd/csrc
cp $D/libita_mi355x.so $D/libita_mi355x_base.so
run() { python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-latency $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
for i in $(seq 1 ${ROUNDS:-2}); do
  for t in "$@"; do
    cp $D/libita_mi355x_$t.so $D/libita_mi355x.so; run $t
  done
done
cp $D/libita_mi355x_base.so $D/libita_mi355x.so
