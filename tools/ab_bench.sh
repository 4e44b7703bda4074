#!/bin/bash
# A/B two builds of the library on the same GPU box: csrc/libita_mi355x.so (A) against csrc/libita_mi355x_b.so (B),
# alternating A B A B so that box-to-box and clock drift cancel.  Prints ms/step and the encoder launch time.
set -e
EXTRA="$*"   # extra bench.py arguments, e.g. --image-dtype u8
cd "$(dirname "$0")/.."
D=drone-oa-iree-vit-accelerator_amd/csrc
cp $D/libita_mi355x.so $D/libita_mi355x_a.so
run() { python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-latency $EXTRA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
for i in 1 2 3; do
  cp $D/libita_mi355x_a.so $D/libita_mi355x.so; run A
  cp $D/libita_mi355x_b.so $D/libita_mi355x.so; run B
done
cp $D/libita_mi355x_a.so $D/libita_mi355x.so
