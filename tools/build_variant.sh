#!/bin/bash
# Build a variant of the library for on-box A/B runs: tools/build_variant.sh <tag> [extra hipcc flags, e.g. -DITA_X=1]
# -> csrc/libita_mi355x_<tag>.so (git-ignored, travels with gpurun).  tools/ab_multi.sh <tag> <tag> ... compares them.
set -e
TAG=$1; shift
cd "$(dirname "$0")/.."
D=drone-oa-iree-vit-accelerator_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function \
  -I include "$@" $D/ita_plugin.hip -o $D/libita_mi355x_$TAG.so
echo built $D/libita_mi355x_$TAG.so
