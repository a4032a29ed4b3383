#!/bin/bash
# usage: tools/exp/build_d128_variant.sh NAME [GEN_* env assignments...]: a library build whose head_dim-128 dK/dV slice body is
# generated with the given experiment switches (timing only: some switches give wrong results) -> tools/exp/librankpo_hip_NAME.so
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
tmp=$(mktemp -d)
cp rankpo_amd/csrc/*.hip rankpo_amd/csrc/*.hpp rankpo_amd/csrc/Makefile "$tmp"/
mkdir -p "$tmp/../include_dummy"
env "$@" python tools/gen/gen_dkdv128_body.py > "$tmp/attention_dkdv128_gen.inc"
sed -i "s|-I../../include|-I$(pwd)/include|; s|../../include/rankpo_hip.h|$(pwd)/include/rankpo_hip.h|" "$tmp/Makefile"
sed -i "s|#include \"../../include/rankpo_hip.h\"|#include \"$(pwd)/include/rankpo_hip.h\"|" "$tmp/common.hpp"
make -C "$tmp" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I$(pwd)/include $CXXEXTRA" > /dev/null
cp "$tmp/librankpo_hip.so" "tools/exp/librankpo_hip_$name.so"
rm -rf "$tmp"
echo "built tools/exp/librankpo_hip_$name.so"
