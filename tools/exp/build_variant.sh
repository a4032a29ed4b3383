#!/bin/bash
# usage: tools/exp/build_variant.sh NAME [extra hipcc flags, e.g. -DRPO_SOMETHING=2]: a build of the in-tree library sources with the
# extra flags (ISA_CHECK=0: a variant's code is not the validated one) -> tools/exp/librankpo_hip_NAME.so (same-process A/B arms for tools/*_ab.py; binaries are not tracked)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
tmp=$(mktemp -d)
cp rankpo_amd/csrc/*.hip rankpo_amd/csrc/*.hpp rankpo_amd/csrc/*.inc rankpo_amd/csrc/Makefile "$tmp"/
python3 tools/gen/gen_fwd128w_body.py > "$tmp/attention_fwd128w_gen_variant.inc"     # (-DRPO_FW_VARIANT_INC + GEN_* in the environment)
sed -i "s|../../include/rankpo_hip.h|$(pwd)/include/rankpo_hip.h|" "$tmp/Makefile"
sed -i "s|#include \"../../include/rankpo_hip.h\"|#include \"$(pwd)/include/rankpo_hip.h\"|" "$tmp/common.hpp"
make -C "$tmp" -j8 ISA_CHECK=0 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -I$(pwd)/include $*" > /dev/null
cp "$tmp/librankpo_hip.so" "tools/exp/librankpo_hip_$name.so"
rm -rf "$tmp"
echo "built tools/exp/librankpo_hip_$name.so"
