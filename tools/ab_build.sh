#!/bin/bash
# Build a variant of libpcc_hip.so with extra compile flags on conv.hip only: tools/ab_build.sh <name> [flags...]
# -> tools/ab/<name>.so (git-ignored; travels to the GPU box).  The other objects come from the regular build.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/demo-learned-point-cloud-compression_amd/csrc
BD=$ROOT/demo-learned-point-cloud-compression_amd/build
name=$1; shift
mkdir -p "$ROOT/tools/ab" /tmp/ab_$name
make -s -j8 -C "$CS"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -c "$CS/conv.hip" -o /tmp/ab_$name/conv.o
objs=$(ls $BD/*.o | grep -v '/conv\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/$name.so" $objs /tmp/ab_$name/conv.o -lpthread
echo "built tools/ab/$name.so"
