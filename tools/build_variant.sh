#!/bin/sh
# Dev build of the library with extra compiler flags into build/variant/ (git-ignored; travels with gpurun), for A/B timing on one
# box:   tools/build_variant.sh -DLSD_STATS ;  PLI_LIB_PATH=build/variant/libpli_frontend.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variant
D=${VARIANT_DIR:-build/variant}; mkdir -p $D; cp pli_slam_amd/csrc/*.hip pli_slam_amd/csrc/*.hpp pli_slam_amd/csrc/Makefile $D/
sed -i "s|\.\./\.\./include|$(pwd)/include|g" $D/Makefile; make -C $D -j6 EXTRA="$*" >/dev/null
ls -la $D/libpli_frontend.so
