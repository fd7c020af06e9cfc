#!/bin/sh
# Dev build of the library with extra compiler flags into a git-ignored directory two levels below the repository root (the sources
# include "../../include/..."; default build/variant; travels with gpurun), for A/B timing on one box:
#   tools/build_variant.sh -DTX_DIAG_PAD=2 ;  PLI_LIB_PATH=build/variant/libpli_frontend.so python bench.py ...
#   VARIANT_DIR=build/padv2 tools/build_variant.sh -DTX_DIAG_PAD=2 ;  tools/ab_libs.sh base build/padv2        (several variants side by side)
set -e
cd "$(dirname "$0")/.."
D=${VARIANT_DIR:-build/variant}
mkdir -p "$D"
cp pli_slam_amd/csrc/*.hip pli_slam_amd/csrc/*.hpp pli_slam_amd/csrc/Makefile "$D/"
# (diagnostic macros are development-build material: the variant is the development build; "product" as first argument builds that one)
if [ "$1" = product ]; then shift; make -C "$D" -j6 product EXTRA="$*" >/dev/null; ls -la "$D/libpli_frontend.so"
else make -C "$D" -j6 dev EXTRA="$*" >/dev/null; ls -la "$D/libpli_frontend_dev.so"; fi
