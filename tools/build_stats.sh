#!/bin/sh
# Dev build of the library with -DLSD_STATS into build/stats/ (git-ignored; travels with gpurun).  Use with PLI_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/stats
cp pli_slam_amd/csrc/*.hip pli_slam_amd/csrc/*.hpp pli_slam_amd/csrc/Makefile build/stats/
sed -i 's|\.\./\.\./include|../../include|g' build/stats/Makefile
make -C build/stats -j6 EXTRA=-DLSD_STATS >/dev/null
ls -la build/stats/libpli_frontend.so
