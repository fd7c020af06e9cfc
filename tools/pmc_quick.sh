#!/bin/bash
# GPU box: one rocprofv3 PMC pass of bench.py; usage: tools/pmc_quick.sh "<counters>" <kernel-regex> [bench args]
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
export PLI_USE_DEV_LIB=${PLI_USE_DEV_LIB-1}      # (environment switches are read by the development build of the library only)
grp=$1; pat=$2; shift; shift
rm -rf $R/gpurun_out/ppq
timeout 900 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/ppq -o s -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/ppq.log 2>&1
python3 $R/tools/rocprof_summary.py pmc $(find $R/gpurun_out/ppq -name "*.db" | head -1) | grep -E "$pat"
rm -rf $R/gpurun_out/ppq
