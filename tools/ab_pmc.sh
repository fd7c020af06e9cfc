#!/bin/bash
# GPU box (dev tool): one rocprofv3 counter pass of the headline batch per variant build; args: "<counters>" <kernel regex> <build dirs | base>...
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
R=$GRAFT_REPO_ROOT; grp=$1; pat=$2; shift; shift
export PLI_USE_DEV_LIB=1      # (environment switches are read by the development build of the library only)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != base ]; then export PLI_LIB_PATH=$(ls $R/$v/libpli_frontend_dev.so 2>/dev/null || echo $R/$v/libpli_frontend.so); else unset PLI_LIB_PATH; fi
  rm -rf $R/gpurun_out/ppq
  PLI_SIDE_MAX=0 timeout 900 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/ppq -o s -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-leg --no-large-batch-leg > $R/gpurun_out/ppq.log 2>&1
  echo "== $v: $grp"
  python3 $R/tools/rocprof_summary.py pmc $(find $R/gpurun_out/ppq -name "*.db" | head -1) | grep -E "$pat"
  rm -rf $R/gpurun_out/ppq
done
