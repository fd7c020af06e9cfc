#!/bin/bash
# GPU box: rocprofv3 kernel stats of the C++ drop-in harness (10 frames x 3 repetitions, four threads)
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
R=$GRAFT_REPO_ROOT
cd $R && python3 - <<PY
import os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from test_cpp_dropin import build_harness, write_input
from pli_slam_amd import synth
exe = build_harness("/tmp")
frames = [synth.make_stereo_pair(40 + s, 752, 480, t=t) for s in range(2) for t in range(5)]
write_input("/tmp/dp.in", frames, 3, 1)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/dp
timeout 600 rocprofv3 --kernel-trace --stats --hip-trace -d $R/gpurun_out/dp -o s -- /tmp/dropin_harness /tmp/dp.in /tmp/dp.out > $R/gpurun_out/dp.log 2>&1
tail -2 $R/gpurun_out/dp.log
python3 $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/dp -name "*.db" | head -1) | head -30
python3 - <<PY
import sqlite3, glob
db = glob.glob("$R/gpurun_out/dp/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
for t in names:
    if 'hip' in t.lower() and 'top' in t.lower() or t == 'top':
        print(t)
try:
    for row in cur.execute("select name, total_calls, total_duration, average from top_hip_api order by total_duration desc limit 15"): print(row)
except Exception as e:
    print("no top_hip_api", e, [n for n in names if 'top' in n or 'api' in n][:20])
PY
rm -rf $R/gpurun_out/dp
