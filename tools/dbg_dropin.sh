#!/bin/bash
# debug helper: build the drop-in harness with symbols and print the backtrace of the first C++ throw
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import sys; sys.path.insert(0, "tests")
import numpy as np
from test_cpp_dropin import build_harness, write_input
from pli_slam_amd import synth
exe = build_harness("/tmp")
fr = [synth.make_stereo_pair(40 + s, 752, 480, t=t) for s in range(10) for t in range(5)]
write_input("/tmp/dbg.in", fr, 1, 0)
PY
/opt/rocm/bin/rocgdb -batch -ex "catch throw" -ex run -ex bt --args /tmp/dropin_harness /tmp/dbg.in /tmp/dbg.out 2>&1 | tail -40
