#!/bin/bash
# Build container (dev tool): VGPR / SGPR / scratch use of the kernels of one source file whose name matches a pattern.
#   tools/kernel_regs.sh lsd_tile.hip k_tx_grow [extra hipcc flags]
set -e -o pipefail
cd "$(dirname "$0")/.."
src=$1; pat=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -S -o /tmp/kregs.s pli_slam_amd/csrc/$src "$@" 2>/dev/null
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|name):" /tmp/kregs.s | paste - - - - - | grep -E "$pat" \
  | sed -E 's/\.name: _ZN3pli[0-9]+([A-Za-z0-9_]+)E[^\t]*/\1/; s/  */ /g'
