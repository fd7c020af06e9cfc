#!/usr/bin/env python3
"""Regenerates pli_slam_amd/csrc/kernels.hpp (kernel prototypes) from the .hip sources."""
import os, re
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pli_slam_amd', 'csrc')
hdr = open(os.path.join(d, 'kernels.hpp')).read()
head = hdr[:hdr.index('__global__')]
out = [head.rstrip('\n')]
for f in ['orb_kernels.hip', 'line_kernels.hip', 'lsd_f64.hip', 'lsd_relax.hip', 'lsd_tile.hip', 'match_kernels.hip']:
    s = open(os.path.join(d, f)).read()
    for m in re.finditer(r'__global__[^{;]*?void\s+(k_\w+)\s*\(([^{]*?)\)\s*\{', s, flags=re.S):
        out.append('__global__ void %s(%s);' % (m.group(1), ' '.join(m.group(2).split())))
out.append('\n}  // namespace pli\n')
open(os.path.join(d, 'kernels.hpp'), 'w').write('\n'.join(out))
