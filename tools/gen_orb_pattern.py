#!/usr/bin/env python3
"""Regenerates include/pli_orb_pattern.inc (the 256x4 rBRIEF test table) from the
constant table at reference src/ORBextractor.cc:148-406.  Dev-time only: needs
/root/reference, which does not exist on the GPU box."""
import re, sys
src = open('/root/reference/src/ORBextractor.cc').read()
i = src.index('static int bit_pattern_31_[256*4]')
body = re.sub(r'/\*.*?\*/', '', src[i:src.index('};', i)], flags=re.S)
nums = [int(x) for x in re.findall(r'-?\d+', body[body.index('{'):])]
assert len(nums) == 1024
hdr = open('include/pli_orb_pattern.inc').read().split('\n')[:5]
rows = [','.join(str(n) for n in nums[k:k + 32]) + ',' for k in range(0, 1024, 32)]
open('include/pli_orb_pattern.inc', 'w').write('\n'.join(hdr + rows) + '\n')
