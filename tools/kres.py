#!/usr/bin/env python3
"""Per-kernel resource usage from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr saved to a file).
usage: tools/kres.py remarks.txt [name-substring ...]"""
import re
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)
for b in blocks[1:]:
    name = b.split('\n')[0]
    if pats and not any(p in name for p in pats):
        continue

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    print(name[:70].ljust(70), 'VGPR', g('VGPRs'), 'AGPR', g('AGPRs'), 'SGPR', g('SGPRs'), 'scratch', g(r'ScratchSize \[bytes/lane\]'),
          'occ', g(r'Occupancy \[waves/SIMD\]'), 'LDS', g(r'LDS Size \[bytes/block\]'))
