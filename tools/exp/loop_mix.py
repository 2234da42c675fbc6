#!/usr/bin/env python3
"""Instruction mix of the largest loop of one kernel in a hipcc -S listing.
usage: tools/exp/loop_mix.py listing.s <mangled-name-prefix>"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
m = re.search(r'^(%s\w*):' % re.escape(sys.argv[2]), s, re.M)
st = m.end()
body = s[st:s.index('s_endpgm', st)].split('\n')
labels = {}
for k, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = k
loops = []
for k, l in enumerate(body):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
        loops.append((labels[mm.group(1)], k))
a, b = max(loops, key=lambda x: x[1] - x[0])
c = Counter()
for l in body[a:b]:
    l = l.strip()
    if not l or l.startswith(('.', ';')):
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'):
        c['mfma'] += 1
    elif op.startswith(('ds_', 's_load', 'global', 'buffer', 'scratch', 's_waitcnt', 's_barrier')):
        c[op] += 1
    elif op.startswith('v_'):
        c['valu'] += 1
    elif op.startswith('s_'):
        c['salu'] += 1
print(m.group(1), 'loop lines', a, b, dict(c))
