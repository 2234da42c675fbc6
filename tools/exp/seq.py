#!/usr/bin/env python3
"""Compact instruction sequence of one kernel in a hipcc -S listing, basic block by basic block.
usage: tools/exp/seq.py listing.s <mangled-name-prefix> [first_line last_line]
M matrix, d ds_read, w ds_write, G LDS-DMA, L global load, S global store, . VALU, , SALU, <..> waitcnt, ^ branch"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(%s\w*):' % re.escape(sys.argv[2]), s, re.M)
body = s[m.end():s.index('s_endpgm', m.end())].split('\n')
a = int(sys.argv[3]) if len(sys.argv) > 3 else 0
b = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
out = []
for l in body[a:b]:
    l = l.strip()
    if not l or l[0] == ';':
        continue
    if l.startswith('.LBB'):
        out.append('\n' + l.split(':')[0] + ': ')
        continue
    if l[0] == '.':
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'):
        out.append('M')
    elif op.startswith('ds_read'):
        out.append('d')
    elif op.startswith('ds_'):
        out.append('w')
    elif op.startswith('global_load_lds'):
        out.append('G')
    elif op.startswith(('global_load', 'scratch_load', 'buffer_load')):
        out.append('L')
    elif op.startswith(('global_store', 'scratch_store', 'global_atomic')):
        out.append('S')
    elif op.startswith('s_waitcnt'):
        out.append('<' + l.split(None, 1)[1].replace(' ', '') + '>')
    elif op.startswith('s_barrier'):
        out.append('|BAR|')
    elif op.startswith(('s_cbranch', 's_branch')):
        out.append('^(' + l.split()[-1] + ')')
    elif op.startswith('v_'):
        out.append('.')
    elif op.startswith('s_'):
        out.append(',')
print(''.join(out))
