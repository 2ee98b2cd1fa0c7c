#!/usr/bin/env python3
"""Instruction-class stream of one kernel in a hipcc -S listing (one character per instruction, a line per s_barrier).
M mfma, G global_load_lds, g other global, r ds_read, w ds_write, T transcendental, v VALU, s SALU, b branch,
W s_waitcnt, | label.  Usage: isa_stream.py file.s mangled_name [--full]"""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^' + re.escape(name) + r':', s, re.M)
e = s.find('.Lfunc_end', m.end())
body = s[m.end():e]
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';')
         and not (l.strip().startswith('.') and not l.strip().startswith('.LBB'))]
def cls(l):
    op = l.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('global_load_lds'): return 'G'
    if op.startswith(('global_', 'buffer_')): return 'g'
    if op.startswith('ds_read'): return 'r'
    if op.startswith('ds_'): return 'w'
    if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')): return 'T'
    if op.startswith('v_'): return 'v'
    if op.startswith('s_waitcnt'): return 'W'
    if op.startswith('s_barrier'): return 'B\n'
    if op.startswith(('s_cbranch', 's_branch')): return 'b'
    if op.startswith('s_'): return 's'
    if op.startswith('.LBB'): return '|'
    return '?'
print(len(lines), 'instructions')
if '--full' in sys.argv:
    print('\n'.join(lines))
else:
    print(''.join(cls(l) for l in lines))
