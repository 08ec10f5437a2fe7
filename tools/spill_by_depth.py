#!/usr/bin/env python3
"""Where a kernel's spill traffic sits: counts v_readlane / v_writelane (scalar spills), scratch_ loads / stores
(vector spills), s_load (kernel-argument reloads) and all instructions per loop depth of its ISA.
usage: hipcc -S --cuda-device-only ... -o k.s; tools/spill_by_depth.py k.s '<mangled kernel name>'"""
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
on = False
cur = 0
cnt = {}
for l in txt:
    if l.startswith(name + ':'):
        on = True
    if not on:
        continue
    if 's_endpgm' in l:
        break
    if l.startswith('.LBB') or l.startswith('; %bb'):
        m = re.search(r'Depth=(\d+)', l)
        cur = int(m.group(1)) if m else 0
    ins = l.strip().split(' ')[0]
    if not ins or ins.startswith((';', '.')) or ins.endswith(':'):
        continue
    key = 'other'
    for k in ('v_readlane', 'v_writelane', 'scratch_load', 'scratch_store', 's_load'):
        if ins.startswith(k):
            key = k
    cnt.setdefault(cur, {}).setdefault(key, 0)
    cnt[cur][key] += 1
for d in sorted(cnt):
    tot = sum(cnt[d].values())
    print(f"depth {d}: {tot:5d} instructions  " + "  ".join(f"{k} {v}" for k, v in sorted(cnt[d].items()) if k != 'other'))
