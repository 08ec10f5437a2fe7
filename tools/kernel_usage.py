#!/usr/bin/env python3
"""Register / LDS / scratch budget of every trace and generator kernel, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c vr_trace.hip 2> usage.txt; tools/kernel_usage.py usage.txt"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
for b in blocks:
    name = b.split()[0]

    def g(k):
        m = re.search(k + r': (\S+)', b)
        return m.group(1) if m else '?'
    n = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('vr::', '').split('(')[0]
    if len(sys.argv) > 2 and not any(k in n for k in sys.argv[2:]):
        continue
    print("%-40s sgpr %4s vgpr %4s scratch %4s occ %2s sspill %3s vspill %3s lds %6s" % (
        n[-40:], g('SGPRs'), g('VGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'),
        g('SGPRs Spill'), g('VGPRs Spill'), g(r'LDS Size \[bytes/block\]')))
