import sys, json
sys.path.insert(0, '/root/repo')
import os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import viennaray_amd as vr
t = vr.TraceDisk(3)
for kind in (2, 6, 2, 6):
    for w in (4, 8):
        r = t.debugIssueRate(kind, w, iters=20000)
        print(kind, w, 'cycles/step', 1024 * r['clock_hz'] / r['rate'], 'clock', r['clock_hz'] / 1e9)
