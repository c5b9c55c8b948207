#!/bin/bash
# left-over points composed with a look-ahead of 8 (before) against the caller's order (now, when they are a minority): host CPU, step, kernel
set -u
for v in "CMS_BA_LEFTOVER_LOOKAHEAD=8" "CMS_BA_LEFTOVER_LOOKAHEAD=0"; do
for i in 1 2; do
echo "== $v"
env $v CMS_BENCH_STEP_TIMES=1 CMS_BENCH_THREAD_CPU=1 python bench.py --steps 150 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 2>&1 | python -c "
import sys, re, json, statistics
txt = sys.stdin.read()
t = [float(x) for x in re.search(r'step times \(ms\): \[(.*?)\]', txt).group(1).split(',')]
print('steps: mean %.2f median %.2f max %.2f, >16 ms: %d of %d' % (statistics.mean(t), statistics.median(t), max(t), sum(1 for x in t if x > 16), len(t)))
for l in txt.splitlines():
    if l.startswith('window threads') or l.startswith('thread CPU'): print(l[:120])
    if l.startswith('{'):
        d = json.loads(l); print('schur us', 1e3 * d['roofline']['ms_per_launch'])
"
done
env $v python tools/prof_ba_many.py 16 track diff 2>&1 | grep "lock-step\|cms_ba_create" | tail -2
done
