#!/bin/bash
# round 5, GPU call 5: one set-up kernel per window; chain streams at high priority; where cms_ba_create's in-step time goes
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "device_plan or put_from_frame or config4_size_eight or tracked_windows" 2>&1 | tail -3
for i in 1 2; do
  bash tools/gb.sh r05e_fast$i | cut -c1-200
  CMS_BA_HOST_PLAN=1 bash tools/gb.sh r05e_host$i | cut -c1-200
  CMS_BENCH_BA_PRIORITY=high bash tools/gb.sh r05e_fast_hi$i | cut -c1-200
done
CMS_BENCH_BA_PRIORITY=high CMS_BA_HOST_PLAN=1 bash tools/gb.sh r05e_host_hi | cut -c1-200
bash tools/gb.sh r05e_fast_wt8 --window-threads 8 | cut -c1-200
CMS_BENCH_BA_PRIORITY=high bash tools/gb.sh r05e_fast_hi_wt8 --window-threads 8 | cut -c1-200
CMS_BA_CREATE_TIMING=1 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-streaming-pass --verify-windows 0 --optimise-only-steps 0 --closed-loop-frames 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 2>&1 | grep "cms_ba_create\]" | tail -400 > $O/create_timing_fast.txt
python - $O/create_timing_fast.txt <<'PY'
import re, collections, sys
acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    for k, v in re.findall(r" ([a-z\-]+) ([0-9.]+)", l.split("ms:")[1]):
        acc[k].append(float(v))
print({k: round(sum(v) / len(v), 3) for k, v in acc.items()}, "windows", len(acc["pass"]), "sum", round(sum(sum(v) / len(v) for v in acc.values()), 2), "ms")
PY
CMS_BENCH_THREAD_CPU=1 bash tools/gb.sh r05e_fast_cpu | cut -c1-120; grep "window threads\|thread CPU" gpurun_out/gb_r05e_fast_cpu.log | cut -c1-300
