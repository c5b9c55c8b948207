#!/bin/bash
# developer helper (GPU box): per-kernel durations of the Schur launches for a few unit counts of the one-wavefront run workgroups
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in "$@"; do
  w0=${cfg%%:*}; w1=${cfg##*:}
  rm -rf $R/gpurun_out/sw_$w0_$w1
  CMS_BA_RW_WAVES0=$w0 CMS_BA_RW_WAVES1=$w1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sw_${w0}_${w1} -o t -- python $R/tools/prof_ba_many.py 16 track same 3 > $R/gpurun_out/sw_${w0}_${w1}.log 2>&1
  echo "== waves/CU class0 $w0 class1 $w1: $(grep lock-step $R/gpurun_out/sw_${w0}_${w1}.log)"
  python - $R/gpurun_out/sw_${w0}_${w1}/t_kernel_trace.csv <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in ("kb_ba_lin_schur_run_wg0", "kb_ba_lin_schur_run_wg1", "kb_ba_lin_schur_edges", "kb_ba_lin_schur_runs", "kb_ba_trial_solve3r", "kb_ba_trial_edges"):
    v = sorted(d.get(k, []))
    if v:
        top = v[len(v) // 2:]          # the rounds in which every window was active
        print("   %-26s n %3d  median of upper half %.1f us  max %.1f us" % (k, len(v), top[len(top) // 2] / 1e3, v[-1] / 1e3))
PY
done
