#!/usr/bin/env python3
"""Condense gpurun_out/prof/<tag>_* (rocprofv3 csv output) into the small, tracked summaries under profiles/."""
import collections, csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
PB = int(os.environ.get("PROF_B", "256"))      # frames per dispatch of the PMC passes (tools/run_profiles.sh)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof"); dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
stats = os.path.join(src, tag + "_bench_kernel_stats_timed_steps.csv")      # cut to the last six timed steps by tools/step_table.py (rocprofv3's own --stats
if not os.path.exists(stats):                                                # file covers the whole process: set-up launches, warm-up and pool priming included)
    stats = os.path.join(src, tag + "_bench_kernel_stats.csv")
if os.path.exists(stats):
    rows = list(csv.DictReader(open(stats)))
    cut = stats.endswith("_timed_steps.csv")
    with open(os.path.join(dst, tag + "_bench_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace -- python bench.py --steps 10 --warmup 2 --cpu-frames 0 --closed-loop-frames 0 --no-streaming-pass "
                "--optimise-only-steps 0 --verify-windows 0 --extract-only-steps 0 --random-views-steps 0 --mapping-only-steps 0 --unpipelined-steps 0 --deterministic-steps 0   (MI355X, 1 GPU; every step "
                "inserts its 32 key frames, runs CreateNewMapPoints and SearchInNeighbors' Fuse searches for them, creates, optimises, reads back and destroys its 32 local-BA windows and writes "
                "their poses back).  %s\n" % ("Statistics of the kernel launches of the LAST SIX TIMED STEPS only (cut from the trace by tools/step_table.py): no set-up, warm-up or pool-priming "
                "launches" if cut else "rocprofv3's --stats over the WHOLE process: the set-up's launches (key-frame stores, track sets, pool priming) are in the totals and skew the Percentage column"))
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            f.write(",".join([r["Name"].split("(")[0][:60], r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]),
                              "%.2f" % float(r["Percentage"]), r["MinNs"], r["MaxNs"]]) + "\n")
    bj = os.path.join(src, tag + "_bench.json")
    if os.path.exists(bj):
        line = [l for l in open(bj) if l.startswith("{")]
        if line:
            open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w").write(line[-1])
for extra, cmd in (("mapping", "python tools/prof_tri.py 8 20 5"), ("ba8", "python tools/prof_ba_many.py 8")):
    st = os.path.join(src, "%s_%s_kernel_stats.csv" % (tag, extra))
    if os.path.exists(st):
        with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, extra)), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- %s   (MI355X, 1 GPU)\n" % cmd)
            lg = os.path.join(src, "%s_%s.log" % (tag, extra))
            if os.path.exists(lg):
                for l in open(lg):
                    if "ms" in l and not l.startswith(("W2", "E2", "I2")):
                        f.write("# " + l.strip() + "\n")
            f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in csv.DictReader(open(st)):
                f.write(",".join([r["Name"].split("(")[0][:60], r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]),
                                  "%.2f" % float(r["Percentage"]), r["MinNs"], r["MaxNs"]]) + "\n")
import shutil
for name in ("step_table.md", "probe_lds_atomics.txt", "probe_global_atomics.txt", "probe_f64_pipes.txt", "rm_phase_cycles.txt", "ba16_track.txt", "ba16_track_valu.txt", "ba16_track_edges_only.txt", "ba16_random.txt", "ba1_track.txt"):
    sp = os.path.join(src, "%s_%s" % (tag, name))
    if os.path.exists(sp):
        txt = [l for l in open(sp) if not l.startswith(("W2", "E2", "I2", "/opt/amdgpu"))]
        open(os.path.join(dst, "%s_%s" % (tag, name)), "w").write("".join(txt))
pmc = {}
for kind in ("fetch", "write", "fetch_ba", "write_ba"):
    p = os.path.join(src, "%s_pmc_%s_counter_collection.csv" % (tag, kind))
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for r in csv.DictReader(open(p)):
        a = agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])]
        v = float(r["Counter_Value"])
        a[0] += v; a[1] += 1; a[2] = max(a[2], v)
    for (k, c), (v, n, mx) in agg.items():
        # local-BA launches of a finished window group do nothing: the busiest dispatch (all windows active) is the one to price
        pmc.setdefault(k, {})[c] = {"sum": v, "dispatches": n, "per_dispatch": (mx if kind.endswith("_ba") else v / n)}
if pmc:
    with open(os.path.join(dst, tag + "_pmc_hbm_traffic.json"), "w") as f:
        json.dump({"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/prof_frames.py %d 550 3  |  python tools/prof_ba_many.py 8" % PB,
                   "frames_per_dispatch": PB, "face": 550, "windows_per_dispatch": 8,
                   "note": "counter unit: KB (rocprofv3 derived metric); separate passes for FETCH_SIZE and WRITE_SIZE; "
                           "gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): bench.py doubles it; "
                           "calibration on this path: k_fast_cells with the XCD-striped cell list reports 183 MB against >= 300 MB of "
                           "non-zero cell bytes it must read (x2 = 366 MB incl. the 3-px halos); the plain row-major list reported 854 MB",
                   "ba_note": "kb_ba_* kernels: per_dispatch = the busiest dispatch (eight windows of K = 20, E = 80k active)",
                   "kernels": {k: v for k, v in pmc.items() if k.startswith(("k_", "kb_"))}}, f, indent=1)
# ---- SQ instruction mix (tools/pmc_mix.sh): per kernel, counters per wave and the vector-ALU issue bound
import glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fam in ("pmc", "pmc_ba"):
    for f in glob.glob(os.path.join(root, "gpurun_out", fam, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
mix = {}
for k, cs in sorted(acc.items()):
    if not k.startswith(("k_", "kb_")) or "SQ_WAVES" not in cs:
        continue
    # frame-path kernels: mean over the dispatches; local-BA kernels: the busiest dispatch (all eight windows active)
    m = {c: (max(v) if k.startswith("kb_") else sum(v) / len(v)) for c, v in cs.items()}
    w = m["SQ_WAVES"] or 1.0
    per = lambda c: round(m.get(c, 0.0) / w, 1)
    mix[k] = {"waves_per_dispatch": int(w),
              "per_wave": {"valu": per("SQ_INSTS_VALU"), "salu": per("SQ_INSTS_SALU"), "lds": per("SQ_INSTS_LDS"), "smem": per("SQ_INSTS_SMEM"),
                           "wave_quad_cycles": per("SQ_WAVE_CYCLES"), "wait_inst_any_quad_cycles": per("SQ_WAIT_INST_ANY"),
                           "lds_bank_conflict_cycles": per("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_cycles": per("SQ_LDS_IDX_ACTIVE"),
                           "wait_inst_lds_quad_cycles": per("SQ_WAIT_INST_LDS")},
              "valu_issue_bound_us": round(m.get("SQ_INSTS_VALU", 0.0) * 4 / (256 * 4 * 2.4e9) * 1e6, 1)}
    if m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) or m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0):
        mix[k]["mfma"] = {"f64_mops_per_wave": per("SQ_INSTS_VALU_MFMA_MOPS_F64"), "busy_cycles_per_wave": per("SQ_VALU_MFMA_BUSY_CYCLES"),
                          "insts_mfma_per_wave": per("SQ_INSTS_MFMA")}
if mix:
    with open(os.path.join(dst, tag + "_pmc_instruction_mix.json"), "w") as f:
        json.dump({"command": "rocprofv3 --kernel-trace --pmc <4 SQ counters per pass, 3 passes> -- python tools/prof_frames.py %d 550 2   (tools/pmc_mix.sh; MI355X; "
                              "one dispatch = %d frames; k_resize = average of its 7 per-level dispatches)" % (PB, PB), "frames_per_dispatch": PB,
                   "note": "valu_issue_bound_us = VALU instructions x 4 cycles (a wave64 VALU instruction occupies its SIMD16 for 4 cycles) / (256 CUs x 4 SIMDs x "
                           "2.4 GHz): the time the kernel would need if it did nothing but issue its vector ALU instructions.  SQ_WAVE_CYCLES / "
                           "SQ_WAIT_INST_ANY are in quad-cycles.",
                   "kernels": mix}, f, indent=1)
print("profiles/:", sorted(os.listdir(dst)))
