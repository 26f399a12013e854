"""Condense gpurun_out/$ROUND/* (scripts/collect_profiles.sh) into the committed profiles/ directory."""
import collections, csv, glob, json, os, shutil, subprocess, sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
ROUND = os.environ.get("ROUND", "r06")
O = R / "gpurun_out" / ROUND
P = R / "profiles"
P.mkdir(exist_ok=True)
rows = list(csv.DictReader(open(O / "stats" / "bench_kernel_stats.csv")))
with open(P / f"{ROUND}_bench_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows[:45]:
        w.writerow([r["Name"].split("(")[0][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
for n in ("bench_fwdbwd.json", "bench_fwd.json", "bench_under_rocprof.json", "bench_eval.json", "bench_c5_4m_fwdbwd.json",
          "bench_c5_4m_fwd.json", "bench_2rank_gloo_one_gpu.json", "bench_1rank_nccl.json", "bench_c5_4m_workloads.json"):
    if not (O / n).exists():
        continue
    txt = [l for l in (O / n).read_text().strip().splitlines() if l.startswith("{")]
    (P / (ROUND + "_" + n)).write_text(json.dumps(json.loads(txt[-1]), indent=1) + "\n")
subprocess.run([sys.executable, str(R / "scripts" / "make_pmc_json.py"), str(P / "pmc_latest.json"),
                str(O / "pmc_FETCH_SIZE" / "pmc_counter_collection.csv"), str(O / "pmc_WRITE_SIZE" / "pmc_counter_collection.csv")], check=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(str(O / "pmc_*" / "pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "s360" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(P / f"{ROUND}_pmc_counters.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
    for k in sorted(acc):
        for c in sorted(acc[k]):
            w.writerow([k, c, round(sum(acc[k][c]) / len(acc[k][c])), len(acc[k][c])])
# VALU-issue view of every kernel (SQ counters are per-SIMD quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs):
# a gfx950 SIMD issues one wave64 VALU instruction per 4 cycles, 1024 SIMDs on the chip.
pm = json.load(open(P / "pmc_latest.json"))
by_kernel = {v["kernel"]: v for k, v in pm.items() if k != "_meta" and isinstance(v, dict) and "kernel" in v}
for k, c in acc.items():
    kk = k.replace("s360::", "")
    if kk in by_kernel and "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        mean = lambda n: sum(c[n]) / len(c[n])
        cycles = mean("GRBM_GUI_ACTIVE") / 8.0
        by_kernel[kk]["valu_insts_per_launch"] = round(mean("SQ_INSTS_VALU")) if "SQ_INSTS_VALU" in c else None
        by_kernel[kk]["valu_busy_frac"] = round(mean("SQ_ACTIVE_INST_VALU") * 4.0 / (cycles * 1024.0), 4)
meta = json.loads((O / "meta.json").read_text())   # kernel-source hash + workload the counters belong to
meta.update(pm.get("_meta", {}))
meta["round"] = ROUND
pm["_meta"] = meta
for n in ("bwd_unit_timing.txt", "bwd_unit_timing_surface_like.txt", "fwd_unit_timing_encoder_like.txt", "fwd_unit_timing_surface_like.txt",
          "fwd_unit_timing_surface_like_unsplit.txt", "rccl_kernel_sequence.txt", "rccl_single_rank.log"):
    if (O / n).exists():
        shutil.copy(O / n, P / f"{ROUND}_{n}")
for leg in ("rccl", "adapter", "dropin", "surface"):      # kernel-stat tables of the side traces: top 30 rows each
    fs = glob.glob(str(O / leg / "**" / "*kernel_stats.csv"), recursive=True)
    if fs:
        rr = list(csv.DictReader(open(fs[0])))
        with open(P / f"{ROUND}_{leg}_kernel_stats.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
            for r in rr[:30]:
                nm = r["Name"]
                nm = (nm[:nm.index("(", 6)] if nm.startswith("void (") and "(" in nm[6:] else nm.split("(")[0])[:110]
                w.writerow([nm, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
calib = R / "gpurun_out" / "calib" / "calibration.json"
if calib.exists():
    shutil.copy(calib, P / "fetch_write_calibration.json")
json.dump(pm, open(P / "pmc_latest.json", "w"), indent=1, sort_keys=True)
print("profiles/ updated")
