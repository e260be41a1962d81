"""Turns the raw outputs of tools/collect_profiles_rNN.sh (gpurun_out/profiles_rNN/) into the files committed under profiles/:
rNN_* copies of the rocprofv3 summaries and profiles/pmc_summary.json (what bench.py's `roofline.traffic` reads).
    python tools/summarise_profiles.py [r03]"""
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = os.path.join(ROOT, "gpurun_out", "profiles_" + RND)
DST = os.path.join(ROOT, "profiles")


def main():
    for name in ("bench_kernel_stats.csv", "sphere_kernel_stats.csv", "backward_kernel_stats.csv", "sweep_kernel_stats.csv",
                 "config5_kernel_stats.csv", "bench_under_rocprof.json", "pmc_raw.json", "sweep.log", "backward.log",
                 "recon_kernel_stats.csv", "recon.log", "hd_gabo_kernel_stats.csv", "hd_gabo.log", "gp_mll_kernel_stats.csv", "gp_mll.log",
                 "recon_native_kernel_stats.csv", "recon_native.log", "pmc_sphere.json", "pmc_headline.json", "pmc_tr_solve.json"):
        p = os.path.join(SRC, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, f"{RND}_{name}"))
    raw = json.load(open(os.path.join(SRC, "pmc_raw.json")))
    kern = "gabo::spd_ai_pairwise_kernel<10>"
    # per-dispatch durations of the pairwise kernel in the bench run
    durs = []
    for r in csv.DictReader(open(os.path.join(SRC, "bench_kernel_trace.csv"))):
        if r["Kernel_Name"].startswith("void gabo::spd_ai_pairwise_kernel<10>") or r["Kernel_Name"].startswith("gabo::spd_ai_pairwise_kernel<10>"):
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    line = json.loads(open(os.path.join(SRC, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    mirror = [k for k in raw["sym_fetch"] if "mirror_upper" in k][0]
    n = 4096
    known = (n * (n - 1) // 2 + n * 31) * 8.0            # the upper triangle read once (32 x 32 tiles touching the diagonal read whole)
    factor = known / (raw["sym_fetch"][mirror]["FETCH_SIZE"] * 1024.0)
    fetch = raw["full_fetch"][kern]["FETCH_SIZE"] * 1024.0 * factor
    write = raw["full_write"][kern]["WRITE_SIZE"] * 1024.0
    sq = raw["full_sq"][kern]
    pairs = n * n
    out = {
        "round": int(RND[1:]), "kernel": kern,
        "workload": "N=4096, d=10, all N^2 pairs (bench.py `value` launches; tools/prof_spd.py 4096 10 x for the PMC passes)",
        "rocprof_kernel_trace_ms": {"avg": sum(durs) / len(durs), "median": statistics.median(durs), "min": min(durs), "max": max(durs),
                                    "calls": len(durs), "note": "all dispatches of the bench command: 60 disclosed preheat launches (the first ~20 at a lower clock), 5 warm-up steps, 20 timed steps"},
        "bench_hip_event_ms_per_step": line["roofline"]["kernel_ms"],
        "FETCH_SIZE_KB_raw": raw["full_fetch"][kern]["FETCH_SIZE"], "WRITE_SIZE_KB_raw": raw["full_write"][kern]["WRITE_SIZE"],
        "fetch_calibration": {"kernel": mirror + " (8 B/lane loads of a known byte count)", "known_bytes": known,
                              "FETCH_SIZE_KB_raw": raw["sym_fetch"][mirror]["FETCH_SIZE"], "factor": factor,
                              "note": "MI355X_MICROARCH.md: FETCH_SIZE under-reports coalesced streaming reads by 2x on gfx950; "
                                      "calibrated here on 8 B/lane loads"},
        "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
        "bytes_per_pair": (fetch + write) / pairs, "compulsory_bytes_per_pair": (pairs * 8.0 + 2 * n * 55 * 8) / pairs,
        "sq": sq,
        "valu_insts_per_wave_row_of_64_pairs": sq["SQ_INSTS_VALU"] / (pairs / 64.0),
        "valu_insts_per_wave_row_round1": 6163.05,
    }
    sph = [k for k in raw.get("sphere_sq", {}) if "sphere_pairwise_kernel" in k]
    if sph:
        s = raw["sphere_sq"][sph[0]]
        out["sphere"] = {"kernel": sph[0], "sq": s, "valu_insts_per_output": s["SQ_INSTS_VALU"] / (pairs / 64.0),
                         "valu_insts_per_output_round1": 102,
                         "hbm_write_bytes_per_launch": raw["sphere_write"][sph[0]]["WRITE_SIZE"] * 1024.0}
    # the write-bound Gram kernels of config 5 (N = 4096 latent 2 x 2 points): instructions per output and write bytes
    for key, name in (("nested_ai_gram", "spd_ai_gauss2_kernel"), ("log_euclidean_gram", "frobenius_pairwise_kernel")):
        ks = [k for k in raw.get("config5_sq", {}) if name in k] if isinstance(raw.get("config5_sq"), dict) else []
        if ks:
            s = raw["config5_sq"][ks[0]]
            row = {"kernel": ks[0], "sq": s, "valu_insts_per_output": s["SQ_INSTS_VALU"] / (pairs / 64.0)}
            kw = [k for k in raw.get("config5_write", {}) if name in k] if isinstance(raw.get("config5_write"), dict) else []
            if kw:
                row["hbm_write_bytes_per_launch"] = raw["config5_write"][kw[0]]["WRITE_SIZE"] * 1024.0
            out[key] = row
    json.dump(out, open(os.path.join(DST, "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
