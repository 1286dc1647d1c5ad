#!/usr/bin/env python3
"""Turn the rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.

    gpurun_out/prof_final/        rocprofv3 --kernel-trace --stats -- python3 bench.py ... (full workload)
    gpurun_out/pmc_FETCH_SIZE/    separate --pmc passes (one counter set per run), reduced token count
    gpurun_out/pmc_WRITE_SIZE/  gpurun_out/pmc_GRBM_GUI_ACTIVE/  gpurun_out/pmc_sq/
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
# on the GPU box the raw per-dispatch CSVs (> 100 MB) cannot travel back: tools/run_profiles.sh summarises there, into
# gpurun_out/profiles_out/, and the files are then copied to profiles/ here
P = sys.argv[2] if len(sys.argv) > 2 else f"{R}/profiles"
os.makedirs(P, exist_ok=True)

ks = glob.glob(f"{R}/gpurun_out/prof_final/*/*_kernel_stats.csv")[0]
shutil.copy(ks, f"{P}/{TAG}_bench_large-v3_b32_kernel_stats.csv")
tr = glob.glob(f"{R}/gpurun_out/prof_final/*/*_kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    agg[(r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
with open(f"{P}/{TAG}_bench_large-v3_b32_kernel_by_grid.csv", "w") as f:
    f.write("kernel,workgroups,calls,total_ms,avg_us,median_us,share_pct\n")
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        f.write('"%s",%d,%d,%.3f,%.2f,%.2f,%.2f\n' % (k, g, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, 100 * sum(v) / tot))


def by(d):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob(f"{R}/gpurun_out/{d}/*/*_counter_collection.csv")[0])):
        wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        a[(r["Kernel_Name"], wg)][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return a


fe, wr, sq, gr = by("pmc_FETCH_SIZE"), by("pmc_WRITE_SIZE"), by("pmc_sq"), by("pmc_GRBM_GUI_ACTIVE")
out = {}
lines = ["kernel,workgroups,launches,FETCH_SIZE_KB_raw,hbm_read_MB_corrected_x2,WRITE_MB,MFMA_busy_cycles,GRBM_GUI_ACTIVE_sum8xcd,clock_GHz,"
         "mfma_busy_frac_of_simd_cycles,wait_any,wait_inst_any,active_inst_any,lds_bank_conflict_cycles"]
keep = ("cross_attn", "gemm256", "encoder_attention", "dec_gemm", "layernorm", "gemm_kernel", "self_attn", "sampler", "mel_")
for key in sorted(fe, key=lambda k: -sum(x[0] for x in fe[k]["FETCH_SIZE"])):
    k, wg = key
    if not any(s in k for s in keep):
        continue
    f = [x[0] for x in fe[key]["FETCH_SIZE"]]
    w = [x[0] for x in wr.get(key, {}).get("WRITE_SIZE", [(0, 0)])]
    q, g = sq.get(key, {}), gr.get(key, {})
    n = len(f)
    avg = lambda name: (sum(x[0] for x in q[name]) / len(q[name])) if name in q and q[name] else 0.0  # noqa: E731
    gui = (sum(x[0] for x in g["GRBM_GUI_ACTIVE"]) / len(g["GRBM_GUI_ACTIVE"])) if "GRBM_GUI_ACTIVE" in g else 0.0
    dur = (sum(x[1] for x in g["GRBM_GUI_ACTIVE"]) / len(g["GRBM_GUI_ACTIVE"])) if "GRBM_GUI_ACTIVE" in g else 0.0
    cyc = gui / 8.0
    # GRBM_GUI_ACTIVE / duration reads high on short dispatches (MI355X_MICROARCH.md "DVFS give-back": the quotient is
    # only meaningful from about 0.3 ms; round 1's table showed 4-13 GHz for sub-10 us kernels): the clock and the MFMA-busy
    # fraction derived from it are left empty below 20 us
    short = dur < 20e3
    clock = cyc / dur if dur > 0 else 0.0
    mfma = avg("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc) if cyc > 0 else 0.0
    wc = avg("SQ_WAVE_CYCLES") or 1.0
    lines.append('"%s",%d,%d,%.1f,%.2f,%.3f,%.4g,%.4g,%s,%s,%.3f,%.3f,%.3f,%.4g' % (
        k, wg, n, sum(f) / n, 2 * sum(f) / n / 1024, sum(w) / len(w) / 1024, avg("SQ_VALU_MFMA_BUSY_CYCLES"), gui,
        "n/a(<20us)" if short else "%.3f" % clock, "n/a(<20us)" if short else "%.3f" % mfma,
        avg("SQ_WAIT_ANY") / wc, avg("SQ_WAIT_INST_ANY") / wc, avg("SQ_ACTIVE_INST_ANY") / wc, avg("SQ_LDS_BANK_CONFLICT")))
    if "cross_attn" in k and wg == 640:
        out["3"] = {"kernel": "cross_attn_kernel", "batch": 32, "hbm_bytes_per_launch": int(2 * sum(f) / n * 1024 + sum(w) / len(w) * 1024),
                    "fetch_size_kb_raw": sum(f) / n, "write_size_kb": sum(w) / len(w),
                    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                            "(gfx950 reports half of a wide coalesced read); single-token steps only (640 workgroups)"}
open(f"{P}/{TAG}_pmc_by_kernel.csv", "w").write("\n".join(lines) + "\n")
json.dump(out, open(f"{P}/{TAG}_pmc_summary.json", "w"), indent=1)
print("\n".join(l[:200] for l in lines[:12]))
