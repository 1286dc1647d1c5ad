#!/bin/bash
# Round profile set, run on the GPU box:   gpurun --timeout 1200 -- bash tools/run_profiles.sh
# then here:                                cp gpurun_out/profiles_out/* profiles/
# One kernel-trace pass of the full bench workload, then PMC passes (one counter set per pass, kernel trace only,
# reduced token count: counters serialise the dispatches) as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes.
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/prof_final $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_GRBM_GUI_ACTIVE $O/pmc_sq
echo "[profiles] bench bf16"; python3 bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err
echo "[profiles] bench f16";  python3 bench.py --steps 20 --warmup 5 --dtype f16 --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
# Kernel durations are a roofline quantity only when a kernel has the chip to itself: the trace the per-kernel figures come
# from runs the sequential schedule (as bench.py's own roofline leg does); a second trace of the default command (four
# decodes of two batches each side by side: a kernel's duration there includes its neighbours' traffic) is kept beside it for the record.
echo "[profiles] kernel trace (sequential schedule)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -- python3 bench.py --phases 0 --pipeline 0 --steps 3 --warmup 0 --no-cpu-baseline --no-latency > $O/prof_final.log 2>&1
echo "[profiles] kernel trace (default command: lanes)"
rm -rf $O/prof_lanes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lanes -- python3 bench.py --steps 8 --warmup 0 --no-cpu-baseline --no-latency > $O/prof_lanes.log 2>&1
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  echo "[profiles] pmc $c"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --phases 0 --pipeline 0 --steps 1 --warmup 0 --tokens 6 --no-cpu-baseline --no-latency > $O/pmc_$c.log 2>&1
done
echo "[profiles] pmc sq"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -- python3 bench.py --phases 0 --pipeline 0 --steps 1 --warmup 0 --tokens 6 --no-cpu-baseline --no-latency > $O/pmc_sq.log 2>&1
# the per-dispatch CSVs (> 100 MB) cannot travel back (64 MiB limit): summarise here, keep only the summaries
TAG=${1:-r02}
python3 tools/summarize_profiles.py $TAG $O/profiles_out > $O/profiles_out.log 2>&1
cp $(ls $O/prof_lanes/*/*_kernel_stats.csv | head -1) $O/profiles_out/${TAG}_bench_large-v3_b32_lanes_kernel_stats.csv
python3 tools/lane_gap_analysis.py $O/prof_lanes > $O/profiles_out/${TAG}_lane_gaps.txt 2>&1 || true
cp $O/bench_final.json $O/profiles_out/${TAG}_bench_large-v3_b32.json
cp $O/bench_f16.json $O/profiles_out/${TAG}_bench_large-v3_b32_f16.json
rm -rf $O/prof_final $O/prof_lanes $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_GRBM_GUI_ACTIVE $O/pmc_sq
ls -la $O/profiles_out
echo "[profiles] done"
