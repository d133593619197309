#!/bin/bash
# Collects the rocprofv3 evidence for one build (run on the GPU box from the repo root):
#   bash profiles/run_profiles.sh r02
# 1. an un-profiled run measures the split-GEMM tiles and saves them (DFD_TILE_CACHE), so that the profiled runs
#    launch no tuning candidates; 2. kernel trace + stats of bench.py (the command BENCH_rNN records, minus the
#    extras) and of the classifier-only / e2e drivers; 3. PMC passes - FETCH_SIZE and WRITE_SIZE in separate
#    passes, SQ counters in two passes of 8 - of the classifier-only driver (fp32 and bf16).
set -u
TAG=${1:-r02}
MODE=${2:-all}          # all | bench (kernel stats of bench.py only) | e2e (kernel stats of bench / e2e / MTCNN only: what changes when the classifier kernels do not)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export DFD_TILE_CACHE=$OUT/tiles.txt
export TMPDIR=/tmp
python3 profiles/b0_profile_driver.py > "$OUT/warm_fp32.log" 2>&1 || exit 1
B0_BF16=1 python3 profiles/b0_profile_driver.py > "$OUT/warm_bf16.log" 2>&1 || exit 1
python3 bench.py --steps 20 --warmup 5 --no-e2e --no-streams --no-cpu-baseline > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err" || exit 1
# the MTCNN / e2e drivers once un-profiled too: their handles warm up other batch shapes, and a shape missing from the
# tile cache is measured inside the profiled process (round 2's MTCNN table was dominated by those tuning launches)
MT_FRAMES=64 python3 profiles/mtcnn_profile_driver.py > "$OUT/warm_mtcnn64.log" 2>&1 || exit 1
MT_DENSE=1 python3 profiles/mtcnn_profile_driver.py > "$OUT/warm_mtcnn_dense.log" 2>&1 || exit 1
E2E_STEPS=2 python3 profiles/e2e_profile_driver.py > "$OUT/warm_e2e.log" 2>&1 || exit 1
cd /tmp
run() { name=$1; shift; rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || { echo "rocprofv3 $name failed"; tail -5 "$OUT/$name.log"; exit 1; }; }
# bench.py keeps two forwards in flight (its instrumented step runs alone): the trace of the default command averages
# kernels that overlap another forward's; DFD_BENCH_LANES=1 is the same loop with one forward in flight - the isolated
# durations bench.py's events (and its roofline object) report.  Both summaries are kept.
run stats_bench_lanes2 --kernel-trace --stats -d "$OUT/stats_bench_lanes2" -o s --output-format csv -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-e2e --no-streams --no-cpu-baseline
export DFD_BENCH_LANES=1
run stats_bench --kernel-trace --stats -d "$OUT/stats_bench" -o s --output-format csv -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-e2e --no-streams --no-cpu-baseline
unset DFD_BENCH_LANES
if [ "$MODE" = "bench" ]; then cd "$ROOT"; find "$OUT" -name "*.csv" | sort; exit 0; fi
export MT_FRAMES=64
run stats_mtcnn_selective --kernel-trace --stats -d "$OUT/stats_mtcnn_selective" -o s --output-format csv -- python3 "$ROOT/profiles/mtcnn_profile_driver.py"
unset MT_FRAMES
export MT_DENSE=1
run stats_mtcnn_stress --kernel-trace --stats -d "$OUT/stats_mtcnn_stress" -o s --output-format csv -- python3 "$ROOT/profiles/mtcnn_profile_driver.py"
unset MT_DENSE
run stats_e2e --kernel-trace --stats -d "$OUT/stats_e2e" -o s --output-format csv -- python3 "$ROOT/profiles/e2e_profile_driver.py"
if [ "$MODE" = "e2e" ]; then cd "$ROOT"; find "$OUT" -name "*.csv" | sort; exit 0; fi
run stats_b0 --kernel-trace --stats -d "$OUT/stats_b0" -o s --output-format csv -- python3 "$ROOT/profiles/b0_profile_driver.py"
export B0_BF16=1
run stats_b0_bf16 --kernel-trace --stats -d "$OUT/stats_b0_bf16" -o s --output-format csv -- python3 "$ROOT/profiles/b0_profile_driver.py"
unset B0_BF16
run pmc_fetch --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o p --output-format csv -- python3 "$ROOT/profiles/b0_profile_driver.py"
run pmc_write --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o p --output-format csv -- python3 "$ROOT/profiles/b0_profile_driver.py"
run pmc_sqa --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace -d "$OUT/pmc_sqa" -o p --output-format csv -- python3 "$ROOT/profiles/b0_profile_driver.py"
run pmc_sqb --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES --kernel-trace -d "$OUT/pmc_sqb" -o p --output-format csv -- python3 "$ROOT/profiles/b0_profile_driver.py"
export B0_BF16=1
run pmc_fetch_bf16 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch_bf16" -o p --output-format csv -- python3 "$ROOT/profiles/b0_profile_driver.py"
run pmc_write_bf16 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write_bf16" -o p --output-format csv -- python3 "$ROOT/profiles/b0_profile_driver.py"
unset B0_BF16
cd "$ROOT"
find "$OUT" -name "*.csv" | sort
