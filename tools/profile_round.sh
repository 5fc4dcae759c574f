#!/bin/bash
# One GPU-box call that refreshes everything under profiles/ for a round:
#   kernel-trace stats, PMC passes, the bench line (with the CPU baseline), the chunk statistics of the rows kernel, the D=16 iteration (+ its
#   kernel stats), the decoder GEMMs alone, the config sweep, the union-of-rows statistics and the micro-benchmarks.  Usage (from the repo root, on the GPU box): tools/profile_round.sh r01
set -u
TAG=${1:-r01}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$O/stats" -o c3 --output-format csv -- \
    python "$R/bench.py" --no-cpu-baseline --no-heavy --steps 5 --warmup 2 > "$O/stats.log" 2>&1
S=$(find "$O/stats" -name '*kernel_stats.csv' | head -1)
[ -n "$S" ] && python "$R/tools/trim_stats.py" "$S" "$O/${TAG}_c3_kernel_stats.csv" 40
"$R/tools/pmc_passes.sh" "$O/pmc" --steps 3 --warmup 1 --no-heavy > /dev/null 2>&1
cp "$O/pmc/summary.txt" "$O/${TAG}_c3_pmc_summary.txt" 2>/dev/null
cp "$O/pmc/traffic.json" "$O/${TAG}_c3_pmc_traffic.json" 2>/dev/null
cd "$R"
timeout 900 python bench.py --steps 20 --warmup 3 > "$O/${TAG}_c3_bench.json" 2> "$O/bench.err"
timeout 300 python tools/chunk_stats.py C3 > "$O/${TAG}_c3_chunk_stats.txt" 2>&1
timeout 300 python tools/row_mask_stats.py C3 > "$O/${TAG}_c3_row_mask_stats.txt" 2>&1
timeout 300 python tools/row_mask_stats.py C3H >> "$O/${TAG}_c3_row_mask_stats.txt" 2>&1
timeout 300 python tools/slot_support.py C3 > "$O/${TAG}_c3_slot_support.txt" 2>&1
timeout 300 python tools/decoder_bench.py > "$O/${TAG}_d16_iteration_exact.json" 2>/dev/null
timeout 300 python tools/decoder_bench.py --bf16x2 > "$O/${TAG}_d16_iteration_bf16x2.json" 2>/dev/null
timeout 300 python tools/decoder_bench.py --f16 > "$O/${TAG}_d16_iteration_f16.json" 2>/dev/null
timeout 300 python tools/decoder_bench.py --bf16 > "$O/${TAG}_d16_iteration.json" 2>/dev/null
timeout 600 python tools/exp_fwd.py --config C3 --truth 16 > "$O/${TAG}_fwd_accuracy.json" 2>/dev/null
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$O/d16rstats" -o d16r --output-format csv -- \
    python "$R/tools/config_sweep.py" 'C3 geometry D=16$' > /dev/null 2>&1 )
S=$(find "$O/d16rstats" -name '*kernel_stats.csv' | head -1)
[ -n "$S" ] && python "$R/tools/trim_stats.py" "$S" "$O/${TAG}_d16_raster_kernel_stats.csv" 45
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$O/d16stats" -o d16 --output-format csv -- \
    python "$R/tools/decoder_bench.py" --f16 > /dev/null 2>&1 )
S=$(find "$O/d16stats" -name '*kernel_stats.csv' | head -1)
[ -n "$S" ] && python "$R/tools/trim_stats.py" "$S" "$O/${TAG}_d16_iteration_kernel_stats.csv" 40
T=$(find "$O/d16stats" -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python "$R/tools/iter_timeline.py" "$T" > "$O/${TAG}_d16_iteration_timeline.txt" 2>&1
timeout 300 python tools/gemm_bench.py > "$O/${TAG}_decoder_gemm.json" 2>/dev/null
timeout 900 python tools/config_sweep.py > "$O/${TAG}_config_sweep.json" 2>/dev/null
timeout 300 python tools/union_rows.py C3 > "$O/${TAG}_union_rows.json" 2>/dev/null
timeout 300 python tools/dp_overhead.py 128 256 512 256,128,128 > "$O/${TAG}_dp_overhead.json" 2>/dev/null
tools/micro/run_all.sh > "$O/${TAG}_micro.txt" 2>&1
tail -1 "$O/${TAG}_c3_bench.json" | cut -c1-600
