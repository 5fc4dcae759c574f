#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py in separate passes (counters only + kernel trace:
# never combined with sys/runtime traces).  Usage: tools/pmc_passes.sh <outdir> [bench args...]
# PMC_CMD="python tools/gemm_bench.py" PMC_KERNELS="gemm_|wgrad_" profiles another command / kernel set.
set -u
OUT=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
while read -r CTRS; do
  [ -z "$CTRS" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT/pass$i" -o pmc --output-format csv -- \
      ${PMC_CMD:-python "$R/bench.py" --no-cpu-baseline} "$@" > "$OUT/pass$i.log" 2>&1
done <<'LIST'
SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
FETCH_SIZE TCC_ATOMIC
WRITE_SIZE TCC_HIT TCC_MISS
TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_ATOMIC TCC_REQ
SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F16
LIST
python "$R/tools/pmc_summary.py" "$OUT" "${PMC_KERNELS:-raster_|rs_|project|tile_|scan_|reduce_rows|gather_grec|make_grec|seg_|slot_rows|dot_}" "$OUT/traffic.json" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
