#!/bin/bash
# Counter-level comparison of the gradient-GEMM dataflows: tools/probe/kbench_res runs k_bgemm64 (the library's
# kernel) and the three resident-accumulator forms k_grad_res / k_grad_q / k_grad_h with their ablations at config-3 shape; this
# script collects SQ / TA / TCP counters per kernel symbol in separate rocprofv3 --pmc passes (no tracing domains beside the counters)
# and tools/pmc_variant_summary.py prints them side by side.  Also: the one-workgroup Householder chain (probe_eigh_np) for its LDS behaviour.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_grad
mkdir -p $out
declare -A PASS
PASS[a]="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"
PASS[b]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
PASS[c]="SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
PASS[d]="TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_READ_LDS_WAVEFRONTS GRBM_GUI_ACTIVE"
PASS[e]="TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_LFIFO_STALL_CYCLES GRBM_GUI_ACTIVE"
PASS[f]="SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU"
for p in ${1:-a b c d e f}; do
  rocprofv3 --pmc ${PASS[$p]} --kernel-trace --output-format csv -d $out -o kb_$p -- $R/tools/probe/kbench_res 60000 > $out/kb_$p.log 2>&1 || echo "pass $p (kbench_res) failed: $(tail -2 $out/kb_$p.log)"
done
for p in a f; do
  rocprofv3 --pmc ${PASS[$p]} --kernel-trace --output-format csv -d $out -o eg_$p -- $R/tools/probe/probe_eigh_np > $out/eg_$p.log 2>&1 || echo "pass $p (probe_eigh_np) failed"
done
python $R/tools/pmc_variant_summary.py $out > $out/summary.txt 2>&1
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -size +30M -delete
cat $out/summary.txt
