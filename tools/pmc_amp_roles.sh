#!/bin/bash
# PMC passes over the fused AMP layer (C = 64, k = 11) with its phases switched off: why do a Snake-only and a conv-only
# workgroup of the same CU not overlap?  Output: gpurun_out/r05/pmc_amp_<skip>.txt
out=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for skip in 0 5 10 64; do
  rm -rf /tmp/p_a /tmp/p_b
  CASES=1 SKIPS=$skip timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_a -- python $R/tools/bench_amp_phases.py > $out/pmc_a.log 2>&1
  CASES=1 SKIPS=$skip timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/p_b -- python $R/tools/bench_amp_phases.py > $out/pmc_b.log 2>&1
  (echo "# PTPP_AMP_SKIP=$skip, C = 64, k = 11, d = 3, B = 64"; python $R/tools/pmc_summary.py /tmp/p_a | grep -A12 amp_fused; python $R/tools/pmc_summary.py /tmp/p_b | grep -A12 amp_fused) > $out/pmc_amp_$skip.txt 2>&1
done
cat $out/pmc_amp_*.txt
