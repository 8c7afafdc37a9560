#!/bin/bash
# usage (on the GPU box): tools/attic/pmc_dcn.sh > gpurun_out/pmc_dcn.txt   — three counter-only passes over the DCN kernels in isolation
echo "# rocprofv3 --kernel-trace --pmc <counters> -- python tools/attic/prof_conv.py dcn2   (separate counter-only passes; N=64 bf16, forward+backward of"
echo "# DCN 128->64@64^2, 128->128@64^2, 256->128@32^2, 64->64@128^2; every dispatch of a dcn_* kernel; tuple = (dispatch, kernel, duration ns)"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  echo "## --pmc $set"
  bash tools/attic/pmc_run.sh dcn2 dcn_ $set
done
