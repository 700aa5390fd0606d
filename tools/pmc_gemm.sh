#!/bin/bash
# PMC passes over one GEMM shape (GM/GN/GK env, default the 54368x1536x256 edge projection; GN_GEMM_MODE picks the
# arithmetic, default 2xfp16-split): issue/stall breakdown.  One counter set per pass (--kernel-trace + --pmc only).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for SET in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -d gpurun_out/pmc_gemm_$i -o r -- python tools/gemm_one.py > /dev/null 2>&1
  python tools/rocprof_summary.py gpurun_out/pmc_gemm_$i/r_results.db 2>/dev/null | grep -E "^# PMC|gn::gemm_|^kernel" | grep -v "counters_collection" | head -12
  rm -rf gpurun_out/pmc_gemm_$i
done
