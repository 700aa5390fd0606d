cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GN_GEMM_MODE=split GM=131072 GN=512 GK=1024
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS -d gpurun_out/pmc_s1 -o g -- python tools/gemm_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d gpurun_out/pmc_s2 -o g -- python tools/gemm_one.py > /dev/null 2>&1
python - <<'PY'
import sqlite3
for d in ("pmc_s1","pmc_s2"):
    db = sqlite3.connect(f"gpurun_out/{d}/g_results.db"); cur = db.cursor()
    for r in cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%gemm%' group by kernel_name, counter_name"):
        print(r[0][:40], r[1], f"{r[2]:.4g}", r[3], f"{r[4]/1e3:.1f}us")
PY
