#!/bin/bash
# PMC passes over one bench step (lmax from $1, default 2): issue/stall/cache counters for the gather kernels.
L=${1:-2}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0; : > gpurun_out/pmc_step_raw.txt
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -d gpurun_out/pmc_step_$i -o r -- python bench.py --lmax $L --no-lmax4 --no-split --no-graph --no-workloads --steps 1 --warmup 1 --no-cpu-baseline --no-forward-only --no-live-traffic --no-static --lanes 1 > /dev/null 2>&1
  python tools/rocprof_summary.py gpurun_out/pmc_step_$i/r_results.db 2>/dev/null | grep -E "avg=" >> gpurun_out/pmc_step_raw.txt
  rm -rf gpurun_out/pmc_step_$i
done
python - <<'PY'
import re, collections
t = collections.defaultdict(dict)
for line in open("gpurun_out/pmc_step_raw.txt"):
    m = re.match(r"^(.*?)\s+(\S+)\s+avg=\s*([\d.]+)", line)
    if not m: continue
    k = re.sub(r"\(.*", "", m.group(1)).replace("void ", "").replace("gn::", "")[:48]
    if not re.search(r"msg_bwd|message_aggregate|htr_|attn_|gemm_f16x2", k): continue
    t[k][m.group(2)] = float(m.group(3))
names = sorted({c for v in t.values() for c in v})
for k, v in t.items():
    print(k)
    for c in names:
        if c in v: print(f"    {c:36s} {v[c]:16.0f}")
PY
