#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats + PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes) for
# lmax 2 and 4 in the default projection mode, kernel stats of the exact-fp32 and bf16x3 modes; summaries under gpurun_out/profiles/.
#   bash tools/profile_round.sh r03
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-lmax4 --no-split --no-graph --no-workloads --no-cpu-baseline --no-forward-only --no-live-traffic --no-static --lanes 1"
for L in 2 4; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats_l$L -o r -- $B --lmax $L --steps 5 --warmup 2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch_l$L -o r -- $B --lmax $L --steps 2 --warmup 1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write_l$L -o r -- $B --lmax $L --steps 2 --warmup 1 > /dev/null 2>&1
done
# BASELINE configs[2] (Ac-Ala3-NHMe, 64 molecules, lmax 2) and configs[4] (nanotube, 8 molecules, lmax 3): traffic only
for W in "md22_ac_ala3 64 2" "md22_nanotube 8 3"; do
  set -- $W
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch_$1 -o r -- $B --workload $1 --batch $2 --lmax $3 --steps 2 --warmup 1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write_$1 -o r -- $B --workload $1 --batch $2 --lmax $3 --steps 2 --warmup 1 > /dev/null 2>&1
done
GN_GEMM_MODE=f32 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats_f32 -o r -- $B --lmax 2 --steps 5 --warmup 2 > /dev/null 2>&1
GN_GEMM_MODE=split rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats_bf16x3 -o r -- $B --lmax 2 --steps 5 --warmup 2 > /dev/null 2>&1

# matrix-pipe occupancy inside the step (lmax 2): SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE) per projection kernel
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_mfma_l2 -o r -- $B --lmax 2 --steps 2 --warmup 1 > /dev/null 2>&1

# steady-state dispatches per step (one-time packing cancels between a 3-step and an 11-step trace)
rocprofv3 --kernel-trace -d gpurun_out/prof_disp_a -o r -- $B --lmax 2 --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace -d gpurun_out/prof_disp_b -o r -- $B --lmax 2 --steps 10 --warmup 1 > /dev/null 2>&1

# summarise on the box (the rocpd databases are too large to ship back) and drop the databases
mkdir -p gpurun_out/profiles
for L in 2 4; do
  python tools/rocprof_summary.py gpurun_out/prof_stats_l$L/r_results.db 2>/dev/null > gpurun_out/profiles/${R}_kernel_stats_lmax$L.txt
  python tools/rocprof_summary.py gpurun_out/prof_fetch_l$L/r_results.db 2>/dev/null | grep -E "^# source|^# PMC|gn::" > gpurun_out/profiles/${R}_pmc_fetch_size_lmax$L.txt
  python tools/rocprof_summary.py gpurun_out/prof_write_l$L/r_results.db 2>/dev/null | grep -E "^# source|^# PMC|gn::" > gpurun_out/profiles/${R}_pmc_write_size_lmax$L.txt
done
for W in "md22_ac_ala3 64 2" "md22_nanotube 8 3"; do
  set -- $W
  python tools/rocprof_summary.py gpurun_out/prof_fetch_$1/r_results.db 2>/dev/null | grep -E "^# source|^# PMC|gn::" > gpurun_out/profiles/${R}_pmc_fetch_size_$1_b$2_lmax$3.txt
  python tools/rocprof_summary.py gpurun_out/prof_write_$1/r_results.db 2>/dev/null | grep -E "^# source|^# PMC|gn::" > gpurun_out/profiles/${R}_pmc_write_size_$1_b$2_lmax$3.txt
done
python tools/rocprof_summary.py gpurun_out/prof_mfma_l2/r_results.db 2>/dev/null | grep -E "^# source|^# PMC|gn::gemm" > gpurun_out/profiles/${R}_pmc_mfma_busy_lmax2.txt
python tools/rocprof_summary.py gpurun_out/prof_stats_f32/r_results.db 2>/dev/null > gpurun_out/profiles/${R}_kernel_stats_lmax2_exact_f32.txt
python tools/rocprof_summary.py gpurun_out/prof_stats_bf16x3/r_results.db 2>/dev/null > gpurun_out/profiles/${R}_kernel_stats_lmax2_bf16x3.txt
python tools/dispatch_per_step.py gpurun_out/prof_disp_a/r_results.db gpurun_out/prof_disp_b/r_results.db 8 > gpurun_out/profiles/${R}_dispatches_per_step_lmax2.txt 2>&1
rm -rf gpurun_out/prof_disp_a gpurun_out/prof_disp_b
rm -rf gpurun_out/prof_stats_l* gpurun_out/prof_fetch_* gpurun_out/prof_write_* gpurun_out/prof_stats_f32 gpurun_out/prof_stats_bf16x3 gpurun_out/prof_mfma_l2
ls -la gpurun_out/profiles
