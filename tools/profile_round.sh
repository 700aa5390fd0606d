#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats + PMC passes for lmax 2 and 4 (outputs under gpurun_out/).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in 2 4; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats_l$L -o r -- python bench.py --lmax $L --no-lmax4 --no-split --no-graph --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch_l$L -o r -- python bench.py --lmax $L --no-lmax4 --no-split --no-graph --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write_l$L -o r -- python bench.py --lmax $L --no-lmax4 --no-split --no-graph --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done

# summarise on the box (the rocpd databases are too large to ship back) and drop the databases
mkdir -p gpurun_out/profiles
for L in 2 4; do
  python tools/rocprof_summary.py gpurun_out/prof_stats_l$L/r_results.db 2>/dev/null > gpurun_out/profiles/r01_kernel_stats_lmax$L.txt
  python tools/rocprof_summary.py gpurun_out/prof_fetch_l$L/r_results.db 2>/dev/null | grep -E "^# source|^# PMC|gn::" > gpurun_out/profiles/r01_pmc_fetch_size_lmax$L.txt
  python tools/rocprof_summary.py gpurun_out/prof_write_l$L/r_results.db 2>/dev/null | grep -E "^# source|^# PMC|gn::" > gpurun_out/profiles/r01_pmc_write_size_lmax$L.txt
done
rm -rf gpurun_out/prof_stats_l* gpurun_out/prof_fetch_l* gpurun_out/prof_write_l*
ls -la gpurun_out/profiles
