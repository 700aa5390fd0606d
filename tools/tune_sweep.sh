#!/bin/bash
# Occupancy-hint sweep for the gather kernels (gn_tune.h).  Stage 1 (here, CPU): build one library per
# (kernel, waves/SIMD) pair.  Stage 2 (GPU box): `bash tools/tune_sweep.sh run` benches each library.
cd "$(dirname "$0")/.."
VARIANTS="${VARIANTS:-MSG_SRC=1 MSG_SRC=2 MSG_SRC=3 MSG_SRC=4 MSG_TGT=3 MSG_TGT=4 HTR_TGT=2 HTR_TGT=3 HTR_TGT=4 HTR_SRC=2 HTR_SRC=3 HTR_EDGE=2 HTR_EDGE=4 ATTN=2 ATTN=4 MSG_SRC_G=2 MSG_SRC_G=3 MSG_TGT_G=2 MSG_TGT_G=3 HTR_TGT_G=2 HTR_TGT_G=3 HTR_SRC_G=2 HTR_SRC_G=3 K6=3 K6_G=3}"
if [ "$1" != "run" ]; then
  mkdir -p gotennet_amd/sweep
  build_one() {
    v=$1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DGN_W_${v} \
        gotennet_amd/csrc/*.hip -o gotennet_amd/sweep/lib_${v}.so 2>/dev/null || echo "build failed: $v"
  }
  export -f build_one
  echo $VARIANTS | tr ' ' '\n' | xargs -P ${JOBS:-6} -I{} bash -c 'build_one {}'
  ls gotennet_amd/sweep | wc -l
else
  cp gotennet_amd/libgotennet_hip.so /tmp/base.so
  one() {
    python bench.py --no-cpu-baseline --no-split --no-graph --breakdown 2>gpurun_out/bd.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-12s lmax2 %.3f ms   lmax4 %.3f ms' % ('$1', d['ms_per_step'], d['also']['lmax4']['ms_per_step']))"
    grep -E "message_backward|htr_backward|message_aggregate|htr_edge|attn_softmax" gpurun_out/bd.txt | awk '{printf "      %-24s %s us\n", $1, $6}'
  }
  one base
  for v in $VARIANTS; do
    cp gotennet_amd/sweep/lib_${v}.so gotennet_amd/libgotennet_hip.so
    one $v
  done
  cp /tmp/base.so gotennet_amd/libgotennet_hip.so
fi
