for m in f32 split; do echo "== GN_GEMM_MODE=$m"; GN_GEMM_MODE=$m python tools/gemm_bench.py 2>&1 | grep -v amdgpu; done
