for pz in 0 512 768; do echo "== GN_GEMM_PERSIST=$pz"; GN_GEMM_PERSIST=$pz GN_GEMM_MODE=f32 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | head -7; done
