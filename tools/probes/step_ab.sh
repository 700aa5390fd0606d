# usage: bash tools/probes/step_ab.sh <variant-name or ""> ...   -> per-kernel breakdown of one step (lmax 2), one step at a time
for v in "$@"; do
  if [ "$v" = "product" ]; then unset GN_LIB_PATH; else export GN_LIB_PATH=gotennet_amd/variants/lib_$v.so; fi
  echo "=== $v"
  timeout 300 python bench.py --steps 10 --lanes 1 --no-cpu-baseline --no-split --no-workloads --no-graph --no-forward-only --no-live-traffic --no-static --no-lmax4 --breakdown 2>&1 >/dev/null | grep -E "gn_gemm\[(54368x256x256|8064x256x768|21504x256x256\+)|per-kernel" 
  timeout 300 python bench.py --steps 10 --lanes 1 --no-cpu-baseline --no-split --no-workloads --no-graph --no-forward-only --no-live-traffic --no-static --no-lmax4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'])"
done
