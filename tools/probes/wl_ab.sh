# usage: bash tools/probes/wl_ab.sh "<workload> <batch> <lmax>" variant...   -> bench value / ms for each library variant (3 lanes and 1 lane)
W=($1); shift
for v in "$@"; do
  if [ "$v" = "product" ]; then unset GN_LIB_PATH; else export GN_LIB_PATH=gotennet_amd/variants/lib_$v.so; fi
  for lanes in 3 1; do
    timeout 300 python bench.py --workload ${W[0]} --batch ${W[1]} --lmax ${W[2]} --steps 12 --lanes $lanes --no-cpu-baseline --no-split --no-workloads --no-graph --no-forward-only --no-live-traffic --no-static --no-lmax4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v lanes $lanes', '${W[*]}', d['value'], d['ms_per_step'], 'gated', (d.get('roofline_gated_gemm') or {}).get('us_per_launch'))"
  done
done
