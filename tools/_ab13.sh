cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-lmax4 --no-split --no-graph --no-workloads --no-cpu-baseline --no-forward-only --no-live-traffic --no-static --steps 16 --warmup 4 --lanes 1"
for r in 1 2; do
for v in new k6w0 k6w3 k6w4 htrt2 htrt3 htrs3; do
  if [ $v = new ]; then unset GN_LIB_PATH; else export GN_LIB_PATH=gotennet_amd/variants/lib_$v.so; fi
  $B > gpurun_out/ab13_${v}_$r.json 2> /dev/null
  python - <<PY
import json
a=json.load(open("gpurun_out/ab13_${v}_$r.json"))
print("$v $r: %.3f ms gs %.1f us (general %.1f) htr_bwd %.1f us"%(a["ms_per_step"],a["roofline_gather_scatter"]["us_per_launch"],a["roofline_gather_scatter"]["general_launches"]["us"],a["roofline_htr_backward"]["us_per_launch"]))
PY
done; done
