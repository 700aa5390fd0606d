cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q > gpurun_out/ab14_tests.log 2>&1; tail -2 gpurun_out/ab14_tests.log
B="python bench.py --no-lmax4 --no-split --no-graph --no-workloads --no-cpu-baseline --no-forward-only --no-live-traffic --no-static --steps 16 --warmup 4"
for r in 1 2; do
for v in new k6w2; do
  if [ $v = new ]; then unset GN_LIB_PATH; else export GN_LIB_PATH=gotennet_amd/variants/lib_$v.so; fi
  $B --lanes 1 > gpurun_out/ab14_${v}_c2_$r.json 2> /dev/null
  $B --lanes 3 > gpurun_out/ab14_${v}_c2x3_$r.json 2> /dev/null
  $B --lanes 1 --workload md22_ac_ala3 --batch 64 > gpurun_out/ab14_${v}_c3_$r.json 2> /dev/null
  $B --lanes 1 --lmax 1 > gpurun_out/ab14_${v}_l1_$r.json 2> /dev/null
  python - <<PY
import json
out=[]
for w in ("c2","c2x3","c3","l1"):
    a=json.load(open("gpurun_out/ab14_${v}_%s_$r.json"%w))
    out.append("%s %.3f ms gs %.1f us"%(w,a["ms_per_step"],a["roofline_gather_scatter"]["us_per_launch"]))
print("$v $r:", " | ".join(out))
PY
done; done
