#!/bin/bash
# A/B helper: bash tools/ab.sh VAR val1 val2 ... [-- extra bench args]
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
  env $var=$v python bench.py --no-lmax4 --no-split --no-graph --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], 'mol/s', d['ms_per_step'], 'ms  gemm', d['roofline'].get('achieved'), 'TF  msg', d['roofline_gather_scatter']['frac'])"
done
