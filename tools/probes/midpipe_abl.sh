for v in "" a1 a2 a3 a4 a8 a24 a32 a63; do
  if [ -z "$v" ]; then timeout 200 python tools/midpipe_ab.py 2>&1 | grep -E "54368x256x256 (gate|plain)\]" ; else GN_LIB_PATH=gotennet_amd/variants/lib_$v.so timeout 200 python tools/midpipe_ab.py 2>&1 | grep -E "54368x256x256 (gate|plain)\]"; fi
done
