python bench.py --breakdown --no-cpu-baseline --steps 10 2>&1 >/dev/null | grep -E "message_aggregate|message_backward"
python bench.py --lmax 4 --breakdown --no-cpu-baseline --steps 5 2>&1 >/dev/null | grep -E "message_aggregate"
