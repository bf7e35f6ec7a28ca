timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
LT_TAIL_TRACE=1 timeout 120 python tools/profile_e2e_batched.py 2>&1 | grep -E "^rep[123]|tracks\+agg|host unpack|union|init_common"
