timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 120 python tools/profile_e2e_batched.py 2>&1 | grep -E "^rep[123]"
