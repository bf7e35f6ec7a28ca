timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/_dbg_replay.py 2>&1 | grep -v amdgpu.ids | tail -6
