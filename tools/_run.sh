cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenario.py tests/test_gpu_actor.py -x -q 2>&1 | tail -8
