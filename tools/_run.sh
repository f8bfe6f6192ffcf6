cd /root/repo
timeout 600 python -m pytest tests/test_gpu_scenario.py -x -q 2>&1 | tail -15
