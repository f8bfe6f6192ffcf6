cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or variants" 2>&1 | tail -5
