export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_generic.py -m gpu -x -q -k "forms_agree" 2>&1 | tail -15
