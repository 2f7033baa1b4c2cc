export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_generic.py tests/test_gpu_config4_ref.py tests/test_abi_symbols.py -m gpu -x -q 2>&1 | tail -12
