export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6
