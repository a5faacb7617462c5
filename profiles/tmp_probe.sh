export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8
