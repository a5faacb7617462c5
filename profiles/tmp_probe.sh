export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
