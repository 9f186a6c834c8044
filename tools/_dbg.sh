python -m pytest tests -x -q -m gpu 2>&1 | tail -3
SF_COMPARE_DENSE=1 python tools/bench_banded.py 4096 128 20 2>&1 | tail -2
