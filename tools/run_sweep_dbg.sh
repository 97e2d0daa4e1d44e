PYDEM_PITS_DEBUG=2 python bench.py --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep "^pits"
