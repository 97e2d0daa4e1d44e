timeout 600 python -m pytest tests -m gpu -x -q -W ignore -k "pits" 2>&1 | grep passed
for k in 1 2; do python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['stages_ms']['pits_ms'],2))"; done
