timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tile', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
