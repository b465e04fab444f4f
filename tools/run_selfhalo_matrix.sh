#!/bin/bash
# full GPU suite + self-halo bench (depth 1 / depth 2, plain / graph)
o=gpurun_out/${1:-r5c}; mkdir -p $o
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $o/pytest.log; tail -5 $o/pytest.log
common="--self-halo --steps 50 --warmup 5 --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 --no-real-mesh --develop 0"
for v in "d1_plain --halo-depth 1" "d1_graph --halo-depth 1 --graph" "d2_plain --halo-depth 2" "d2_graph --halo-depth 2 --graph"; do
  set -- $v; tag=$1; shift
  timeout -k 10 300 python bench.py $common "$@" > $o/selfhalo_$tag.json 2> $o/selfhalo_$tag.err || echo "bench $tag failed"
  python3 -c "import json,sys; d=json.loads(open('$o/selfhalo_$tag.json').read().strip().splitlines()[-1]); r=d['per_rank'][0]; print('$tag', round(d['ms_per_step'],4), round(d['value']), r['exchanges_per_step'], round(r['halo_ms_per_step'],4), r['step_graph'])"
done
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 --no-real-mesh --develop 0 > $o/nohalo.json 2> $o/nohalo.err
python3 -c "import json,sys; d=json.loads(open('$o/nohalo.json').read().strip().splitlines()[-1]); print('nohalo', round(d['ms_per_step'],4), round(d['value']))"
