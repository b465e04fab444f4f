#!/bin/bash
# halo tests + self-halo bench matrix after: folded packs for two layers, ghost limiter inside k_upd_superbee, lazy ghost carry
o=gpurun_out/${1:-r5d}; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_two_layers.py tests/test_gpu_multirank.py tests/test_gpu_partition.py tests/test_gpu_amr.py tests/test_gpu_cpp_adapter.py -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $o/pytest.log; tail -5 $o/pytest.log
common="--self-halo --steps 50 --warmup 5 --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 --no-real-mesh --develop 0"
for v in "d1_plain --halo-depth 1" "d1_graph --halo-depth 1 --graph" "d2_plain --halo-depth 2" "d2_graph --halo-depth 2 --graph"; do
  set -- $v; tag=$1; shift
  timeout -k 10 300 python bench.py $common "$@" > $o/selfhalo_$tag.json 2> $o/selfhalo_$tag.err || echo "bench $tag failed"
  python3 -c "import json,sys; d=json.loads(open('$o/selfhalo_$tag.json').read().strip().splitlines()[-1]); r=d['per_rank'][0]; print('$tag', round(d['ms_per_step'],4), round(d['value']), r['exchanges_per_step'], round(r['halo_ms_per_step'],4), r['step_graph']['state'], r['halo_plan'])"
done
