import sys, json; sys.path.insert(0, '.')
import bench
for depth in (1, 2):
    print(json.dumps(bench.amr_one_rank(0, 119, depth=depth)))
print(json.dumps(bench.amr_one_rank(0, 32, depth=2)))
