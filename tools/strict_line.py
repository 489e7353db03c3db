import sys, json
sys.path.insert(0, '/root/repo')
import bench
import graphgan_amd as ga
from graphgan_amd import _lib
print(json.dumps(bench.strict_mode_line(ga, _lib)))
