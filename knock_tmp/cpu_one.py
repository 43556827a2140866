import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcrcg_amd import indoor_config, synthetic
bench.RECIPE = "S30k"
cfg = indoor_config(); limits = synthetic.LIMITS["S30k"]
print(bench._cpu_front_end(("S30k", 100, dict(cfg), limits)))
print(bench._cpu_front_end(("S30k", 101, dict(cfg), limits)))
