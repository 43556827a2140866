import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
if __name__ == "__main__":
    import bench
    from pcrcg_amd import indoor_config, synthetic
    bench.RECIPE = "S30k"
    cfg = indoor_config(); limits = synthetic.LIMITS["S30k"]
    t = time.time()
    with mp.get_context("spawn").Pool(4) as pool:
        print(len(set(pool.map(bench._cpu_warm, range(16), chunksize=1))), time.time() - t)
        t = time.time()
        r = pool.map(bench._cpu_front_end, [("S30k", 100 + i, dict(cfg), limits) for i in range(4)], chunksize=1)
        print(r, 4 / (time.time() - t))
