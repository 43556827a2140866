"""Host-side cProfile of scripts/bench_train.py (where the Python time of a train step goes; GPU box only)."""
import cProfile, pstats, sys, os, io
sys.argv = ["bench_train.py", "--steps", "6", "--warmup", "2"]
sys.path.insert(0, os.getcwd())
pr = cProfile.Profile()
pr.enable()
exec(open("scripts/bench_train.py").read())
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
