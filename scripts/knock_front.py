"""Knock-out accounting of the FRONT END (GPU box): bench.py's engine with every pyramid builder answering from a
cache after its first call per shape (results wrong -- every pair gets the tables of an earlier one -- timing meaningful):
what the three model streams sustain when the front-end stream does no work.  Complements scripts/knockout.py."""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pcrcg_amd import pyramid  # noqa: E402

_real = pyramid.NativePyramid.build


def cached_build(self, points, lengths, fresh_arena=False, defer_restore=False, group=0):
    if isinstance(points, (list, tuple)):            # a grouped build hands its pairs over as parts
        key = (sum(int(p.shape[0]) for p in points) // 1000, sum(int(l.shape[0]) for l in lengths), bool(defer_restore), int(group))
    else:
        key = (int(points.shape[0]) // 1000, int(lengths.shape[0]), bool(defer_restore), int(group))
    cache = self.__dict__.setdefault("_knock_cache", {})
    if key not in cache:
        cache[key] = _real(self, points, lengths, fresh_arena, defer_restore, group)
    return cache[key]


if os.environ.get("PCRCG_KNOCK_FRONT", "1") == "1":
    pyramid.NativePyramid.build = cached_build
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "100", "--repeats", "3"] + sys.argv[1:]
runpy.run_path(os.path.join(REPO, "bench.py"), run_name="__main__")
