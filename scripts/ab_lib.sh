#!/bin/bash
# GPU box: in-engine A/B of whole library builds (scripts/build_variant.sh -> ab/NAME.so), interleaved so that box drift
# shows: for each round, for each library, one bench.py run.  "cur" = the in-tree build.
# usage: scripts/ab_lib.sh [-s STEPS] [-r ROUNDS] [-a "bench args"] cur base_gemm ...
STEPS=480; ROUNDS=2; ARGS=""
while [ "${1#-}" != "$1" ]; do case $1 in -s) STEPS=$2;; -r) ROUNDS=$2;; -a) ARGS=$2;; esac; shift 2; done
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/pcrcg_amd/libpcrcg_hip.so /tmp/cur.so
# whatever happens below (a failed run, an interrupt), the in-tree library is put back: later tests and benchmarks must not
# run against a variant build (pcrcg_amd/_lib.py loads the in-tree file and nothing else)
trap 'cp /tmp/cur.so $R/pcrcg_amd/libpcrcg_hip.so' EXIT
for round in $(seq $ROUNDS); do
  for name in "$@"; do
    if [ "$name" = "cur" ]; then cp /tmp/cur.so $R/pcrcg_amd/libpcrcg_hip.so; else cp $R/ab/$name.so $R/pcrcg_amd/libpcrcg_hip.so; fi
    v=$(python $R/bench.py $ARGS --no-cpu-baseline --no-extras --steps $STEPS --repeats 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'], d['config'].get('lib_sha16'))")
    echo "[$name] $v"
  done
done
