#!/bin/bash
# GPU box: A/B of whole library builds (scripts/build_variant.sh -> ab/NAME.so) on the TRAIN step (scripts/bench_train.py),
# interleaved so that box drift shows.  "cur" = the in-tree build.
# usage: scripts/ab_train.sh [-s STEPS] [-r ROUNDS] cur base ...
STEPS=30; ROUNDS=2
while [ "${1#-}" != "$1" ]; do case $1 in -s) STEPS=$2;; -r) ROUNDS=$2;; esac; shift 2; done
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/pcrcg_amd/libpcrcg_hip.so /tmp/cur.so
# whatever happens below (a failed run, an interrupt), the in-tree library is put back: later tests and benchmarks must not
# run against a variant build (pcrcg_amd/_lib.py loads the in-tree file and nothing else)
trap 'cp /tmp/cur.so $R/pcrcg_amd/libpcrcg_hip.so' EXIT
for round in $(seq $ROUNDS); do
  for name in "$@"; do
    if [ "$name" = "cur" ]; then cp /tmp/cur.so $R/pcrcg_amd/libpcrcg_hip.so; else cp $R/ab/$name.so $R/pcrcg_amd/libpcrcg_hip.so; fi
    v=$(timeout 300 python $R/scripts/bench_train.py --steps $STEPS --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$name] $v ms/step"
  done
done
