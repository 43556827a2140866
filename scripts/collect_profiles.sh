#!/bin/bash
# GPU box: bench line + rocprofv3 summaries for profiles/ (run through gpurun; writes under gpurun_out/$1_*).
# usage: scripts/collect_profiles.sh r02_v1
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py 2> $O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench.json                       # (the default: 480-step regions)
python3 $R/bench.py --steps 48 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_48_step_regions.json
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_driver_setting_20_steps.json     # the driver's command, extras included
db() { find "$1" -name "*results.db" | head -1; }
# 1. the timed bench under the kernel tracer
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python3 $R/bench.py --steps 48 --warmup 5 --repeats 1 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_under_rocprof.json
# pairs under the tracer: 24 priming (4 x model streams x pairs per forward) + 5 warm-up + 50 timed + 3 isolated forwards (+ 3 isolated pyramid builds)
python3 $R/scripts/prof_summary.py $(db /tmp/p1) $O/${TAG}_kernel_stats.csv 82
python3 $R/scripts/front_chain.py $(db /tmp/p1) > $O/${TAG}_front_chain.txt 2>&1
python3 $R/scripts/concurrency.py $(db /tmp/p1) > $O/${TAG}_concurrency.txt 2>&1
# 2. the isolated forwards (one stream, nothing else running)
rm -rf /tmp/p2; rocprofv3 --kernel-trace --stats -d /tmp/p2 -o p -- python3 $R/bench.py --isolated-only --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_isolated.json
python3 $R/scripts/prof_summary.py $(db /tmp/p2) $O/${TAG}_isolated_kernel_stats.csv 24
# 3. PMC passes (separate runs, counters only + kernel trace)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p3; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p3 -o p -- python3 $R/bench.py --isolated-only --steps 3 --warmup 1 > /dev/null 2>&1
  cp $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_$C.csv
done
python3 $R/scripts/pmc_summary.py $O/${TAG}_pmc_FETCH_SIZE.csv $O/${TAG}_pmc_WRITE_SIZE.csv $O/${TAG}_pmc_traffic.json
rm -f $O/${TAG}_pmc_FETCH_SIZE.csv $O/${TAG}_pmc_WRITE_SIZE.csv
ls -la $O | grep ${TAG}
# 4. secondary lines (no tracer): other workloads, index tie order, bf16 feature-storage variant
if [ "${2:-}" = "all" ]; then
  for W in U30k K120k T30k; do
    S=480; [ $W = K120k ] && S=120        # (regions of whole groups: four pairs, K120k three)
    python3 $R/bench.py --workload $W --steps $S --warmup 5 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_$W.json
  done
  PCRCG_TIE_ORDER=index python3 $R/bench.py --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_tie_index.json
  python3 $R/bench.py --variant bf16 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_bf16_variant.json
  python3 $R/bench.py --pairs-per-forward 1 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_one_pair_per_forward.json
fi
ls -la $O | grep ${TAG}
# 5. the other workloads under the kernel tracer (kernel breakdown, front chain, concurrency) -- with the SAME library
if [ "${2:-}" = "all" ]; then
  cd /tmp
  for W in K120k T30k; do
    rm -rf /tmp/pw; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pw -o p -- python3 $R/bench.py --workload $W --steps $([ $W = K120k ] && echo 30 || echo 48) --warmup 5 --repeats 1 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_${W}_under_rocprof.json
    python3 $R/scripts/prof_summary.py $(db /tmp/pw) $O/${TAG}_${W}_kernel_stats.csv 62
    python3 $R/scripts/front_chain.py $(db /tmp/pw) > $O/${TAG}_${W}_front_chain.txt 2>&1
    python3 $R/scripts/concurrency.py $(db /tmp/pw) > $O/${TAG}_${W}_concurrency.txt 2>&1
  done
  bash $R/scripts/collect_train_profile.sh $TAG > /dev/null 2>&1
  bash $R/scripts/collect_forward_sequence.sh
  mv $O/r05_forward_sequence.txt $O/${TAG}_forward_sequence.txt 2>/dev/null
fi
ls -la $O | grep ${TAG}
