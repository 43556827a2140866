#!/bin/bash
# GPU box: everything DESIGN.md section 9 quotes about k_radius_cells on the S30k 60 000 x 43 table, in one text file:
# per-phase shader cycles at 1 and 6 workgroups per CU (PCRCG_DEBUG=radius_prof=1), instruction counts per class, SQ
# wave-cycle shares, L1->L2 / TLB counters for the cell kernel and the per-query kernel, and the load-latency microbenchmark.
# usage: scripts/radius_counters.sh TAG  -> gpurun_out/TAG_radius_counters.txt
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_radius_counters.txt
export RADIUS_BENCH_ONLY=conv0
{
  echo "# k_radius_cells on the S30k conv0 table (60 000 queries x 43 columns over 60 000 supports), MI355X"
  for B in 256 1536; do
    echo; echo "## per-phase shader cycles, radius_blocks=$B (PCRCG_DEBUG=radius_prof=1; sums over wavefronts)"
    PCRCG_DEBUG=radius_prof=1,radius_blocks=$B python3 $R/scripts/radius_bench.py S30k --mode new --reps 5 2>&1 | grep -E "k_radius_cells"
  done
  echo; echo "## instructions per class (rocprofv3 --pmc SQ_INSTS_*; the 8 timed conv0 launches)"
  $R/scripts/pmc_radius_insts.sh 2>&1 | grep SQ_INSTS
  echo; echo "## SQ wave-cycle shares (rocprofv3 --pmc, scripts/pmc_cmd.sh; averages over the run's launches of each kernel)"
  $R/scripts/pmc_cmd.sh ${TAG}_radius_conv0 $R/scripts/radius_bench.py S30k --mode new --reps 5 2>&1 | grep -E "^kernel|k_radius|^act/"
  echo; echo "## L1 -> L2 requests, their summed latency, TLB, L2 hits / misses, fabric requests: cell kernel (new) and per-query kernel (old)"
  $R/scripts/pmc_radius_mem.sh 2>&1 | grep -E "^(new|old) "
  echo; echo "## scripts/micro/load_latency: one dependent load per iteration, cycles per iteration vs resident workgroups"
  $R/scripts/micro/load_latency 2>&1 | grep -E "working|27 lanes|vector load \+ barrier, 24"
} > $O 2>&1
rm -f $R/gpurun_out/${TAG}_radius_conv0_sq_pass*.err
cat $O
