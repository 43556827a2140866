#!/bin/bash
# Build an A/B variant of libpcrcg_hip.so: the CURRENT csrc/ with some files taken from a git revision (or from a path).
#   scripts/build_variant.sh base_gemm gemm_x6.hip=HEAD            -> ab/base_gemm.so
#   scripts/build_variant.sh try1 gemm_x6.hip=/tmp/try1/gemm_x6.hip EXTRA=-DFOO=1
# The variants travel with gpurun (ab/ is git-ignored, not gpurun-ignored); scripts/ab_bench.sh swaps them in on the GPU box.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
W=/tmp/pcrcg_variant_$NAME
rm -rf $W; mkdir -p $W/pcrcg_amd $R/ab
cp -r $R/pcrcg_amd/csrc $W/pcrcg_amd/csrc; rm -rf $W/pcrcg_amd/csrc/build
cp -r $R/include $W/include
EXTRA=""
for kv in "$@"; do
  k=${kv%%=*}; v=${kv#*=}
  if [ "$k" = "EXTRA" ]; then EXTRA="$v"; continue; fi
  if [ -f "$v" ]; then cp "$v" $W/pcrcg_amd/csrc/$k; else (cd $R && git show "$v:pcrcg_amd/csrc/$k") > $W/pcrcg_amd/csrc/$k; fi
done
make -s -j8 -C $W/pcrcg_amd/csrc EXTRA="$EXTRA"
cp $W/pcrcg_amd/libpcrcg_hip.so $R/ab/$NAME.so
echo "ab/$NAME.so $(sha256sum $R/ab/$NAME.so | cut -c1-16)"
