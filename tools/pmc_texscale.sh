#!/bin/bash
# The lean kernel's L2 / fabric counters against the texture footprint (tools/quick_bench.py's tex_scale).  Usage: tools/pmc_texscale.sh <outdir> scale ...
set -o pipefail
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
for ts in "$@"; do
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d $OUT/ts$ts -- python3 tools/quick_bench.py 3840 2160 3 0 $ts > $OUT/ts$ts.log 2>&1
  rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/tcp$ts -- python3 tools/quick_bench.py 3840 2160 3 0 $ts > $OUT/tcp$ts.log 2>&1
  echo "== tex_scale $ts"
  python3 tools/pmc_summary.py $OUT/ts$ts $OUT/tcp$ts | grep -A 7 "k_shade_lean"
done
