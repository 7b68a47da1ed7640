#!/bin/bash
# A/B harness: runs tools/quick_bench.py once per library given on the command line (AWSM_HIP_LIB override), prints the stage times.
for lib in "$@"; do
  echo "=== $lib"
  AWSM_HIP_LIB=$lib timeout -k 10 200 python tools/quick_bench.py 3840 2160 30 2>&1 | grep -v "^scene\|^{'name\|^warm" | sed -e "s/'triangles_in.*//" | tail -3
done
