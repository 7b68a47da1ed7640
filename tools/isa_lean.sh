#!/bin/bash
# Static ISA report of one kernel of kernels_shade.hip (no GPU needed).  Usage: tools/isa_lean.sh [kernel_substring] [extra hipcc flags]
K=${1:-k_shade_leanILb0ELi0ELb0}; shift
mkdir -p /tmp/isa && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function --cuda-device-only -S "$@" -o /tmp/isa/shade2.s kernels_shade.hip 2>&1 | grep -v "hip-link"
python /root/repo/tools/isa_cost.py /tmp/isa/shade2.s $K | head -4
L=$(grep -n "^_ZN4awsm[0-9]*$K" /tmp/isa/shade2.s | head -1 | cut -d: -f1)
awk -v s=$L 'NR>=s' /tmp/isa/shade2.s | awk '/s_endpgm/{print; exit} {print}' > /tmp/isa/kernel.s
awk -v s=$L 'NR>=s' /tmp/isa/shade2.s | grep -m4 "NumVgprs\|ScratchSize\|Occupancy\|LDSByteSize"
