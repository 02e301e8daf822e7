#!/bin/bash
# Device ISA of one kernel: tools/isa.sh <mangled-name-prefix> [out.s]   (writes /tmp/eaz.s and the kernel's slice)
set -e
cd /root/repo/e-alphazero_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -S --cuda-device-only -o /tmp/eaz.s eaz_api.hip ${ISA_FLAGS}
s=$(grep -n "^$1" /tmp/eaz.s | head -1 | cut -d: -f1)
e=$(grep -n "amdhsa_kernel $1" /tmp/eaz.s | head -1 | cut -d: -f1)
sed -n "${s},${e}p" /tmp/eaz.s > ${2:-/tmp/kernel.s}
wc -l ${2:-/tmp/kernel.s}
grep -A30 "amdhsa_kernel $1" /tmp/eaz.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed" 
