#!/bin/bash
# Register / spill / LDS summary of the pair kernels (cross-compiles on CPU).  Usage: bash tests/isa_stats.sh [extra hipcc flags]
cd "$(dirname "$0")/../arpeggia_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../include "$@" -S --cuda-device-only kernels.hip -o /tmp/arp_kernels.s 2>/dev/null || exit 1
awk '/^    .name: / {n=$2} /vgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size|sgpr_count/ {v[n]=v[n]" "$1" "$2} END {for (k in v) if (k ~ /k_pairs/) print k, v[k]}' /tmp/arp_kernels.s | sed 's/_ZN3arp//; s/EvNS_8DevAtoms.*Py / /; s/ENS_8DevAtoms[^ ]* / /'
