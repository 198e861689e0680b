#!/bin/bash
# Register / spill / LDS summary of the kernels (cross-compiles on CPU).  Usage: bash tests/isa_stats.sh [pattern] [extra hipcc flags]
# (reads the .amdhsa_kernel blocks of the assembly: one block per kernel, unlike the metadata's .name lines, which also name arguments)
PAT=${1:-k_emit}; shift
cd "$(dirname "$0")/../arpeggia_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -I../../include "$@" -S --cuda-device-only kernels.hip -o /tmp/arp_kernels.s 2>/dev/null || exit 1
awk -v pat="$PAT" '$1 == ".amdhsa_kernel" {n=$2; v=""} $1 ~ /^\.amdhsa_(group_segment_fixed_size|next_free_vgpr|next_free_sgpr|private_segment_fixed_size|accum_offset)$/ {k=$1; sub(/\.amdhsa_/,"",k); v=v" "k"="$2} $1 == ".end_amdhsa_kernel" {if (n ~ pat) print n, v}' /tmp/arp_kernels.s | c++filt | sed 's/(arp::DevAtoms[^)]*)//' | sort
