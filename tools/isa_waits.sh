#!/bin/bash
# Where does a kernel wait for memory?  tools/isa_waits.sh <file.hip> <mangled-name-prefix> [more flags for hipcc]
# Compiles the translation unit to gfx950 assembly (device only) and prints, for the kernel whose symbol starts with the prefix, its
# resource usage and every VMEM instruction, s_waitcnt vmcnt, barrier, scratch access and loop header in program order. Look for
# `s_waitcnt vmcnt(0)` right behind a prefetch: loads in exec-masked branches (cond ? load : 0), register rotations (cur = nxt) and
# vector loads of wave-uniform values all make the compiler wait out the loads it has just issued.
set -e
SRC=$1; PFX=$2; shift 2
D=$(cd "$(dirname "$SRC")" && pwd)
OUT=/tmp/isa_$(basename "$SRC" .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I"$D" -S --cuda-device-only "$@" "$SRC" -o "$OUT" 2>/dev/null
L=$(grep -n "^$PFX" "$OUT" | head -1 | cut -d: -f1)
[ -n "$L" ] || { echo "no kernel starting with $PFX; candidates:"; grep -o "^_Z[A-Za-z0-9_]*" "$OUT" | sort -u | head -12; exit 1; }
awk -v s="$L" 'NR>=s' "$OUT" | awk '{print} /s_endpgm/{exit}' > /tmp/isa_kernel.s
awk -v s="$L" 'NR>=s' "$OUT" | grep -m1 -A40 "\.amdhsa_kernel" | grep "next_free_vgpr\|accum_offset\|private_segment_fixed_size\|group_segment_fixed_size" || true
echo "instructions: $(grep -c '^\s*[vsd]_\|^\s*global_\|^\s*buffer_\|^\s*scratch_\|^\s*flat_' /tmp/isa_kernel.s)  (listing in /tmp/isa_kernel.s)"
grep -n "vmcnt\|global_load\|global_store\|global_atomic\|buffer_load\|buffer_store\|scratch_\|s_barrier\|Loop Header" /tmp/isa_kernel.s | cut -c1-110
