#!/bin/bash
# Print per-kernel VGPR/SGPR/LDS/scratch usage of a HIP shared object (reads the code-object notes).
set -e
tmp=$(mktemp -d); cp "$(readlink -f "$1")" "$tmp/lib.so"; so="$tmp/lib.so"; cd "$tmp"  # the extractor writes next to its input
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading "$so" >/dev/null 2>&1 || true
for f in "$(dirname "$so")"/"$(basename "$so")".*gfx950 ./*gfx950; do
  [ -f "$f" ] || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | awk '
    /\.name:/ {name=$2}
    /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.agpr_count:/ {a=$2}
    /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
    /\.vgpr_spill_count:/ {sp=$2; printf "%-90s vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s spill=%s\n", name, v, a, s, g, p, sp}'
  rm -f "$f"
done
