#!/bin/bash
# register / spill figures of ONE source's kernels without linking the library:  tools/spills_of.sh decoder.hip [grep-pattern] [extra hipcc flags...]
# (device-only compile into /tmp, metadata read with llvm-readelf; prints vgpr sgpr | vgpr-spill sgpr-spill scratch-bytes | kernel)
src=$1; pat=${2:-.}; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/spills_$$.co
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on --cuda-device-only --no-gpu-bundle-output -c "$@" "$root/tinyvc_amd/csrc/$src" -o $out 2>/dev/null || { echo "compile failed"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on --cuda-device-only --no-gpu-bundle-output -c "$@" "$root/tinyvc_amd/csrc/$src" -o $out 2>&1 | grep -E "error" | head; exit 1; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out | awk '
/^    \.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2} /\.sgpr_count:/ {sg=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {vg=$2}
/\.vgpr_spill_count:/ {printf "%4d %4d | %4d %4d %5d | %s\n", vg, sg, $2, ss, scr, name}' | c++filt | sed 's/tvc::(anonymous namespace):://g; s/tvc:://g' | grep -E "$pat" | sort -t'|' -k2 -r | cut -c1-240
rm -f $out
