#!/bin/bash
# Build csrc/ as of a git revision into tools/lib/libwavtok_hip_prev.so (same-box A/B: tools/ab_lib.sh,
# WAVTOK_HIP_LIB=...).  Usage: tools/build_prev.sh [rev]   (default HEAD)
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/wt_prev.XXXX)
git -C "$root" archive "$rev" wavtokenizer_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/wavtokenizer_amd/csrc" -j4 > "$tmp/build.log" 2>&1 || { tail -20 "$tmp/build.log"; exit 1; }
mkdir -p "$root/tools/lib"; cp "$tmp/wavtokenizer_amd/libwavtok_hip.so" "$root/tools/lib/libwavtok_hip_prev.so"
rm -rf "$tmp"
echo "built $rev -> tools/lib/libwavtok_hip_prev.so"
