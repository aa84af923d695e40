#!/bin/bash
# Build csrc/ as of a git revision into wavtokenizer_amd/libwavtok_hip_prev.so (same-box A/B: tools/ab_lib.sh,
# WAVTOK_HIP_LIB=...).  Usage: tools/build_prev.sh [rev]   (default HEAD)
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/wt_prev.XXXX)
git -C "$root" archive "$rev" wavtokenizer_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/wavtokenizer_amd/csrc" -j4 > "$tmp/build.log" 2>&1 || { tail -20 "$tmp/build.log"; exit 1; }
cp "$tmp/wavtokenizer_amd/libwavtok_hip.so" "$root/wavtokenizer_amd/libwavtok_hip_prev.so"
rm -rf "$tmp"
echo "built $rev -> wavtokenizer_amd/libwavtok_hip_prev.so"
