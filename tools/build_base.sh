#!/bin/bash
# Builds the committed state of the engine (HEAD) as molecular-dynamics_amd/libmdx_base.so beside the working tree's
# libmdx.so, for A/B timings on one GPU box (MDX_LIB selects the library). Usage: tools/build_base.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive HEAD molecular-dynamics_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/molecular-dynamics_amd/csrc" -j8 ../libmdx.so > /dev/null 2>&1
cp "$tmp/molecular-dynamics_amd/libmdx.so" "$root/molecular-dynamics_amd/libmdx_base.so"
rm -rf "$tmp"
echo built molecular-dynamics_amd/libmdx_base.so from $(git -C "$root" rev-parse --short HEAD)
