#!/bin/bash
# gpurun wrapper: stamps the tree's commit (+ "-dirty") into .fami_sha so records written on the GPU box (no .git there) can
# say which tree they describe.   usage: tools/gpu.sh <timeout s> '<command>'
cd "$(dirname "$0")/.."
sha=$(git rev-parse --short HEAD 2>/dev/null || echo unknown)
git diff --quiet 2>/dev/null || sha="$sha-dirty"
echo "$sha" > .fami_sha
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
