#!/bin/bash
# Builds the stand-alone gfx950 probes next to their sources (run here; the .bin files travel to the GPU box with gpurun).
cd "$(dirname "$0")"
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value "$f" -o "${f%.hip}.bin" || exit 1; done
