#!/bin/bash
# Build libtgis_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")/../text-generation-inference_amd"
mkdir -p lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o lib/libtgis_hip.so csrc/*.hip "$@"
