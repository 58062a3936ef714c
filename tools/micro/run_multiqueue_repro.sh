#!/bin/bash
# Build and run the standalone multi-queue reproducer against the system ROCm (7.2) runtime — no torch in the process.
set -e
cd "$(dirname "$0")/../.."
hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/micro/multiqueue_repro.cpp -o tools/micro/multiqueue_repro \
  -Lhunyuanworld-mirror_amd -lwm_hip -Wl,-rpath,"$PWD/hunyuanworld-mirror_amd" -Wl,-rpath,/opt/rocm/lib
LD_LIBRARY_PATH=/opt/rocm/lib ./tools/micro/multiqueue_repro "${1:-12}" "${2:-3}"
