#!/bin/bash
set -e
cd "$(dirname "$0")/../.."
hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/micro/splat_hazard_repro.cpp -o tools/micro/splat_hazard_repro \
  -Lhunyuanworld-mirror_amd -lwm_hip -Wl,-rpath,"$PWD/hunyuanworld-mirror_amd" -Wl,-rpath,/opt/rocm/lib
LD_LIBRARY_PATH=/opt/rocm/lib ./tools/micro/splat_hazard_repro "${1:-30}" "${2:-5}"
