#!/bin/bash
# Timing-only variants of attn_v4's tile loop (attention_v4.hip WM_V4_DIAG: 1 no barrier, 2 no LDS fragment reads, 4 no DMA, 8 no
# vmcnt wait), each as a stamps build of its own: hunyuanworld-mirror_amd/libwm_hip_diag<N>.so.  Their outputs are wrong by
# construction; only tools/attn_stamps.py's cycles per MFMA are read off them.
#   build (here):  [DIAG_EXTRA=-DWM_V4_STAGED=0] bash tools/attn_v4_diag.sh build 1 2 4 ...
#   run (GPU box): bash tools/attn_v4_diag.sh run 1 2 4 ...
set -e
cd "$(dirname "$0")/../hunyuanworld-mirror_amd/csrc"
mode=$1; shift
if [ "$mode" = build ]; then
  make stamps > /dev/null
  flags=$(grep '^CXXFLAGS' Makefile | cut -d= -f2- | sed 's/\$(ARCH)/gfx950/; s/\$(EXTRA)//')
  for n in "$@"; do
    hipcc $flags $DIAG_EXTRA -DWM_ATTN_STAMPS -DWM_V4_DIAG=$n -c attention_v4.hip -o build_stamps/attention_v4_diag$n.o
    objs=$(ls build_stamps/*.o | grep -v 'attention_v4' | tr '\n' ' ')
    hipcc -shared -fPIC --offload-arch=gfx950 -o ../libwm_hip_diag$n.so $objs build_stamps/attention_v4_diag$n.o -L/opt/rocm/lib -lrccl -lpthread -Wl,-rpath,/opt/rocm/lib
    echo "built libwm_hip_diag$n.so"
  done
else
  cd ../..
  for n in "$@"; do
    lib=hunyuanworld-mirror_amd/libwm_hip_diag$n.so
    echo "# WM_V4_DIAG=$n"
    STAMP_QB=8 WM_HIP_LIB=$lib timeout -k 10 120 python tools/attn_stamps.py bf16 32
  done
fi
