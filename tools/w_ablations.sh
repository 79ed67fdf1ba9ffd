#!/bin/bash
# Compile-time ablations of the wave-split conv k_spconv_w (timing only, results are wrong): one library per mask
# (-DIMF_W_ABL: 1 no main loop, 2 no neighbour-table loads, 4 no combine / epilogue, 8 launch + dispatch only; bf16x3 loop,
# round 6: 16 no weight loads, 32 no row pieces, 64 no MFMAs, 128 no split, 256 no LDS fragment reads; masks add up).
# usage (here): tools/w_ablations.sh build "0 1 3 5 7 8"   then on the GPU box: [BATCH=2] tools/w_ablations.sh run "..."
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  make -C imfnet_amd/csrc >/dev/null || exit 1
  mkdir -p imfnet_amd/_abl
  OBJS=$(ls build/obj/*.o | grep -v spconv_w.o)
  for m in $2; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Iimfnet_amd/csrc -DIMF_W_ABL=$m -c imfnet_amd/csrc/spconv_w.hip -o /tmp/spconv_w_$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/spconv_w_$m.o -o imfnet_amd/_abl/libw_$m.so -lz ) &
  done; wait
else
  for m in $2; do echo "== mask $m"; IMF_LIB=$PWD/imfnet_amd/_abl/libw_$m.so timeout 300 python ${TOOL:-tools/conv_iso.py} 2>&1 | grep -v amdgpu.ids | tail -16; done
fi
