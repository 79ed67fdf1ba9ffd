#!/bin/bash
# Round 6: timing-only ablations of k_spconv_w's bf16x3 main loop (wrong results): what does a sub-stage wait for?
# build here:  tools/w_ablations.sh build "0 16 32 48 64 128 192 240 256"      then on the GPU box:  tools/r06_ab.sh
# masks: 16 no weight loads, 32 no row pieces, 48 neither, 64 no MFMAs, 128 no split, 192 neither, 240 none of the four, 256 no LDS fragment reads
for m in ${MASKS:-0 16 32 48 64 128 192 240 256 0}; do
  echo "== mask $m"; IMF_LIB=$PWD/imfnet_amd/_abl/libw_$m.so BATCH=2 VARIANT=3 timeout 300 python tools/conv_iso.py wave4h wave4 wave8 2>&1 | grep -v amdgpu.ids | grep -E "block2_tr|block2 |block3|block4 |conv2 |sum" | cut -c1-150
done
