#!/bin/bash
# Round 6 A/B on ONE box: libw_0 (the tree's k_spconv_w) against libw_512 (-DIMF_W_ABL=512: single gathered-row buffer, the
# loop of the round's first commits).  build here: tools/w_ablations.sh build "0 512"   then on the GPU box: tools/r06_ab.sh
for m in ${MASKS:-512 0 512 0}; do
  echo "== mask $m"; IMF_LIB=$PWD/imfnet_amd/_abl/libw_$m.so BATCH=2 VARIANT=3 timeout 300 python tools/conv_iso.py wave4h wave4 wave8 wave8u 2>&1 | grep -v amdgpu.ids | grep -E "block2_tr|block2 |block3|block4 |conv2 |sum" | cut -c1-150
  IMF_LIB=$PWD/imfnet_amd/_abl/libw_$m.so timeout 300 python tools/step_pair.py 2>&1 | grep -v amdgpu.ids | tail -3
done
