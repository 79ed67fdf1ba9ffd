#!/bin/bash
# Round 6 A/B on ONE box: k_spconv_w builds under imfnet_amd/_abl/ (libw_<name>.so), isolated launches of the pair's shapes
# (tools/conv_iso.py) and the pair step (tools/step_pair.py).   LIBS="base new" MODES="wave4h wave4 wave4u wave8 wave8u" tools/r06_ab.sh
for m in ${LIBS:-base new base new}; do
  echo "== lib $m"; IMF_LIB=$PWD/imfnet_amd/_abl/libw_$m.so BATCH=2 VARIANT=3 timeout 300 python tools/conv_iso.py ${MODES:-wave4h wave4 wave4u wave8 wave8u} 2>&1 | grep -v amdgpu.ids | grep -E "block2_tr|block2 |block3|block4 |conv2 |sum" | cut -c1-170
  [ -n "$NOSTEP" ] || for tag in ${L0TAGS:-72}; do echo "-- IMF_L0_TAG=$tag"; IMF_L0_TAG=$tag IMF_LIB=$PWD/imfnet_amd/_abl/libw_$m.so timeout 300 python tools/step_pair.py 2>&1 | grep -E "ms/step|spconv_w" ; done
done
