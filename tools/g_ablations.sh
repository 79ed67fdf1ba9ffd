#!/bin/bash
# Compile-time ablations of the LDS-DMA conv k_spconv_g (timing only, results are wrong): one library per mask
# (-DIMF_G_ABL: 1 no MFMAs, 2 no hi/lo split, 4 no row gathers, 8 no weight copies).
# usage (here): tools/g_ablations.sh build "0 1 2 4 8 12 15"   then on the GPU box: [BATCH=2] tools/g_ablations.sh run "..."
cd "$(dirname "$0")/.."
SRCS="$(cd imfnet_amd/csrc && ls core.hip pipeline.hip geometry.hip spconv.hip spconv_pack.hip spconv_g.hip spconv_w.hip head.hip fusion.hip image.hip matching.hip keypoints.hip ransac.hip executor.hip codecs.hip backward.hip | sed 's#^#imfnet_amd/csrc/#') tools/diagnostic/spconv_h3.hip -Iimfnet_amd/csrc"
if [ "$1" = build ]; then
  mkdir -p imfnet_amd/_abl
  for m in $2; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DIMF_G_ABL=$m $SRCS -o imfnet_amd/_abl/libg_$m.so -lz & done; wait
else
  for m in $2; do echo -n "mask $m: "; IMF_LIB=$PWD/imfnet_amd/_abl/libg_$m.so timeout 200 python ${TOOL:-tools/layer_times.py} 2>&1 | grep -E " us$|sum of" | awk '{printf "%s ", $NF=="us"?$(NF-1):$0} END {print ""}'; done
fi
