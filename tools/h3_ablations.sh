#!/bin/bash
# Compile-time ablations of the register-staged split-f16 conv k_spconv_h3 (diagnostic builds only since round 3; timing only,
# results are wrong): one library per mask; run with kernel_tag bit 1 / IMF_H3_GLDS=0 so that the launches take k_spconv_h3.
# usage (here): tools/h3_ablations.sh build "0 1 2 4 8 ..."   then on the GPU box: tools/h3_ablations.sh run "..."
cd "$(dirname "$0")/.."
SRCS="$(cd imfnet_amd/csrc && ls core.hip pipeline.hip geometry.hip spconv.hip spconv_pack.hip spconv_g.hip spconv_w.hip head.hip fusion.hip image.hip matching.hip keypoints.hip ransac.hip executor.hip codecs.hip backward.hip | sed 's#^#imfnet_amd/csrc/#') tools/diagnostic/spconv_h3.hip -Iimfnet_amd/csrc"
if [ "$1" = build ]; then
  mkdir -p imfnet_amd/_abl
  for m in $2; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DIMF_WITH_H3 -DIMF_H3_ABL=$m $SRCS -o imfnet_amd/_abl/lib_$m.so -lz & done; wait
else
  for m in $2; do echo -n "mask $m: "; IMF_LIB=$PWD/imfnet_amd/_abl/lib_$m.so timeout 200 python tools/layer_times.py 2>&1 | grep -E " us$|sum of" | awk '{printf "%s ", $NF=="us"?$(NF-1):$0} END {print ""}'; done
fi
