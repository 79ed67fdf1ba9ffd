#!/bin/bash
# Compile-time ablations of k_insert_points (timing only, results are wrong): one library per mask (IMF_GEO_ABL in csrc/geometry.hip).
# usage (here): tools/geo_ablations.sh build "0 1 2 4 8 15"   then on the GPU box: tools/geo_ablations.sh run "0 1 2 4 8 15"
cd "$(dirname "$0")/.."
SRCS=$(cd imfnet_amd/csrc && ls core.hip geometry.hip spconv.hip spconv_pack.hip spconv_g.hip spconv_w.hip head.hip fusion.hip image.hip matching.hip keypoints.hip ransac.hip executor.hip codecs.hip backward.hip | sed 's#^#imfnet_amd/csrc/#')
if [ "$1" = build ]; then
  mkdir -p imfnet_amd/_abl
  for m in $2; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DIMF_GEO_ABL=$m $SRCS -o imfnet_amd/_abl/geo_$m.so -lz & done; wait
else
  for m in $2; do echo -n "mask $m: "; IMF_LIB=$PWD/imfnet_amd/_abl/geo_$m.so timeout 200 python tools/insert_time.py 2>&1 | tail -1; done
fi
