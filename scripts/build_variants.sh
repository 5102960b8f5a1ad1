#!/bin/bash
# Builds A/B variants of libcrx.so next to it (cpprobotics_amd/alt_<name>.so); used with CRX_LIB_PATH.
# usage: scripts/build_variants.sh name1="-DFOO=1 -mllvm -bar" name2="..."
cd "$(dirname "$0")/../cpprobotics_amd/csrc"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function"
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  /opt/rocm/bin/hipcc $BASE $flags -shared -o ../alt_$name.so crx_api.hip && echo "built alt_$name.so ($flags)" &
done
wait
