#!/bin/bash
# build tools/ecn_exp_<tag>_<CURVE>.bin from the curve-layer sources in <csrc dir> (see tools/ecn_exp.hip)
#   tools/build_ecn_exp.sh <tag> <csrc dir> <CURVE> [extra hipcc flags]
set -e
tag=$1; src=$2; c=$3; shift 3
case $c in
  ED25519|ED448|ED248|ED376|ED500|NUMS256E) cls="ma::Edwards<ma::C_$c>";;
  *) cls="ma::Weierstrass<ma::C_$c>";;
esac
here=$(cd "$(dirname "$0")" && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-codegenprepare-mul24=0 -I"$src" \
  -DCURVE_HDR="\"generated/curve_$c.h\"" -DCURVE_CLASS="$cls" -DCURVE_NAME="\"$c/$tag\"" "$@" \
  "$here/ecn_exp.hip" -o "$here/ecn_exp_${tag}_$c.bin" 2>&1 | grep -v "hip-link" || true
