#!/bin/bash
# build a VARIANT of the library for A/B measurements: tools/build_variant.sh NAME FILE.hip "-DX=1 ..."  -> libxsmm_amd/lib/variants/NAME/libxsmm_amd.so
# (one translation unit recompiled with extra defines, linked with the objects of the shipped build; a GPU script copies it over libxsmm_amd/lib/libxsmm_amd.so on the box)
set -e
cd "$(dirname "$0")/../libxsmm_amd/csrc"
name=$1; src=$2; defs=$3
out=../lib/variants/$name; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-result -I../../include $defs -c $src -o $out/$src.o
objs=$(ls ../lib/obj/*.o | grep -v "/$src.o" | grep -v mono)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $out/$src.o -ldl -o $out/libxsmm_amd.so
rm -f $out/$src.o
ls -la $out/libxsmm_amd.so
