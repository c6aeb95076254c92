#!/bin/bash
# A/B builds of the C-ABI library: tools/variant.sh <tag> [extra hipcc flags...] -> tools/build/libvms_<tag>.so
# (select at run time with VMS_HIP_LIB=tools/build/libvms_<tag>.so)
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/video-mamba-suite_amd/csrc
out=$root/tools/build/$tag
mkdir -p $out
for f in $src/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function "$@" -c $f -o $out/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/build/libvms_$tag.so $out/*.o
echo built tools/build/libvms_$tag.so
