#!/bin/bash
# usage: build_variant.sh name "extra flags"   -> build/variants/libks265hip_<name>.so (all sources, variant flags on every file)
set -e
name=$1; shift
mkdir -p /root/repo/build/variants
cd /root/repo/ks265codec_amd/csrc
mkdir -p /tmp/w/v_$name
pids=()
for s in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-parameter -Wno-unused-function $@ -c $s -o /tmp/w/v_$name/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build/variants/libks265hip_$name.so /tmp/w/v_$name/*.o
ls -la /root/repo/build/variants/libks265hip_$name.so
