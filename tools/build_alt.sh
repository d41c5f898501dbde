#!/bin/bash
# tools/build_alt.sh NAME KNOB=VAL ... — build a variant library into alt/NAME.so (for tools/gpu_ab.sh)
set -e
name=$1; shift
mkdir -p alt /tmp/alt_$name
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -I include -I headtrackr_amd/csrc"
for kv in "$@"; do FL="$FL -D$kv"; done
rm -f /tmp/alt_$name/*.o
for f in ht_context ht_pyramid ht_scan ht_camshift; do /opt/rocm/bin/hipcc $FL -c headtrackr_amd/csrc/$f.hip -o /tmp/alt_$name/$f.o & done; wait
for f in ht_context ht_pyramid ht_scan ht_camshift; do [ -f /tmp/alt_$name/$f.o ] || { echo "build_alt: $f.hip failed to compile"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt/$name.so /tmp/alt_$name/*.o -ldl
ls -la alt/$name.so
