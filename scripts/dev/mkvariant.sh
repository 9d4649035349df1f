#!/bin/bash
# a variant of the library with scan_coop.hip edited by a sed expression: scripts/dev/mkvariant.sh <name> '<sed expr>' -> andi_amd/libandihip_<name>.so (use with ANDI_HIP_LIB)
set -e
cd /root/repo/andi_amd/csrc
name=$1; expr=$2; src=${3:-scan_coop}
sed "$expr" $src.hip > /tmp/${src}_$name.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -I/root/repo/include -I/root/repo/andi_amd/csrc -c /tmp/${src}_$name.hip -o build/${src}_$name.o
objs=$(ls build/api.o build/esa_build.o build/scan.o build/scan_coop.o build/scan_lane.o build/scan_lane_quad.o build/sa_device.o build/bootstrap.o build/host_sais.o build/host_seq.o build/host_model.o | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/andi_amd/libandihip_$name.so $objs build/${src}_$name.o -lm -lpthread -ldl
bash /root/repo/scripts/kregs.sh build/${src}_$name.o | grep "Li4E" || true
