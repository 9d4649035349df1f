#!/bin/bash
# a variant of the library with scan_lane.hip compiled under extra -D flags: scripts/dev/mkdef.sh <name> "-DANDI_STITCH_FIRST=24 ..." -> andi_amd/libandihip_<name>.so (use with ANDI_HIP_LIB)
set -e
cd /root/repo/andi_amd/csrc
name=$1; defs=$2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I/root/repo/include -I. $defs -c scan_lane.hip -o build/scan_lane_$name.o
objs=$(ls build/api.o build/esa_build.o build/scan.o build/scan_coop.o build/scan_lane_quad.o build/sa_device.o build/bootstrap.o build/host_sais.o build/host_seq.o build/host_model.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/andi_amd/libandihip_$name.so $objs build/scan_lane_$name.o -lm -lpthread -ldl
