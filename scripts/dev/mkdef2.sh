#!/bin/bash
# a variant of the library with the scan kernels (scan_lane.hip twice, scan_coop.hip) compiled under extra -D flags:
# scripts/dev/mkdef2.sh <name> "-DANDI_MULTI_MAX=8" -> andi_amd/libandihip_<name>.so (use with ANDI_HIP_LIB)
set -e
cd /root/repo/andi_amd/csrc
name=$1; defs=$2
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I/root/repo/include -I. $defs"
/opt/rocm/bin/hipcc $F -c scan_lane.hip -o build/v_scan_lane_$name.o &
/opt/rocm/bin/hipcc $F -DANDI_QUAD_TU -DWAVES_PER_BLOCK=1 -c scan_lane.hip -o build/v_scan_lane_quad_$name.o &
/opt/rocm/bin/hipcc $F -c scan_coop.hip -o build/v_scan_coop_$name.o &
wait
objs=$(ls build/api.o build/esa_build.o build/scan.o build/sa_device.o build/bootstrap.o build/host_sais.o build/host_seq.o build/host_model.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/andi_amd/libandihip_$name.so $objs build/v_scan_lane_$name.o build/v_scan_lane_quad_$name.o build/v_scan_coop_$name.o -lm -lpthread -ldl
