#!/bin/bash
# a variant of the SHIPPED library with ONE translation unit compiled under extra -D flags:
#   scripts/dev/mkvar.sh <name> <unit: scan_coop|scan_lane|esa_build|api|sa_device|scan> "-DFOO=1 ..."  -> andi_amd/libandihip_<name>.so
# (A/B runs: scripts/dev/ablib.sh libandihip.so libandihip_<name>.so; `make clean-variants` removes them)
# HOOKS=1: a variant of the test-hook build (libandihip_test.so: the experiment switches of knobs.h work in it)
set -e
cd "$(dirname "$0")/../../andi_amd/csrc"
name=$1; unit=$2; defs=$3
extra=""
hookdir=""
if [ -n "$HOOKS" ]; then defs="$defs -DANDI_TEST_HOOKS"; hookdir="hooks/"; fi
[ "$unit" = scan_coop ] && extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -I. $extra $defs -c $unit.hip -o build/v_${unit}_$name.o
objs=""
for o in api esa_build scan scan_coop scan_lane scan_lane_quad sa_device bootstrap host_sais host_seq host_model; do
  if [ "$o" = "$unit" ]; then objs="$objs build/v_${unit}_$name.o"; elif [ -n "$hookdir" ] && [ -f build/hooks/$o.o ]; then objs="$objs build/hooks/$o.o"; else objs="$objs build/$o.o"; fi
done
if [ "$unit" = scan_lane ]; then # (its second compilation: k_lane_quad with one wavefront per block)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -I. $defs -DANDI_QUAD_TU -DWAVES_PER_BLOCK=1 -c scan_lane.hip -o build/v_scan_lane_quad_$name.o
  objs=${objs/build\/scan_lane_quad.o/build\/v_scan_lane_quad_$name.o}
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libandihip_$name.so $objs -lm -lpthread -ldl
echo "andi_amd/libandihip_$name.so"
