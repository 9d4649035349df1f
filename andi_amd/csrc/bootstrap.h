// bootstrap.h — launch interface of the pairwise bootstrap kernel (bootstrap.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "andi_hip.h"

hipError_t andi_launch_bootstrap(const andi_hip_model *M_dev, andi_hip_model *B_dev, uint32_t n,
								 uint32_t replicates, uint64_t seed, hipStream_t st);
