// sa_device.h — suffix array construction on the device (sa_device.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// bytes of device workspace andi_sa_device needs for a text of n characters
size_t andi_sa_device_workspace(int32_t n);
// SA[0..n) of the text S[0..n) (device pointers; S readable 64 bytes past n, zeros there), bytes in unsigned
// order -- what divsufsort() computes at src/esa.c:303.  h_pinned4: four ints of pinned host memory.  Synchronises
// the stream once per round.  hipErrorInvalidSymbol: the text holds a byte outside {A C G T ! ; #}.
// rec (n entries, or null): the suffixes' records for a probe table of depth recK, in suffix-array order
// (esa_build.hip: suffix_rec) -- a by-product of the first round's sorted keys.  rec2 (n entries, or null): from the
// same keys, the up to four symbols behind each suffix's first recK (andi_dev.h: DEEP_SINGLE, the short extended form).
hipError_t andi_sa_device(const uint8_t *S, int32_t n, int32_t *SA, void *workspace, size_t workspace_bytes,
						  int32_t *h_pinned4, hipStream_t st, int *rounds_out, uint32_t *rec, int recK, uint16_t *rec2 = nullptr);
