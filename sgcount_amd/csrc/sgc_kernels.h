// sgc_kernels.h — host-callable launchers of the gfx950 kernels (sgc_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sgc_format.h"

void sgc_launch_count_direct(hipStream_t st, const uint64_t *recs, uint64_t n, uint32_t L, bool rec16,
                             const sgc_table_view &lib, const sgc_table_view &perm, bool one_mm, uint32_t *counts,
                             unsigned long long *matched);
void sgc_launch_lookup(hipStream_t st, const uint64_t *keys, uint64_t n, const sgc_table_view &lib,
                       const sgc_table_view &perm, int which, bool has_perm, int32_t *out);
void sgc_launch_fold(hipStream_t st, uint32_t *c32, unsigned long long *c64, uint32_t n);
void sgc_launch_pack_reads(hipStream_t st, const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t L,
                           bool rec16, int reverse, uint32_t o, int recursion, uint64_t *recs);
