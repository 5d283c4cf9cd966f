/* salva_hip_diag.h — entry points that exist only in the kernel-development build (`make -C salva_amd/csrc VARIANT=diag` ->
 * libsalva_hip_diag.so, compiled with -DSALVA_HIP_DIAG).  Not part of the drop-in boundary (include/salva_hip.h); bench.py and the
 * tests never load the diag library. */
#pragma once
#include "../../../include/salva_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics for kernel development (tools/variant_probe.py): times execution variant `variant` of the same kernel
 * (0: one tile per workgroup; 1: the same with co-resident workgroups de-phased by `param` x 64 cycles; 2: persistent
 * double-buffered pipeline; 3: one tile per workgroup with LDS-DMA staging) and returns a checksum of the kappa and
 * error partials it wrote, so that variants can be checked for bit-identical results. */
float salva_hip_time_variant(SalvaHipWorld* world, int32_t variant, uint32_t param, int32_t reps, uint64_t* checksum);

#ifdef __cplusplus
}
#endif
