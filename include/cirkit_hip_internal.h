/* cirkit_hip_internal.h -- entry points libcirkit_hip.so exports that are NOT part of the drop-in boundary
 * (include/cirkit_hip.h): process-wide switches the parity tests use to run a layer through BOTH of its kernels.  Nothing a
 * reference-side binding would call. */
#ifndef CIRKIT_HIP_INTERNAL_H
#define CIRKIT_HIP_INTERNAL_H

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: route ck_sum_lse_fwd (and the region / CP-block launches) through the shape-generic kernels even where the MFMA
 * kernels apply (A/B parity of the two implementations, tests/test_gpu_layers.py). */
int ck_debug_force_generic(int on);
/* The same for ck_sum_lse_bwd (tests/test_training.py::test_mfma_and_generic_backward_kernels_agree). */
int ck_debug_force_generic_bwd(int on);

#ifdef __cplusplus
}
#endif
#endif /* CIRKIT_HIP_INTERNAL_H */
