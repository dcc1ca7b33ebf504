// The optimizer's update of one parameter entry inside a backward epilogue (ck_jobs.hip's job epilogues, ck_table_dense_bwd,
// ck_param_softmax_bwd_batch): torch.optim.Adam without weight decay / amsgrad, or SGD; the step's constants and clock live in a
// DEVICE ck_opt_state that ck_opt_tick advances once per step (a batch with an illegal category: skip_now, nothing changes).
#pragma once

#include "ck_internal.h"

namespace {

// Adam's update of one entry (torch.optim.Adam without weight decay / amsgrad; the bias corrections come from `ck_opt_tick`)
// (the step's constants once per thread: lr / bc1 and 1 / bc2, so that an entry costs v_sqrt_f32 + v_rcp_f32 -- 1 ulp each --
//  instead of three IEEE divisions and a library square root: the Categorical epilogue updates 16 K entries per workgroup)
struct OptK {
  float b1, c1, b2, c2, step, rbc2, eps, lr;
  int kind;
};
__device__ __forceinline__ OptK opt_k(const ck_opt_state& o) {
  return {o.b1, 1.f - o.b1, o.b2, 1.f - o.b2, o.lr / o.bc1, 1.f / o.bc2, o.eps, o.lr, o.kind};
}
__device__ __forceinline__ float opt_update(const OptK& o, float p, float g, float& m1, float& m2) {
  if (o.kind == 0) return p - o.lr * g;
  m1 = fmaf(o.b1, m1, o.c1 * g);
  m2 = fmaf(o.b2, m2, o.c2 * g * g);
  return fmaf(-o.step * m1, __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(m2 * o.rbc2) + o.eps), p);
}

}  // namespace
