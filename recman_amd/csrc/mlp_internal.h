// Internal (not part of the C ABI): pieces of mlp.hip that step.hip reuses.
#pragma once
#include "rm_common.h"

// Layout of one block's small-gradient partial (mlp.hip: mlp_small_grads_mfma / mlp_finish_kernel):
// [2 x 1024 dW_l (l >= 1) | NL x 32 db_l | 32 d w_out | 1 sum g (+31 pad) | 32 g^T xd].
constexpr int kRmSgDense = 2 * 1024 + 3 * 32 + 32 + 32;  // offset of the g^T xd slots
constexpr int kRmSgStride = kRmSgDense + 32;

// The finishing launch of the skinny MLP's backward (mlp_finish_kernel): dW0 from `nslab` slabs [Kp][32],
// the small gradients from `nblk2` per-block partials of kRmSgStride floats, the loss from n_loss partial sums.
// dW / db: NL entries (dW[0] unused).  Any output pointer may be NULL.
int rm_internal_mlp_finish(const float *dw0_part, int nslab, int K, int Kp, int H0, float *dW0, const float *sg_part,
                           int nblk2, int NL, const int *H, float *const *dW, float *const *db, float *d_w_out,
                           float *d_w0_out, float *d_xd_wsum, float *d_g_sum, int Dn, const float *loss_partial,
                           int64_t n_loss, int64_t B, float *loss, hipStream_t st);
