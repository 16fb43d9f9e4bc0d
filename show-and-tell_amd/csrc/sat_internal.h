// Host-side launchers shared between the translation units of libsat_hip.so (not part of the C ABI).
#pragma once
#include "sat_common.h"
#include "../../include/sat_hip.h"

int sat_conv_launch(const sat_op* op, int parity, hipStream_t s);
int sat_image_prep_launch(const sat_op* op, hipStream_t s);
int sat_bn_finalize_launch(const sat_op* op, int parity, hipStream_t s);
int sat_bn_eval_batch_launch(const sat_op* op, hipStream_t s);
int sat_bn_act_launch(const sat_op* op, bool add, int parity, hipStream_t s);
int sat_bn_relu_maxpool_launch(const sat_op* op, int parity, hipStream_t s);
int sat_avgpool_launch(const sat_op* op, hipStream_t s);
int sat_maxpool2_launch(const sat_op* op, hipStream_t s);
int sat_pool3_launch(const sat_op* op, bool avg, hipStream_t s);
// bn3 from the Gram matrix of conv3's input (sat_gram.hip)
int sat_gram_launch(const sat_op* op, int parity, hipStream_t s);
int sat_gram_cov_launch(const sat_op* op, hipStream_t s);
int sat_gemm_bf16_op_launch(const sat_op* op, hipStream_t s);
int sat_bn_from_gram_launch(const sat_op* op, hipStream_t s);

// f32 operands on the bf16 matrix pipe (sat_gemm_bf16.hip): C[M,N] = op(A) op(B)^T + bias + bias2, operands row-major [rows][K]
// (kmajor 0) or [K][rows] (kmajor 1), cast / transposed into bf16 copies in `scratch`
int64_t sat_gemm_mixed_scratch_bytes(int M, int N, int K);
int sat_gemm_mixed_nt(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor, float* C, long ldc,
                      const float* bias, const float* bias2, int M, int N, int K, void* scratch, int64_t scratch_bytes, hipStream_t s);

int sat_skinny_store(const float* A, long lda, const float* W, long ldw, int wkm, int M, int N, int K, int nz,
                     float* out, long ldo, long slab_stride, const float* bias, hipStream_t s);
int sat_skinny_lstm(const float* h_prev, const float* w_hh, const float* x, const float* w_ih, int In,
                    const float* bias, const float* bias2, const float* xg, long ldxg, int M, int H,
                    float* c_state, float* ga, long ldga, float* cs, float* h_out, float* h_out2, int m2,
                    hipStream_t s);
int sat_lstm_bwd_step(const float* dHS, const float* DG_next, int n_next, const float* w_hh, const float* GA, const float* CS,
                      const float* CS_prev, float* dc_state, float* DG, int n, int H, hipStream_t s);
int sat_lstm_bwd_point_launch(const float* dHS, const float* dh_part, int nz, long slab_stride, int n_next,
                              const float* GA, const float* CS, const float* CS_prev, float* dc_state, float* DG,
                              int n, int H, hipStream_t s);
int sat_bn1d_fwd_launch(const float* part, int nz, long slab_stride, const float* b_fc, const float* gamma,
                        const float* beta, float* rm, float* rv, float momentum, float eps, int training, int B, int E,
                        float* zbuf, float* feats, float* xhat, float* rstd, hipStream_t s);
int sat_outer_wgrad_launch(const float* dz, const float* x, int B, int E, int F, float* dw, hipStream_t s);
int sat_bn1d_bwd_launch(const float* dy, const float* xhat, const float* rstd, const float* gamma, int B, int E,
                        float* dz, float* dgamma, float* dbeta, float* db_fc, hipStream_t s);

// fixed-point scale (2^22) of the integer-atomic BatchNorm statistics shared by the conv epilogue and its consumers
#define SAT_STAT_SCALE 4194304.0

#define SAT_TRY(expr)                 \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != SAT_OK) return rc__; \
    } while (0)
