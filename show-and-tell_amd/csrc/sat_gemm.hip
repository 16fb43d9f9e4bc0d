// LDS-tiled MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[M,N] = A[M,K] * B[N,K]^T  (+bias)        A is a plain matrix, a transposed matrix, or an NHWC
//                                              activation gathered on the fly (implicit GEMM, no im2col)
//
// One workgroup = 256 threads = 4 waves (2x2); each wave owns a (BM/2)x(BN/2) sub-tile built from 32x32
// MFMA tiles (v_mfma_f32_32x32x16_bf16 for bf16, v_mfma_f32_32x32x2_f32 -- exact f32 -- for f32).
// A K-step is 128 bytes of K per row (64 bf16 / 32 f32).  Global -> registers -> LDS staging with the
// next K-step's global loads in flight under the current step's MFMAs; LDS rows are padded 128 -> 144 B so
// the 16-byte fragment reads (ds_read_b128) are bank-conflict free.  The blockIdx -> tile map is
// XCD-aware: the tiles that share an activation panel run on one XCD and hit in its L2.
//
// Replaces: cuDNN conv under `self.resnet(images)` (models.py:27) and the cuBLAS GEMMs under nn.LSTM /
// nn.Linear (models.py:52-53) with their backward (train.py:144).
#include "sat_internal.h"
#include <stdlib.h>

namespace {

constexpr int ROWB = 144;  // LDS bytes per k-contiguous tile row (128 data + 16 pad)

enum { AM_ROW = 0, AM_CONV = 1, AM_KM = 2 };
enum { BMODE_NT = 0, BMODE_KM = 1 };

struct GemmArgs {
    const void* A;
    const void* B;
    void* C;
    const float* bias;
    const float* bias2;
    float* stat_partial;
    int M, N, K;
    long lda, ldb, ldc;
    int Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad, padw;
    long sN, sH, sW;
    int tiles_n;
    int ksplit;          // gridDim.y: K-steps are dealt to ksplit slices, slice z writes C + z*slab_stride
    long slab_stride;
};

template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4 type; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; };

template <typename T, int BM, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    constexpr int CH = 16 / (int)sizeof(T);    // elements per 16-byte chunk
    constexpr int BK = 128 / (int)sizeof(T);   // elements of K per step
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int NA = BM * 8 / 256, NB = BN * 8 / 256;  // 16-byte chunks per thread per K-step
    constexpr int A_BYTES = (AMODE == AM_KM) ? BK * BM * 4 : BM * ROWB;
    constexpr int B_BYTES = (BMODE == BMODE_KM) ? BK * BN * 4 : BN * ROWB;
    static_assert(sizeof(T) == 4 || (AMODE != AM_KM && BMODE != BMODE_KM), "bf16: k-contiguous operands only");
    __shared__ __attribute__((aligned(16))) char smem[A_BYTES + B_BYTES];
    char* As = smem;
    char* Bs = smem + A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware bijective remap: blocks b and b+8 share an XCD (speed only, never correctness)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_m = swz / p.tiles_n, tile_n = swz - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const T* __restrict__ Ag = (const T*)p.A;
    const T* __restrict__ Bg = (const T*)p.B;

    // ---- per-thread, K-invariant row state of the A loader ----
    const int kc = tid & 7;        // chunk (16 B) inside a k-contiguous 128-byte row
    long a_base[NA];
    int a_hi0[NA], a_wi0[NA];
    if constexpr (AMODE != AM_KM) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int grow = m0 + (tid >> 3) + i * 32;
            if constexpr (AMODE == AM_ROW) {
                a_base[i] = (grow < p.M) ? (long)grow * p.lda : -1;
                a_hi0[i] = 0; a_wi0[i] = 0;
            } else {
                if (grow < p.M) {
                    const int hw = p.Hout * p.Wout;
                    const int n = grow / hw;
                    const int rem = grow - n * hw;
                    const int ho = rem / p.Wout;
                    const int wo = rem - ho * p.Wout;
                    a_hi0[i] = ho * p.stride - p.pad;
                    a_wi0[i] = wo * p.stride - p.padw;
                    a_base[i] = (long)n * p.sN + (long)a_hi0[i] * p.sH + (long)a_wi0[i] * p.sW;
                } else {
                    a_hi0[i] = -(1 << 28); a_wi0[i] = -(1 << 28); a_base[i] = 0;
                }
            }
        }
    }
    long b_base[NB];
    if constexpr (BMODE == BMODE_NT) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int gn = n0 + (tid >> 3) + i * 32;
            b_base[i] = (gn < p.N) ? (long)gn * p.ldb : -1;
        }
    }
    const bool cin_uniform = (AMODE == AM_CONV) && (p.Cin % BK == 0);

    u32x4 ra[NA], rb[NB];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        // ---- A ----
        if constexpr (AMODE == AM_ROW) {
            const int kk = k0 + kc * CH;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const bool ok = (a_base[i] >= 0) && (kk < p.K);
                ra[i] = ok ? *(const u32x4*)(Ag + a_base[i] + kk) : zero4;
            }
        } else if constexpr (AMODE == AM_CONV) {
            const int kk = k0 + kc * CH;
            int tap, c;
            if (cin_uniform) {
                tap = k0 / p.Cin;
                c = k0 - tap * p.Cin + kc * CH;
            } else {
                tap = kk / p.Cin;
                c = kk - tap * p.Cin;
            }
            const int kh = tap / p.KW;
            const int kw = tap - kh * p.KW;
            const long koff = (long)kh * p.sH + (long)kw * p.sW + c;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
                const bool ok = (kk < p.K) && ((unsigned)hi < (unsigned)p.Hin) && ((unsigned)wi < (unsigned)p.Win);
                ra[i] = ok ? *(const u32x4*)(Ag + a_base[i] + koff) : zero4;
            }
        } else {  // AM_KM : A[k*lda + m], image [BK][BM]
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int qid = tid + i * 256;
                const int krow = qid / (BM / CH), mc = qid - krow * (BM / CH);
                const int k = k0 + krow, m = m0 + mc * CH;
                const bool ok = (k < p.K) && (m < p.M);
                ra[i] = ok ? *(const u32x4*)(Ag + (long)k * p.lda + m) : zero4;
            }
        }
        // ---- B ----
        if constexpr (BMODE == BMODE_NT) {
            const int kk = k0 + kc * CH;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const bool ok = (b_base[i] >= 0) && (kk < p.K);
                rb[i] = ok ? *(const u32x4*)(Bg + b_base[i] + kk) : zero4;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int qid = tid + i * 256;
                const int krow = qid / (BN / CH), nc = qid - krow * (BN / CH);
                const int k = k0 + krow, n = n0 + nc * CH;
                const bool ok = (k < p.K) && (n < p.N);
                rb[i] = ok ? *(const u32x4*)(Bg + (long)k * p.ldb + n) : zero4;
            }
        }
    };

    auto store_tiles = [&]() {
        if constexpr (AMODE != AM_KM) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                *(u32x4*)(As + ((tid >> 3) + i * 32) * ROWB + kc * 16) = ra[i];
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) *(u32x4*)(As + (tid + i * 256) * 16) = ra[i];
        }
        if constexpr (BMODE == BMODE_NT) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *(u32x4*)(Bs + ((tid >> 3) + i * 32) * ROWB + kc * 16) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) *(u32x4*)(Bs + (tid + i * 256) * 16) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk_all = (p.K + BK - 1) / BK;
    const int kz = blockIdx.y;
    const int per = (nk_all + p.ksplit - 1) / p.ksplit;
    const int kt0 = kz * per;
    const int nk = (kt0 + per < nk_all) ? kt0 + per : nk_all;
    if (kt0 < nk) load_tiles(kt0);
    for (int kt = kt0; kt < nk; ++kt) {
        store_tiles();
        __syncthreads();
        if (kt + 1 < nk) load_tiles(kt + 1);   // global loads in flight under the MFMAs below
        if constexpr (sizeof(T) == 2) {
            // bf16: 4 k-steps of 16; lane (r,h) holds A[row r][16ks + 8h .. +7], B likewise
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *(const bf16x8*)(As + (wm * WM + i * 32 + r) * ROWB + ks * 32 + h * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[j] = *(const bf16x8*)(Bs + (wn * WN + j * 32 + r) * ROWB + ks * 32 + h * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // f32: 4 groups of 8 k; lane half h takes k = 8kq + 4h + e for MFMA e (same map for A and B,
            // so each 32x32x2 MFMA contracts k in {8kq+e, 8kq+4+e}: a permutation of the K order only)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * WM + i * 32 + r;
                    if constexpr (AMODE != AM_KM) {
                        af[i] = *(const f32x4*)(As + row * ROWB + (kq * 8 + h * 4) * 4);
                    } else {
                        const float* a = (const float*)As + (kq * 8 + h * 4) * BM + row;
                        af[i][0] = a[0]; af[i][1] = a[BM]; af[i][2] = a[2 * BM]; af[i][3] = a[3 * BM];
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * WN + j * 32 + r;
                    if constexpr (BMODE == BMODE_NT) {
                        bf[j] = *(const f32x4*)(Bs + row * ROWB + (kq * 8 + h * 4) * 4);
                    } else {
                        const float* b = (const float*)Bs + (kq * 8 + h * 4) * BN + row;
                        bf[j][0] = b[0]; bf[j][1] = b[BN]; bf[j][2] = b[2 * BN]; bf[j][3] = b[3 * BN];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
    T* __restrict__ Cg = (T*)p.C + (long)kz * p.slab_stride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WN + j * 32 + r;
        float badd = 0.0f;
        if (col < p.N && kz == 0) {
            if (p.bias) badd += p.bias[col];
            if (p.bias2) badd += p.bias2[col];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < p.M && col < p.N) Cg[(long)row * p.ldc + col] = from_f32<T>(acc[i][j][e] + badd);
            }
        }
    }

    if (p.stat_partial) {
        // per-tile column sum / sum of squares of the f32 accumulators (rows >= M are exact zeros)
        float* red = (float*)smem;  // [wm][2][BN]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.0f, q = 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = acc[i][j][e];
                    s += v;
                    q += v * v;
                }
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (h == 0) {
                red[(wm * 2 + 0) * BN + wn * WN + j * 32 + r] = s;
                red[(wm * 2 + 1) * BN + wn * WN + j * 32 + r] = q;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += 256) {
            const int col = n0 + c;
            if (col < p.N) {
                p.stat_partial[((long)tile_m * 2 + 0) * p.N + col] = red[(0 * 2 + 0) * BN + c] + red[(1 * 2 + 0) * BN + c];
                p.stat_partial[((long)tile_m * 2 + 1) * p.N + col] = red[(0 * 2 + 1) * BN + c] + red[(1 * 2 + 1) * BN + c];
            }
        }
    }
}

template <typename T, int BM, int BN, int AMODE, int BMODE>
int launch(GemmArgs& a, hipStream_t s) {
    const int tm = sat_cdiv(a.M, BM), tn = sat_cdiv(a.N, BN);
    a.tiles_n = tn;
    if (a.ksplit < 1) a.ksplit = 1;
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, AMODE, BMODE>), dim3(tm * tn, a.ksplit), dim3(256), 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

template <int AMODE, int BMODE>
int launch_f32_auto(GemmArgs& a, hipStream_t s) {
    // fill the 256 CUs: fall to smaller tiles when the big ones leave most of the chip idle
    const int ks = a.ksplit < 1 ? 1 : a.ksplit;
    const long t128 = (long)sat_cdiv(a.M, 128) * sat_cdiv(a.N, 128) * ks;
    const long t12864 = (long)sat_cdiv(a.M, 128) * sat_cdiv(a.N, 64) * ks;
    if (t128 >= 384) return launch<float, 128, 128, AMODE, BMODE>(a, s);
    if (t12864 >= 384) {
        // rounds of 256 workgroups x tile area x a per-flop penalty: 128x64 only where its last round is not mostly
        // empty (measured on the decoder's shapes, tools/microbench.py gemm: dW_vocab 138 -> 125 us with 64x64)
        const long t64 = (long)sat_cdiv(a.M, 64) * sat_cdiv(a.N, 64) * ks;
        const double c12864 = (double)((t12864 + 255) / 256) * 128 * 64 * 1.1, c64 = (double)((t64 + 255) / 256) * 64 * 64 * 1.25;
        if (c12864 <= c64) return launch<float, 128, 64, AMODE, BMODE>(a, s);
    }
    return launch<float, 64, 64, AMODE, BMODE>(a, s);
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int sat_gemm_f32_splitk(int amode, int bmode, const float* A, int64_t lda, const float* B, int64_t ldb,
                                   float* C, int64_t ldc, const float* bias, const float* bias2, int M, int N, int K,
                                   int ksplit, int64_t slab_stride, sat_stream_t stream);

extern "C" int sat_gemm_f32(int amode, int bmode, const float* A, int64_t lda, const float* B, int64_t ldb,
                            float* C, int64_t ldc, const float* bias, const float* bias2,
                            int M, int N, int K, sat_stream_t stream) {
    return sat_gemm_f32_splitk(amode, bmode, A, lda, B, ldb, C, ldc, bias, bias2, M, N, K, 1, 0, stream);
}

extern "C" int sat_gemm_f32_splitk(int amode, int bmode, const float* A, int64_t lda, const float* B, int64_t ldb,
                                   float* C, int64_t ldc, const float* bias, const float* bias2, int M, int N, int K,
                                   int ksplit, int64_t slab_stride, sat_stream_t stream) {
    if (ksplit < 1 || ksplit > 64 || (ksplit > 1 && slab_stride < (int64_t)M * ldc)) return SAT_ERR_ARG;
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return SAT_ERR_ARG;
    if (!aligned16(A) || !aligned16(B) || (lda & 3) || (ldb & 3)) return SAT_ERR_ARG;
    // 16-byte chunks run along K (k-contiguous operands), M (amode 2) or N (bmode 1).  A ragged last chunk is fine
    // when the operand's rows are padded (ld >= the extent rounded up to 4) and the pad holds zeros.
    const bool a_pad = lda >= ((amode == 0 ? (int64_t)K : (int64_t)M) + 3) / 4 * 4;
    if (bmode == 0 && (K & 3)) return SAT_ERR_ARG;
    if (amode == 0 && (K & 3) && !a_pad) return SAT_ERR_ARG;
    if (amode == 2 && (M & 3) && !a_pad) return SAT_ERR_ARG;
    if (bmode == 1 && (N & 3)) return SAT_ERR_ARG;
    GemmArgs a = {};
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.bias2 = bias2; a.stat_partial = nullptr;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.ksplit = ksplit; a.slab_stride = slab_stride;
    hipStream_t s = (hipStream_t)stream;
    if (amode == 0 && bmode == 0) return launch_f32_auto<AM_ROW, BMODE_NT>(a, s);
    if (amode == 0 && bmode == 1) return launch_f32_auto<AM_ROW, BMODE_KM>(a, s);
    if (amode == 2 && bmode == 1) return launch_f32_auto<AM_KM, BMODE_KM>(a, s);
    if (amode == 2 && bmode == 0) return launch_f32_auto<AM_KM, BMODE_NT>(a, s);
    return SAT_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------
// vocab projection + cross entropy (models.py:53 + train.py:53,143) as one entry point: the logits GEMM, then the row-wise CE
// that overwrites the logits with d(loss)/d(logits) -- what the fused trainer does in its f32 mode.  (A form with the
// log-sum-exp in the GEMM epilogue and the softmax gradient formed inside the gradient GEMMs' operand loads was built in round
// 2, measured slower -- 6.94 vs 6.70 ms/step, profiles/r02_fused_ce_ab.txt: expf in the epilogue / operand path costs more
// than the passes over Infinity-Cache-resident logits it saves -- and removed in round 4.)
extern "C" int sat_vocab_logits_fwd(const float* Hs, const float* w, const float* b, int N, int H, int V, float* logits, int64_t ldl,
                                    sat_stream_t stream);
extern "C" int sat_ce_rows(float* logits, int64_t ldl, const int64_t* targets, int N, int V, float inv_denom, int write_grad,
                           float* row_loss, float* loss_out, sat_stream_t stream);

extern "C" int sat_vocab_ce_fwd(const float* Hs, const float* w, const float* b, const int64_t* targets, int N, int H, int V,
                                float inv_denom, float* logits, int64_t ldl, float* row_loss, float* loss_out, sat_stream_t stream) {
    if (!targets || !row_loss || !loss_out) return SAT_ERR_ARG;
    SAT_TRY(sat_vocab_logits_fwd(Hs, w, b, N, H, V, logits, ldl, stream));
    return sat_ce_rows(logits, ldl, targets, N, V, inv_denom, 1, row_loss, loss_out, stream);
}

extern "C" int sat_conv_tiles_m(int64_t M) { return sat_cdiv(M, 128); }

int sat_conv_glds_launch(const sat_op* op, int parity, hipStream_t s);   // sat_conv_glds.hip

static bool conv_legacy() { return false; }

// SAT_OP_CONV
int sat_conv_launch(const sat_op* op, int parity, hipStream_t s) {
    if (!op->in0 || !op->w || !op->out) return SAT_ERR_ARG;
    const int esz = op->dtype == SAT_BF16 ? 2 : 4;
    const int ch = 16 / esz;
    if (op->Cin % ch) return SAT_ERR_ARG;             // a 16-byte chunk must stay inside one pixel
    if (!aligned16(op->in0) || !aligned16(op->w)) return SAT_ERR_ARG;
    // every 16-byte chunk address n*sN + hi*sH + wi*sW + c must be 16-byte aligned
    if ((op->sN % ch) || (op->sH % ch)) return SAT_ERR_ARG;
    if ((op->KW > 1 || op->pad || (op->flags & SAT_CONV_PADW)) ? (op->sW % ch) != 0 : ((long)op->stride * op->sW) % ch != 0) return SAT_ERR_ARG;
    GemmArgs a = {};
    a.A = op->in0; a.B = op->w; a.C = op->out; a.bias = nullptr; a.bias2 = nullptr;
    a.stat_partial = op->stat_partial;
    a.M = op->N * op->Hout * op->Wout; a.N = op->Cout; a.K = op->KH * op->KW * op->Cin;
    a.lda = 0; a.ldb = a.K; a.ldc = op->Cout;
    a.Hin = op->Hin; a.Win = op->Win; a.Cin = op->Cin; a.Hout = op->Hout; a.Wout = op->Wout;
    a.KH = op->KH; a.KW = op->KW; a.stride = op->stride; a.pad = op->pad;
    a.padw = (op->flags & SAT_CONV_PADW) ? op->pad_w : op->pad;
    if (op->ldc) { if (op->ldc < op->Cout) return SAT_ERR_ARG; a.ldc = op->ldc; }
    a.sN = op->sN; a.sH = op->sH; a.sW = op->sW;
    if (op->stat_partial && op->tiles_m != sat_cdiv(a.M, 128)) return SAT_ERR_ARG;
    if (op->dtype == SAT_BF16 && (op->Cout % 8) == 0 && !conv_legacy()) return sat_conv_glds_launch(op, parity, s);
    if (op->scale1 || op->shift1 || op->in1) return SAT_ERR_UNSUPPORTED;   // fused inference epilogue: bf16 LDS-DMA kernel only
    // register-staged kernel (f32 parity mode, odd shapes).  BM is always 128 (it fixes the partial-slab geometry); BN 64 for narrow layers or to fill the chip
    const long t128 = (long)sat_cdiv(a.M, 128) * sat_cdiv(a.N, 128);
    const bool narrow = (a.N <= 64) || (t128 < 512);
    if (op->dtype == SAT_BF16) {
        return narrow ? launch<bf16_t, 128, 64, AM_CONV, BMODE_NT>(a, s) : launch<bf16_t, 128, 128, AM_CONV, BMODE_NT>(a, s);
    } else {
        return narrow ? launch<float, 128, 64, AM_CONV, BMODE_NT>(a, s) : launch<float, 128, 128, AM_CONV, BMODE_NT>(a, s);
    }
}
