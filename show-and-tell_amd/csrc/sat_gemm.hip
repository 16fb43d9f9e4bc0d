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
    // Fused vocab projection + CE (models.py:53 + train.py:53,143).  XF = 1 (epilogue): besides C the kernel emits, per
    // output row and per (tile column, wave column), the max and the sum of exp(x - max) over that wave's columns ->
    // lse_part[row][2*tiles_n][2]; a tiny combine kernel turns them into the row's log-sum-exp, so the CE never re-reads
    // the logits.  XF = 2 (operand): the A operand is the softmax gradient of the stored logits, formed on the fly --
    // a = (exp(x - lse[row]) - (col == target[row])) * inv_denom, 0 beyond the V valid columns -- so d(loss)/d(logits) is
    // never written; with AMODE = AM_KM (dW = dlogits^T Hs) the kernel also accumulates the column sums of that operand
    // = the bias gradient (tile column 0 writes them).
    float* lse_part;
    const float* lse;
    const int64_t* targets;
    float inv_denom;
    int xf_cols;         // V: valid columns of the logits
    float* colsum_out;   // XF = 2, AM_KM: [M] bias gradient
};

template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4 type; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; };

template <typename T, int BM, int BN, int AMODE, int BMODE, int XF = 0>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    static_assert(XF == 0 || sizeof(T) == 4, "the CE fusions are f32");
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
    // XF = 2: AM_ROW -- the row's log-sum-exp and target are K-invariant; AM_KM -- running column sums of the operand
    float xf_lse[NA];
    int xf_tgt[NA];
    float xf_colsum[NA][4];
    int xf_kt = 0;
    if constexpr (XF == 2) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            xf_lse[i] = 0.0f; xf_tgt[i] = -1;
#pragma unroll
            for (int e = 0; e < 4; ++e) xf_colsum[i][e] = 0.0f;
            if constexpr (AMODE == AM_ROW) {
                const int grow = m0 + (tid >> 3) + i * 32;
                if (grow < p.M) { xf_lse[i] = p.lse[grow]; xf_tgt[i] = (int)p.targets[grow]; }
            }
        }
    }

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
                if constexpr (XF == 2) {          // k = packed token: its log-sum-exp / target travel with the tile
                    xf_lse[i] = (k < p.K) ? p.lse[k] : 0.0f;
                    xf_tgt[i] = (k < p.K) ? (int)p.targets[k] : -1;
                }
            }
        }
        xf_kt = kt;
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
        if constexpr (XF == 2) {
            // the softmax gradient is formed HERE -- after the MFMAs of the previous K-step, right before the LDS store -- so
            // the global loads above stay in flight under the matrix work
            const int k0 = xf_kt * BK;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                f32x4 v = *(f32x4*)&ra[i];
                if constexpr (AMODE == AM_ROW) {          // rows = packed tokens, k = vocab column
                    const int kk = k0 + kc * CH;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = kk + e;
                        v[e] = (a_base[i] >= 0 && col < p.xf_cols)
                                   ? (expf(v[e] - xf_lse[i]) - (col == xf_tgt[i] ? 1.0f : 0.0f)) * p.inv_denom : 0.0f;
                    }
                } else {                                  // AM_KM: k = packed token, m = vocab column
                    const int qid = tid + i * 256;
                    const int krow = qid / (BM / CH), mc = qid - krow * (BM / CH);
                    const int k = k0 + krow, m = m0 + mc * CH;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = m + e;
                        v[e] = (k < p.K && col < p.xf_cols) ? (expf(v[e] - xf_lse[i]) - (col == xf_tgt[i] ? 1.0f : 0.0f)) * p.inv_denom : 0.0f;
                        xf_colsum[i][e] += v[e];
                    }
                }
                ra[i] = *(u32x4*)&v;
            }
        }
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

    if constexpr (XF == 1) {
        // per row: max and sum exp(x - max) over THIS wave's WN columns (bias included, columns >= N excluded); the 32 lanes
        // that share h hold one row's 32 columns per j: butterfly within the half-wave
        const int part_ld = 2 * p.tiles_n;                         // partials per row: (tile_n, wn)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * WN + j * 32 + r;
                    float badd = 0.0f;
                    if (col < p.N) {
                        if (p.bias) badd += p.bias[col];
                        if (p.bias2) badd += p.bias2[col];
                        mx = fmaxf(mx, acc[i][j][e] + badd);
                    }
                }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                float sm = 0.0f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * WN + j * 32 + r;
                    if (col < p.N) {
                        float badd = 0.0f;
                        if (p.bias) badd += p.bias[col];
                        if (p.bias2) badd += p.bias2[col];
                        sm += expf(acc[i][j][e] + badd - mx);
                    }
                }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) sm += __shfl_xor(sm, o, 64);
                const int row = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (r == 0 && row < p.M) {
                    float* dst = p.lse_part + ((long)row * part_ld + (tile_n * 2 + wn)) * 2;
                    dst[0] = mx;
                    dst[1] = sm;
                }
            }
        }
    }
    if constexpr (XF == 2 && AMODE == AM_KM) {
        // bias gradient: this thread summed its 4 columns (m) over every k it loaded; the threads that share the m chunk
        // (tid % (BM/4)) are reduced through LDS in a fixed order; tile column 0 writes
        if (tile_n == 0 && p.colsum_out && kz == 0) {
            __syncthreads();
            float* red = (float*)smem;                             // [256 / (BM/4) * NA][BM] floats
            constexpr int MC = BM / 4, G = 256 / MC;               // threads per k-row group, groups per block
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[((tid / MC) * NA + i) * BM + (tid % MC) * 4 + e] = xf_colsum[i][e];
            __syncthreads();
            for (int c = tid; c < BM; c += 256) {
                float sacc = 0.0f;
                for (int q = 0; q < G * NA; ++q) sacc += red[q * BM + c];
                if (m0 + c < p.M) p.colsum_out[m0 + c] = sacc;
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

template <typename T, int BM, int BN, int AMODE, int BMODE, int XF = 0>
int launch(GemmArgs& a, hipStream_t s) {
    const int tm = sat_cdiv(a.M, BM), tn = sat_cdiv(a.N, BN);
    a.tiles_n = tn;
    if (a.ksplit < 1) a.ksplit = 1;
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, AMODE, BMODE, XF>), dim3(tm * tn, a.ksplit), dim3(256), 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

template <int AMODE, int BMODE, int XF = 0>
int launch_f32_auto(GemmArgs& a, hipStream_t s) {
    // fill the 256 CUs: fall to smaller tiles when the big ones leave most of the chip idle
    const int ks = a.ksplit < 1 ? 1 : a.ksplit;
    static int force = -1;              // SAT_GEMM_TILE=1/2/3: tuning override (128x128 / 128x64 / 64x64)
    if (force < 0) { const char* e = getenv("SAT_GEMM_TILE"); force = e ? atoi(e) : 0; }
    if (force == 1) return launch<float, 128, 128, AMODE, BMODE, XF>(a, s);
    if (force == 2) return launch<float, 128, 64, AMODE, BMODE, XF>(a, s);
    if (force == 3) return launch<float, 64, 64, AMODE, BMODE, XF>(a, s);
    const long t128 = (long)sat_cdiv(a.M, 128) * sat_cdiv(a.N, 128) * ks;
    const long t12864 = (long)sat_cdiv(a.M, 128) * sat_cdiv(a.N, 64) * ks;
    if (t128 >= 384) return launch<float, 128, 128, AMODE, BMODE, XF>(a, s);
    if (t12864 >= 384) {
        // rounds of 256 workgroups x tile area x a per-flop penalty: 128x64 only where its last round is not mostly
        // empty (measured on the decoder's shapes, tools/microbench.py gemm: dW_vocab 138 -> 125 us with 64x64)
        const long t64 = (long)sat_cdiv(a.M, 64) * sat_cdiv(a.N, 64) * ks;
        const double c12864 = (double)((t12864 + 255) / 256) * 128 * 64 * 1.1, c64 = (double)((t64 + 255) / 256) * 64 * 64 * 1.25;
        if (c12864 <= c64) return launch<float, 128, 64, AMODE, BMODE, XF>(a, s);
    }
    return launch<float, 64, 64, AMODE, BMODE, XF>(a, s);
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
// fused vocab projection + cross entropy (models.py:53 + train.py:53,143-144)
namespace {
// per row: combine the (max, sum exp) partials of every (tile column, wave column) into the log-sum-exp; the row loss needs
// ONE logit (the target's) -- the logits themselves are not re-read
__global__ __launch_bounds__(256) void ce_combine_kernel(const float* __restrict__ part, int nparts, const float* __restrict__ logits,
                                                         long ldl, const int64_t* __restrict__ targets, int N, int V,
                                                         float* __restrict__ lse, float* __restrict__ row_loss) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* pp = part + (long)row * nparts * 2;
    float m = -INFINITY;
    for (int i = lane; i < nparts; i += 64) m = fmaxf(m, pp[2 * i]);
    m = wave_max(m);
    float s = 0.0f;
    for (int i = lane; i < nparts; i += 64) {
        const float pm = pp[2 * i];
        if (pm > -INFINITY) s += pp[2 * i + 1] * expf(pm - m);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float l = m + logf(s);
        long tg = targets[row];
        tg = tg < 0 ? 0 : (tg >= V ? V - 1 : tg);          // memory safety only (sat_validate_ids reports bad ids)
        lse[row] = l;
        row_loss[row] = l - logits[(long)row * ldl + tg];
    }
}
__global__ __launch_bounds__(256) void sum_scale2_kernel(const float* __restrict__ v, int n, float scale, float* out) {
    __shared__ float sh[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0] * scale;
}
int vocab_bwd_ksplit(int N, int H, int V) {
    const long tiles = (long)sat_cdiv(N, 64) * sat_cdiv(H, 64);
    int ks = (int)(512 / (tiles > 0 ? tiles : 1));
    const int nk = sat_cdiv(V, 32);
    if (ks > nk / 8) ks = nk / 8;
    return ks < 1 ? 1 : (ks > 16 ? 16 : ks);
}
}  // namespace

extern "C" int64_t sat_vocab_ce_fwd_ws_bytes(int N, int V) {
    return (int64_t)N * 2 * sat_cdiv(V, 128) * 2 * (int64_t)sizeof(float);
}

extern "C" int sat_vocab_ce_fwd(const float* Hs, const float* w, const float* b, const int64_t* targets, int N, int H, int V,
                                float inv_denom, float* logits, int64_t ldl, float* lse, float* row_loss, float* loss_out,
                                float* workspace, int64_t ws_bytes, sat_stream_t stream) {
    if (!Hs || !w || !b || !targets || !logits || !lse || !row_loss || N < 1 || H < 4 || (H & 3) || V < 1 || ldl < V) return SAT_ERR_ARG;
    if (!workspace || ws_bytes < sat_vocab_ce_fwd_ws_bytes(N, V)) return SAT_ERR_WORKSPACE;
    if (!aligned16(Hs) || !aligned16(w)) return SAT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    GemmArgs a = {};
    a.A = Hs; a.B = w; a.C = logits; a.bias = b;
    a.M = N; a.N = V; a.K = H; a.lda = H; a.ldb = H; a.ldc = ldl; a.ksplit = 1;
    a.lse_part = workspace;
    SAT_TRY((launch<float, 128, 128, AM_ROW, BMODE_NT, 1>(a, s)));
    const int nparts = 2 * sat_cdiv(V, 128);
    hipLaunchKernelGGL(ce_combine_kernel, dim3(sat_cdiv(N, 4)), dim3(256), 0, s, workspace, nparts, logits, (long)ldl, targets, N, V, lse, row_loss);
    SAT_LAUNCH_CHECK();
    if (loss_out) {
        hipLaunchKernelGGL(sum_scale2_kernel, dim3(1), dim3(256), 0, s, row_loss, N, inv_denom, loss_out);
        SAT_LAUNCH_CHECK();
    }
    return SAT_OK;
}

extern "C" int64_t sat_vocab_ce_bwd_fused_ws_bytes(int N, int H, int V) {
    const int ks = vocab_bwd_ksplit(N, H, V);
    return ks > 1 ? (int64_t)ks * N * H * (int64_t)sizeof(float) : 0;
}

extern "C" int sat_sum_slabs_f32(const float* in, int nslab, int64_t slab_stride, int64_t n, float* out, sat_stream_t stream);

extern "C" int sat_vocab_ce_bwd_fused(const float* logits, int64_t ldl, const float* lse, const int64_t* targets, float inv_denom,
                                      const float* Hs, const float* w, int N, int H, int V, float* dw, float* db, float* dHs,
                                      float* workspace, int64_t ws_bytes, sat_stream_t stream) {
    if (!logits || !lse || !targets || !Hs || !w || !dw || !db || !dHs || N < 1) return SAT_ERR_ARG;
    if ((H & 3) || (ldl & 3) || ldl < ((V + 3) & ~3) || !aligned16(logits) || !aligned16(Hs) || !aligned16(w)) return SAT_ERR_UNSUPPORTED;
    const int ks = vocab_bwd_ksplit(N, H, V);
    if (ks > 1 && (!workspace || ws_bytes < sat_vocab_ce_bwd_fused_ws_bytes(N, H, V) || (((long)N * H) & 3))) return SAT_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    // dW[V,H] = dlogits^T Hs with dlogits formed in the operand load; db = its column sums (same launch)
    GemmArgs a = {};
    a.A = logits; a.B = Hs; a.C = dw; a.M = V; a.N = H; a.K = N; a.lda = ldl; a.ldb = H; a.ldc = H; a.ksplit = 1;
    a.lse = lse; a.targets = targets; a.inv_denom = inv_denom; a.xf_cols = V; a.colsum_out = db;
    SAT_TRY((launch_f32_auto<AM_KM, BMODE_KM, 2>(a, s)));
    // dHs[N,H] = dlogits W  (K = V is long: split-K slabs summed in fixed order)
    GemmArgs c = {};
    c.A = logits; c.B = w; c.M = N; c.N = H; c.K = V; c.lda = ldl; c.ldb = H; c.ldc = H;
    c.lse = lse; c.targets = targets; c.inv_denom = inv_denom; c.xf_cols = V;
    if (ks == 1) {
        c.C = dHs; c.ksplit = 1;
        return launch_f32_auto<AM_ROW, BMODE_KM, 2>(c, s);
    }
    c.C = workspace; c.ksplit = ks; c.slab_stride = (long)N * H;
    SAT_TRY((launch_f32_auto<AM_ROW, BMODE_KM, 2>(c, s)));
    return sat_sum_slabs_f32(workspace, ks, (int64_t)N * H, (int64_t)N * H, dHs, stream);
}

extern "C" int sat_conv_tiles_m(int64_t M) { return sat_cdiv(M, 128); }

int sat_conv_glds_launch(const sat_op* op, int parity, hipStream_t s);   // sat_conv_glds.hip

static bool conv_legacy() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SAT_CONV_LEGACY"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

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
