// Show-Attend-Tell decoder kernels (`/root/reference/model2.py:38-111`, the model train.py:37 constructs): the soft
// attention step (`attention_layer`, model2.py:73-78) forward and backward, a 2x2/2 max-pool for the VGG16
// `features[:-3]` stack (model2.py:15-16), and small row utilities.  Decoder arithmetic is f32 (ocml tanhf / expf), like
// the Show-and-Tell decoder; the dense contractions around these kernels go through sat_gemm_f32 / the skinny kernels.
#include "sat_internal.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.0f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sh[i]);
    return r;
}

// Soft attention step (model2.py:73-78), spread over the chip in two launches per direction so that a 64-row decode step is
// not a 64-workgroup launch:
//   h_att[p,c] = tanh(ctx_enc[b,p,c] + proj[b,c]);  s[p] = sum_c h_att[p,c] * w_att[c];  alpha = softmax_p(s)
//   context[b,c] = (1/P) * sum_p alpha[p] * feats[b,p,c]            (the reference takes the MEAN of the weighted features)
// (1) row-dot kernel, grid (rows, position chunks): one wave per position, lanes stride the channels (16 B per lane), fixed-order
//     wave reduction -> raw scores (forward) or d_alpha (backward) in the workspace;
// (2) channel kernel, grid (rows, 64-channel chunks): every workgroup redoes the P-long softmax (a few hundred expf) and owns 64
//     channels; its 8 waves take positions p = w (mod 8), partials are combined through LDS in wave order (deterministic).
constexpr int kAttWaves = 8;

// TANH = true: v[p] = sum_c tanh(x[b,p,c] + add[b,c]) * mul[c]      (scores)
// TANH = false: v[p] = scale * sum_c x[b,p,c] * mul_row[b,c]         (d_alpha = feats . d_ctx / P)
template <bool TANH>
__global__ __launch_bounds__(256) void att_rowdot_kernel(const float* __restrict__ x, const float* __restrict__ add, long ld_add,
                                                         const float* __restrict__ mul, long ld_mul,
                                                         const float* __restrict__ mul2, long ld_mul2, float scale, int P, int C,
                                                         int pchunk, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [C] add | [C] mul
    float* s_add = sm;
    float* s_mul = sm + C;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        if (TANH) s_add[c] = add[(long)b * ld_add + c];
        s_mul[c] = mul[(long)b * ld_mul + c] + (mul2 ? mul2[(long)b * ld_mul2 + c] : 0.0f);
    }
    __syncthreads();
    const int p0 = blockIdx.y * pchunk;
    const int p1 = p0 + pchunk < P ? p0 + pchunk : P;
    const float* xb = x + (long)b * P * C;
    for (int p = p0 + wave; p < p1; p += 4) {
        float acc = 0.0f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = *(const f32x4*)(xb + (long)p * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += TANH ? tanhf(v[e] + s_add[c + e]) * s_mul[c + e] : v[e] * s_mul[c + e];
        }
        acc = wave_sum(acc);
        if (lane == 0) out[(long)b * P + p] = acc * scale;
    }
}

__global__ __launch_bounds__(kAttWaves * 64) void att_context_kernel(const float* __restrict__ scores, const float* __restrict__ feats,
                                                                     int P, int C, float* __restrict__ alpha,
                                                                     float* __restrict__ context, long ld_ctx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [P4] alpha | [8] scratch | [kAttWaves][64] partials
    const int P4 = (P + 3) & ~3;
    float* s_a = sm;
    float* s_red = sm + P4;
    float* s_part = s_red + 8;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int p = tid; p < P; p += blockDim.x) s_a[p] = scores[(long)b * P + p];
    __syncthreads();
    float m = -INFINITY;
    for (int p = tid; p < P; p += blockDim.x) m = fmaxf(m, s_a[p]);
    m = block_max(m, s_red);
    float z = 0.0f;
    for (int p = tid; p < P; p += blockDim.x) z += expf(s_a[p] - m);
    z = block_sum(z, s_red);
    const float inv = 1.0f / z;
    __syncthreads();
    for (int p = tid; p < P; p += blockDim.x) {
        const float a = expf(s_a[p] - m) * inv;
        s_a[p] = a;
        if (alpha && blockIdx.y == 0) alpha[(long)b * P + p] = a;
    }
    __syncthreads();
    const int c = blockIdx.y * 64 + lane;
    float acc = 0.0f;
    if (c < C) {
        const float* fb = feats + (long)b * P * C + c;
#pragma unroll 4
        for (int p = wave; p < P; p += kAttWaves) acc += s_a[p] * fb[(long)p * C];
    }
    s_part[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && c < C) {
        float r = 0.0f;
#pragma unroll
        for (int w = 0; w < kAttWaves; ++w) r += s_part[w * 64 + lane];
        context[(long)b * ld_ctx + c] = r / (float)P;
    }
}

// Backward of the step for one (row, 64-channel chunk) (h_att recomputed, never stored):
//   d_alpha[p] = (1/P) feats[b,p,:] . d_ctx[b,:]                    (att_rowdot_kernel<false>, in `d_alpha`)
//   d_s[p]     = alpha[p] * (d_alpha[p] - sum_q alpha[q] d_alpha[q])
//   d_pre[p,c] = d_s[p] * w_att[c] * (1 - h_att[p,c]^2)        -> d_ctx_enc[b,p,c] += d_pre   (accumulated over steps)
//   d_proj[b,c] = sum_p d_pre[p,c];   d_w_att partial[b,c] = sum_p d_s[p] * h_att[p,c]
__global__ __launch_bounds__(kAttWaves * 64) void att_bwd_channel_kernel(const float* __restrict__ ctx_enc,
                                                                         const float* __restrict__ proj, long ld_proj,
                                                                         const float* __restrict__ w_att,
                                                                         const float* __restrict__ alpha,
                                                                         const float* __restrict__ d_alpha,
                                                                         const float* __restrict__ d_ctx, long ld_dctx,
                                                                         const float* __restrict__ d_ctx2, long ld_dctx2, int P, int C,
                                                                         float* __restrict__ d_ctx_enc, float* __restrict__ d_proj,
                                                                         float* __restrict__ d_watt_part, float* __restrict__ d_feats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [P4] d_s | [P4] alpha | [8] | [2][kAttWaves][64]
    const int P4 = (P + 3) & ~3;
    float* s_ds = sm;
    float* s_al = sm + P4;
    float* s_red = sm + 2 * P4;
    float* s_part = s_red + 8;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float dot = 0.0f;
    for (int p = tid; p < P; p += blockDim.x) {
        const float a = alpha[(long)b * P + p], da = d_alpha[(long)b * P + p];
        s_al[p] = a;
        s_ds[p] = da;
        dot += a * da;
    }
    dot = block_sum(dot, s_red);
    for (int p = tid; p < P; p += blockDim.x) s_ds[p] = s_al[p] * (s_ds[p] - dot);
    __syncthreads();
    const int c = blockIdx.y * 64 + lane;
    float dp = 0.0f, dw = 0.0f;
    if (c < C) {
        const float pj = proj[(long)b * ld_proj + c], w = w_att[c];
        const float* ce = ctx_enc + (long)b * P * C + c;
        float* dce = d_ctx_enc + (long)b * P * C + c;
#pragma unroll 4
        for (int p = wave; p < P; p += kAttWaves) {
            const float ha = tanhf(ce[(long)p * C] + pj);
            const float ds = s_ds[p];
            const float dpre = ds * w * (1.0f - ha * ha);
            dce[(long)p * C] += dpre;
            dp += dpre;
            dw += ds * ha;
        }
        if (d_feats) {      // fine-tuning: context = mean_p alpha[p] feats[p]  =>  d feats[p,c] += alpha[p] * d_ctx[c] / P
            const float dc = (d_ctx[(long)b * ld_dctx + c] + (d_ctx2 ? d_ctx2[(long)b * ld_dctx2 + c] : 0.0f)) / (float)P;
            float* dfe = d_feats + (long)b * P * C + c;
#pragma unroll 4
            for (int p = wave; p < P; p += kAttWaves) dfe[(long)p * C] += s_al[p] * dc;
        }
    }
    s_part[wave * 64 + lane] = dp;
    s_part[(kAttWaves + wave) * 64 + lane] = dw;
    __syncthreads();
    if (wave == 0 && c < C) {
        float rp = 0.0f, rw = 0.0f;
#pragma unroll
        for (int w2 = 0; w2 < kAttWaves; ++w2) {
            rp += s_part[w2 * 64 + lane];
            rw += s_part[(kAttWaves + w2) * 64 + lane];
        }
        d_proj[(long)b * C + c] = rp;
        d_watt_part[(long)b * C + c] = rw;
    }
}

// 2x2 / stride 2 max-pool, NHWC, 16 B per lane
template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int Hin, int Win, int C) {
    constexpr int V = 16 / (int)sizeof(T);
    const int Ho = Hin / 2, Wo = Win / 2, cch = C / V;
    const long total = (long)N * Ho * Wo * cch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cch);
        long r = i / cch;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const T* p = in + (((long)n * Hin + 2 * ho) * Win + 2 * wo) * C + cc * V;
        float best[V];
#pragma unroll
        for (int k = 0; k < V; ++k) best[k] = -INFINITY;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const T* q = p + ((long)(d >> 1) * Win + (d & 1)) * C;
            if constexpr (sizeof(T) == 4) {
                const f32x4 x = *(const f32x4*)q;
#pragma unroll
                for (int k = 0; k < 4; ++k) best[k] = fmaxf(best[k], x[k]);
            } else {
                const bf16x8 x = *(const bf16x8*)q;
#pragma unroll
                for (int k = 0; k < 8; ++k) best[k] = fmaxf(best[k], (float)x[k]);
            }
        }
        T* o = out + i * V;
        if constexpr (sizeof(T) == 4) {
            *(f32x4*)o = (f32x4){best[0], best[1], best[2], best[3]};
        } else {
            bf16x8 y;
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = (bf16_t)best[k];
            *(bf16x8*)o = y;
        }
    }
}

// 3x3 pools of the Inception blocks (BASELINE configs[3]), NHWC, 16 B per lane.  MAX: stride 2, no padding, output row pitch
// ldo (the pooled branch is a slice of the block's concatenated output).  AVG: stride 1, pad 1, count_include_pad (/9).
template <typename T, bool AVG>
__global__ void pool3_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int Hin, int Win, int C, int Ho, int Wo, long ldo) {
    constexpr int V = 16 / (int)sizeof(T);
    const int cch = C / V;
    const long total = (long)N * Ho * Wo * cch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cch);
        long r = i / cch;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = AVG ? 0.0f : -INFINITY;
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = AVG ? ho - 1 + dh : ho * 2 + dh;
            if ((unsigned)hi >= (unsigned)Hin) continue;
            for (int dw = 0; dw < 3; ++dw) {
                const int wi = AVG ? wo - 1 + dw : wo * 2 + dw;
                if ((unsigned)wi >= (unsigned)Win) continue;
                const T* q = in + (((long)n * Hin + hi) * Win + wi) * C + cc * V;
                if constexpr (sizeof(T) == 4) {
                    const f32x4 x = *(const f32x4*)q;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = AVG ? acc[k] + x[k] : fmaxf(acc[k], x[k]);
                } else {
                    const bf16x8 x = *(const bf16x8*)q;
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = AVG ? acc[k] + (float)x[k] : fmaxf(acc[k], (float)x[k]);
                }
            }
        }
        T* o = out + (((long)n * Ho + ho) * Wo + wo) * ldo + cc * V;
        if constexpr (sizeof(T) == 4) {
            f32x4 y;
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = AVG ? acc[k] * (1.0f / 9.0f) : acc[k];
            *(f32x4*)o = y;
        } else {
            bf16x8 y;
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = (bf16_t)(AVG ? acc[k] * (1.0f / 9.0f) : acc[k]);
            *(bf16x8*)o = y;
        }
    }
}

// out[r*ldo + c] = in[idx(r)*ldi + c]: row gather / strided row copy (idx NULL: identity)
__global__ __launch_bounds__(256) void rows_copy_kernel(const float* __restrict__ in, long ldi, const int64_t* __restrict__ idx,
                                                        long idx_stride, long nrows_in, int cols, float* __restrict__ out, long ldo) {
    const int r = blockIdx.x;
    long src = r;
    if (idx) {
        src = idx[(long)r * idx_stride];
        src = src < 0 ? 0 : (src >= nrows_in ? nrows_in - 1 : src);      // memory safety only (sat_validate_ids reports bad ids)
    }
    for (int c = threadIdx.x; c < cols; c += blockDim.x) out[(long)r * ldo + c] = in[src * ldi + c];
}

// out[c] = sum_r in[r*ld + c] over rows (small r): fixed order
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ in, long ld, int rows, int cols, float* __restrict__ out,
                                                       int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float acc = accumulate ? out[c] : 0.0f;
    for (int r = 0; r < rows; ++r) acc += in[(long)r * ld + c];
    out[c] = acc;
}

// out[r][c] = a[r][c] + b[r][c] (strided rows)
__global__ __launch_bounds__(256) void rows_add_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                       int cols, float* __restrict__ out, long ldo) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) out[(long)r * ldo + c] = a[(long)r * lda + c] + b[(long)r * ldb + c];
}

// packed token ids: out[row(t,b)] = captions[b][t + col0] for b < batch_sizes[t]
__global__ void pack_tokens_kernel(const int64_t* __restrict__ captions, long cap_stride, const int* __restrict__ prefix, int T,
                                   int N, int col0, int64_t* __restrict__ out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    int t = 0;
    while (t + 1 < T && row >= prefix[t + 1]) ++t;
    out[row] = captions[(long)(row - prefix[t]) * cap_stride + t + col0];
}

// deterministic scatter-add of rows into a table (dense embedding gradient, nn.Embedding sparse=False): the workgroup of
// the FIRST row carrying an id sums every row with that id in row order; the table is zeroed by the launcher.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ rows, const int64_t* __restrict__ ids, int N,
                                                           int E, int V, float* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) int tok[];          // [N] ids | [ceil(N/32)] match bits of the later rows
    const int row = blockIdx.x;
    const int nw = (N + 31) >> 5;
    unsigned* bits = (unsigned*)(tok + N);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const long x = ids[i];
        tok[i] = (int)(x < 0 ? 0 : (x >= V ? V - 1 : x));            // memory safety only (sat_validate_ids reports bad ids)
    }
    for (int i = threadIdx.x; i < nw; i += blockDim.x) bits[i] = 0u;
    __syncthreads();
    const int v = tok[row];
    int dup = 0;
    for (int i = threadIdx.x; i < row; i += blockDim.x) dup |= (tok[i] == v);
    if (__syncthreads_or(dup)) return;
    // later rows with the same id, as a bit set (built in any order, walked in row order: the sum's order is fixed)
    for (int i = row + 1 + threadIdx.x; i < N; i += blockDim.x)
        if (tok[i] == v) atomicOr(&bits[i >> 5], 1u << (i & 31));
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float acc = rows[(long)row * E + e];
        for (int w = (row + 1) >> 5; w < nw; ++w) {
            unsigned m = bits[w];
            while (m) {
                const int i = (w << 5) + __ffs(m) - 1;
                m &= m - 1;
                acc += rows[(long)i * E + e];
            }
        }
        table[(long)v * E + e] = acc;
    }
}

}  // namespace

// ---- conv-stack backward helpers (fine-tuning, model2.py:87-89 `finetune(allow=True)`), f32 NHWC ----
namespace {
// out = zero-bordered copy of in (border `pad` pixels); with `y` != NULL the ReLU mask is applied on the way:
// out[n][h+pad][w+pad][c] = y[n][h][w][c] > 0 ? in[n][h][w][c] : 0      (d pre-activation = d post-activation * (y > 0))
__global__ void pad_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ y, int N, int H, int W, int C, int pad,
                                float* __restrict__ out) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad, c4 = C / 4;
    const long total = (long)N * Hp * Wp * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4);
        long r = i / c4;
        const int wp = (int)(r % Wp); r /= Wp;
        const int hp = (int)(r % Hp);
        const int n = (int)(r / Hp);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int h = hp - pad, w = wp - pad;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
            const long src = (((long)n * H + h) * W + w) * C + cc * 4;
            v = *(const f32x4*)(in + src);
            if (y) {
                const f32x4 m = *(const f32x4*)(y + src);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.0f ? v[e] : 0.0f;
            }
        }
        *(f32x4*)(out + i * 4) = v;
    }
}
// backward of the 2x2/2 max-pool: the gradient goes to the FIRST maximum of each window in scan order (torch's rule)
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N, int Hin, int Win, int C,
                                    float* __restrict__ dx) {
    const int Ho = Hin / 2, Wo = Win / 2;
    const long total = (long)N * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const long base = (((long)n * Hin + 2 * ho) * Win + 2 * wo) * C + c;
        const long off[4] = {0, (long)C, (long)Win * C, (long)Win * C + C};
        int best = 0;
        float bv = x[base];
#pragma unroll
        for (int d = 1; d < 4; ++d) {
            const float v = x[base + off[d]];
            if (v > bv) { bv = v; best = d; }
        }
        const float g = dy[i];
#pragma unroll
        for (int d = 0; d < 4; ++d) dx[base + off[d]] = d == best ? g : 0.0f;
    }
}
// out[b][p][c] += scale * v[b][c]   (mean over p in the forward)
__global__ void bcast_add_kernel(const float* __restrict__ v, int B, int P, int C, float scale, float* __restrict__ out) {
    const long total = (long)B * P * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((long)P * C));
        out[i] += scale * v[(long)b * C + c];
    }
}
inline int ew_grid2(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }
}  // namespace

extern "C" int sat_pad_nhwc_f32(const float* in, const float* relu_mask_of, int N, int H, int W, int C, int pad, float* out,
                                sat_stream_t stream) {
    if (!in || !out || N < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || pad < 0 || in == out) return SAT_ERR_ARG;
    const long total = (long)N * (H + 2 * pad) * (W + 2 * pad) * (C / 4);
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(ew_grid2(total)), dim3(256), 0, (hipStream_t)stream, in, relu_mask_of, N, H, W, C, pad, out);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_maxpool2_bwd_f32(const float* x, const float* dy, int N, int Hin, int Win, int C, float* dx, sat_stream_t stream) {
    if (!x || !dy || !dx || N < 1 || (Hin & 1) || (Win & 1) || C < 1) return SAT_ERR_ARG;
    const long total = (long)N * (Hin / 2) * (Win / 2) * C;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid2(total)), dim3(256), 0, (hipStream_t)stream, x, dy, N, Hin, Win, C, dx);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_bcast_add_f32(const float* v, int B, int P, int C, float scale, float* out, sat_stream_t stream) {
    if (!v || !out || B < 1 || P < 1 || C < 1) return SAT_ERR_ARG;
    hipLaunchKernelGGL(bcast_add_kernel, dim3(ew_grid2((long)B * P * C)), dim3(256), 0, (hipStream_t)stream, v, B, P, C, scale, out);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// On-device collate (data_loader.py:48-62 `collate_fn`): captions arrive ragged (one flat id array + per-sample offsets) in
// dataset order; `order[r]` = the sample that lands in row r (sorted by decreasing length, ties in dataset order -- computed
// on the host from the host-side lengths).  out[r][t] = t < len(order[r]) ? flat[offset[order[r]] + t] : 0.
namespace {
__global__ __launch_bounds__(256) void collate_captions_kernel(const int64_t* __restrict__ flat, const int64_t* __restrict__ offsets,
                                                               const int32_t* __restrict__ order, int B, int Tmax,
                                                               int64_t* __restrict__ out) {
    const int r = blockIdx.x;
    const int src = order[r];
    const int64_t o0 = offsets[src], len = offsets[src + 1] - o0;
    for (int t = threadIdx.x; t < Tmax; t += blockDim.x) out[(long)r * Tmax + t] = t < len ? flat[o0 + t] : 0;
}
// out row r = in row order[r]; rows of `cols` floats, 16 B per lane when aligned
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ order, long cols,
                                                          float* __restrict__ out) {
    const long r = blockIdx.y;
    const float* src = in + (long)order[r] * cols;
    float* dst = out + r * cols;
    const long n4 = ((cols & 3) == 0 && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0) ? cols / 4 : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        ((f32x4*)dst)[i] = ((const f32x4*)src)[i];
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < cols; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace

extern "C" int sat_collate_captions(const int64_t* flat, const int64_t* offsets, const int32_t* order, int B, int Tmax,
                                    int64_t* out, sat_stream_t stream) {
    if (!flat || !offsets || !order || !out || B < 1 || Tmax < 1) return SAT_ERR_ARG;
    hipLaunchKernelGGL(collate_captions_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, flat, offsets, order, B, Tmax, out);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_gather_rows_f32(const float* in, const int32_t* order, int rows, int64_t cols, float* out, sat_stream_t stream) {
    if (!in || !order || !out || rows < 1 || cols < 1 || in == out) return SAT_ERR_ARG;
    const long per = (cols / 4 + 255) / 256;
    const int gx = (int)(per < 1 ? 1 : (per > 64 ? 64 : per));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, rows), dim3(256), 0, (hipStream_t)stream, in, order, (long)cols, out);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_rows_add(const float* a, int64_t lda, const float* b, int64_t ldb, int rows, int cols, float* out, int64_t ldo,
                            sat_stream_t stream) {
    if (!a || !b || !out || rows < 0 || cols < 1 || lda < cols || ldb < cols || ldo < cols) return SAT_ERR_ARG;
    if (rows == 0) return SAT_OK;
    hipLaunchKernelGGL(rows_add_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, (long)lda, b, (long)ldb, cols, out, (long)ldo);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_pack_tokens(const int64_t* captions, int64_t cap_stride, const int32_t* prefix, int T, int N, int col0,
                               int64_t* out, sat_stream_t stream) {
    if (!captions || !prefix || !out || T < 1 || N < 1 || col0 < 0) return SAT_ERR_ARG;
    hipLaunchKernelGGL(pack_tokens_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, captions, (long)cap_stride,
                       prefix, T, N, col0, out);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_scatter_rows_add(const float* rows, const int64_t* ids, int N, int E, int V, float* table, sat_stream_t stream) {
    if (!rows || !ids || !table || N < 1 || E < 1 || V < 1) return SAT_ERR_ARG;
    const size_t dyn = ((size_t)N + (size_t)((N + 31) / 32)) * 4;
    if (dyn > 150 * 1024) return SAT_ERR_UNSUPPORTED;
    if (dyn > 48 * 1024) {
        hipError_t ea = hipFuncSetAttribute((const void*)scatter_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (ea != hipSuccess) return (int)ea;
    }
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(table, 0, (size_t)V * E * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(N), dim3(256), dyn, s, rows, ids, N, E, V, table);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_lstmcell_bwd_point(const float* dh_out, const float* dh_carry, int n_carry, const float* gates, const float* c,
                                      const float* c_prev, float* dc_state, float* DG, int n, int H, sat_stream_t stream) {
    if (!dh_out || !gates || !c || !dc_state || !DG || n < 1 || H < 1 || (n_carry > 0 && !dh_carry)) return SAT_ERR_ARG;
    return sat_lstm_bwd_point_launch(dh_out, dh_carry, n_carry > 0 ? 1 : 0, 0, n_carry, gates, c, c_prev, dc_state, DG, n, H,
                                     (hipStream_t)stream);
}

int sat_pool3_launch(const sat_op* op, bool avg, hipStream_t s) {
    if (!op->in0 || !op->out || op->in0 == op->out) return SAT_ERR_ARG;
    const int C = op->Cout;
    const int V = op->dtype == SAT_BF16 ? 8 : 4;
    const long ldo = op->ldc ? op->ldc : C;
    if ((C % V) || (ldo % V) || ldo < C) return SAT_ERR_ARG;
    const int Ho = avg ? op->Hin : (op->Hin - 3) / 2 + 1, Wo = avg ? op->Win : (op->Win - 3) / 2 + 1;
    if (Ho < 1 || Wo < 1 || (op->Hout && (op->Hout != Ho || op->Wout != Wo))) return SAT_ERR_ARG;
    const long total = (long)op->N * Ho * Wo * (C / V);
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipStream_t st = s;
    if (op->dtype == SAT_BF16) {
        if (avg) hipLaunchKernelGGL((pool3_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)op->in0, (bf16_t*)op->out, op->N, op->Hin, op->Win, C, Ho, Wo, ldo);
        else hipLaunchKernelGGL((pool3_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)op->in0, (bf16_t*)op->out, op->N, op->Hin, op->Win, C, Ho, Wo, ldo);
    } else {
        if (avg) hipLaunchKernelGGL((pool3_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)op->in0, (float*)op->out, op->N, op->Hin, op->Win, C, Ho, Wo, ldo);
        else hipLaunchKernelGGL((pool3_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)op->in0, (float*)op->out, op->N, op->Hin, op->Win, C, Ho, Wo, ldo);
    }
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

int sat_maxpool2_launch(const sat_op* op, hipStream_t s) {
    if (!op->in0 || !op->out || (op->Hin & 1) || (op->Win & 1)) return SAT_ERR_ARG;
    const int C = op->Cout;
    const int V = op->dtype == SAT_BF16 ? 8 : 4;
    if (C % V) return SAT_ERR_ARG;
    const long total = (long)op->N * (op->Hin / 2) * (op->Win / 2) * (C / V);
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    if (op->dtype == SAT_BF16)
        hipLaunchKernelGGL(maxpool2_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)op->in0, (bf16_t*)op->out, op->N, op->Hin, op->Win, C);
    else
        hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)op->in0, (float*)op->out, op->N, op->Hin, op->Win, C);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

static int att_pchunk(int rows, int P) {
    int nch = 512 / rows;                                    // aim at >= 512 workgroups
    if (nch < 1) nch = 1;
    const int maxch = sat_cdiv(P, 4);                         // at least one position per wave
    if (nch > maxch) nch = maxch;
    return sat_cdiv(P, nch);
}

extern "C" int64_t sat_attention_ws_bytes(int rows, int P) { return (int64_t)rows * P * (int64_t)sizeof(float); }

extern "C" int sat_attention_fwd(const float* ctx_enc, const float* feats, const float* proj, int64_t ld_proj, const float* w_att,
                                 int rows, int P, int C, float* alpha, float* context, int64_t ld_ctx, float* workspace,
                                 int64_t ws_bytes, sat_stream_t stream) {
    if (!ctx_enc || !feats || !proj || !w_att || !context || rows < 1 || P < 1 || C < 4 || (C & 3) || ld_proj < C || ld_ctx < C)
        return SAT_ERR_ARG;
    if (!workspace || ws_bytes < sat_attention_ws_bytes(rows, P)) return SAT_ERR_WORKSPACE;
    const size_t lds1 = (size_t)2 * C * sizeof(float);
    const size_t lds2 = (size_t)(((P + 3) & ~3) + 8 + kAttWaves * 64) * sizeof(float);
    if (lds1 > 60 * 1024 || lds2 > 60 * 1024) return SAT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int pch = att_pchunk(rows, P);
    hipLaunchKernelGGL(att_rowdot_kernel<true>, dim3(rows, sat_cdiv(P, pch)), dim3(256), lds1, s, ctx_enc, proj, (long)ld_proj, w_att,
                       0L, (const float*)nullptr, 0L, 1.0f, P, C, pch, workspace);
    SAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(att_context_kernel, dim3(rows, sat_cdiv(C, 64)), dim3(kAttWaves * 64), lds2, s, workspace, feats, P, C, alpha,
                       context, (long)ld_ctx);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_attention_bwd(const float* ctx_enc, const float* feats, const float* proj, int64_t ld_proj, const float* w_att,
                                 const float* alpha, const float* d_ctx, int64_t ld_dctx, const float* d_ctx2, int64_t ld_dctx2,
                                 int rows, int P, int C, float* d_ctx_enc, float* d_proj, float* d_watt_part, float* d_feats,
                                 float* workspace, int64_t ws_bytes, sat_stream_t stream) {
    if (!ctx_enc || !feats || !proj || !w_att || !alpha || !d_ctx || !d_ctx_enc || !d_proj || !d_watt_part || rows < 1 || P < 1 ||
        C < 4 || (C & 3) || ld_proj < C || ld_dctx < C || (d_ctx2 && ld_dctx2 < C))
        return SAT_ERR_ARG;
    if (!workspace || ws_bytes < sat_attention_ws_bytes(rows, P)) return SAT_ERR_WORKSPACE;
    const size_t lds1 = (size_t)2 * C * sizeof(float);
    const size_t lds2 = (size_t)(2 * ((P + 3) & ~3) + 8 + 2 * kAttWaves * 64) * sizeof(float);
    if (lds1 > 60 * 1024 || lds2 > 60 * 1024) return SAT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int pch = att_pchunk(rows, P);
    hipLaunchKernelGGL(att_rowdot_kernel<false>, dim3(rows, sat_cdiv(P, pch)), dim3(256), lds1, s, feats, (const float*)nullptr, 0L,
                       d_ctx, (long)ld_dctx, d_ctx2, (long)ld_dctx2, 1.0f / (float)P, P, C, pch, workspace);
    SAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(att_bwd_channel_kernel, dim3(rows, sat_cdiv(C, 64)), dim3(kAttWaves * 64), lds2, s, ctx_enc, proj, (long)ld_proj,
                       w_att, alpha, workspace, d_ctx, (long)ld_dctx, d_ctx2, (long)ld_dctx2, P, C, d_ctx_enc, d_proj, d_watt_part,
                       d_feats);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_rows_copy(const float* in, int64_t ldi, const int64_t* idx, int64_t idx_stride, int64_t nrows_in, int rows,
                             int cols, float* out, int64_t ldo, sat_stream_t stream) {
    if (!in || !out || rows < 0 || cols < 1 || ldi < cols || ldo < cols) return SAT_ERR_ARG;
    if (rows == 0) return SAT_OK;
    hipLaunchKernelGGL(rows_copy_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, in, (long)ldi, idx, (long)idx_stride,
                       (long)nrows_in, cols, out, (long)ldo);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_rows_sum(const float* in, int64_t ld, int rows, int cols, float* out, int accumulate, sat_stream_t stream) {
    if (!in || !out || rows < 0 || cols < 1 || ld < cols) return SAT_ERR_ARG;
    hipLaunchKernelGGL(rows_sum_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, (long)ld, rows, cols, out,
                       accumulate);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}
