// HBM-bound kernels of the Show-and-Tell hot path: every access is a 16-byte-per-lane coalesced stream
// (NHWC activations, flat parameter buffers), reductions are fixed-order (bitwise reproducible), no atomics.
//   encoder : image prep, batch-norm finalize / apply (+ReLU, +residual, +maxpool), global average pool
//   decoder : embedding gather / deterministic scatter, row-wise softmax-CE, column sums, LSTM backward
//             pointwise step, BatchNorm1d forward/backward
//   trainer : fused elementwise clamp + Adam over one flat buffer (train.py:88-91,146)
#include "sat_internal.h"
#include <stdlib.h>

namespace {

constexpr int EW_BLOCK = 256;
inline int ew_grid(long n_items) {
    long cap;
    // 768 workgroups (3 per CU), measured under the encoder look-ahead (tools/run_gpu_bn_ab.sh, round 3): with three stacks in
    // flight the HBM-bound BatchNorm passes of one stack run beside the convs of the others, and a grid that takes fewer wave
    // slots per CU leaves those convs their CUs: 2048 / 1024 / 768 / 512 / 384 = 13.55 / 13.8 / 14.0 / 13.95 k img/s (with the
    // non-temporal operand loads 13.75 / 14.2 / 14.27 / 14.3 / 14.3 k); the sequential step does not move until 384 (6.36 -> 6.49 ms)
    cap = 768;
    long b = (n_items + EW_BLOCK - 1) / EW_BLOCK;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));   // cap and grid-stride the rest
}

template <typename T> struct Vec;   // 16-byte vector of T
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<bf16_t> { static constexpr int N = 8; };

template <typename T> __device__ __forceinline__ void load_chunk(const T* p, float (&v)[Vec<T>::N]);
template <> __device__ __forceinline__ void load_chunk<float>(const float* p, float (&v)[4]) {
    const f32x4 x = *(const f32x4*)p;
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}
template <> __device__ __forceinline__ void load_chunk<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const bf16x8 x = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
}
// raw 16-byte load now, conversion later: lets a kernel put loads in flight before work they do not depend on
template <typename T> __device__ __forceinline__ u32x4 load_raw(const T* p) { return *(const u32x4*)p; }
// NT: non-temporal (streaming) load -- for an operand no later kernel reads again
template <typename T, bool NT> __device__ __forceinline__ u32x4 load_in(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load((const u32x4*)p);
    else return *(const u32x4*)p;
}
template <typename T> __device__ __forceinline__ void unpack_chunk(const u32x4& r, float (&v)[Vec<T>::N]);
template <> __device__ __forceinline__ void unpack_chunk<float>(const u32x4& r, float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(r[i]);
}
template <> __device__ __forceinline__ void unpack_chunk<bf16_t>(const u32x4& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // bf16 -> f32 is a 16-bit shift
        v[2 * i] = __uint_as_float(r[i] << 16);
        v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
    }
}
template <typename T> __device__ __forceinline__ void store_chunk(T* p, const float (&v)[Vec<T>::N]);
template <> __device__ __forceinline__ void store_chunk<float>(float* p, const float (&v)[4]) {
    f32x4 x = {v[0], v[1], v[2], v[3]};
    *(f32x4*)p = x;
}
template <> __device__ __forceinline__ void store_chunk<bf16_t>(bf16_t* p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16_t)v[i];
    store16_wt(p, __builtin_bit_cast(u32x4, x));     // bulk activations the next launch reads: write-through (sat_common.h)
}

// ------------------------------------------------------------------------------------------------------
// image prep: NCHW f32 -> zero-bordered NHWC4 (channel 3 = 0).  Border/extra pixels are never written:
// the caller zero-fills the buffer once.
template <typename T, int CH = 4>
__global__ void image_prep_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int H, int W,
                                  int Hp, int Wp, int pad) {
    const long total = (long)N * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const long t = i / W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const long plane = (long)H * W;
        const float* src = in + (long)n * 3 * plane + (long)h * W + w;
        T* dst = out + (((long)n * Hp + h + pad) * Wp + w + pad) * CH;
        dst[0] = from_f32<T>(src[0]);
        dst[1] = from_f32<T>(src[plane]);
        dst[2] = from_f32<T>(src[2 * plane]);
#pragma unroll
        for (int c = 3; c < CH; ++c) dst[c] = from_f32<T>(0.0f);
    }
}

// ------------------------------------------------------------------------------------------------------
// batch-norm finalize: per-tile partial (sum, sumsq) -> scale/shift; running-stat update (momentum).
// block = 32 channels x 32 partial groups (1024 threads: the kernel is pure load latency, so go wide);
// f64 accumulation, fixed order.
constexpr int FIN_G = 32;
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int tiles_m, int C,
                                                           double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* running_mean,
                                                           float* running_var, float momentum, float eps,
                                                           int training, float* scale, float* shift) {
    __shared__ double ss[FIN_G][32], sq[FIN_G][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0, q = 0.0;
    if (training && c < C) {
        int t = rg;
        for (; t + 3 * FIN_G < tiles_m; t += 4 * FIN_G) {      // 8 independent loads in flight per thread
            const float a0 = partial[((long)t * 2 + 0) * C + c], b0 = partial[((long)t * 2 + 1) * C + c];
            const float a1 = partial[((long)(t + FIN_G) * 2 + 0) * C + c], b1 = partial[((long)(t + FIN_G) * 2 + 1) * C + c];
            const float a2 = partial[((long)(t + 2 * FIN_G) * 2 + 0) * C + c], b2 = partial[((long)(t + 2 * FIN_G) * 2 + 1) * C + c];
            const float a3 = partial[((long)(t + 3 * FIN_G) * 2 + 0) * C + c], b3 = partial[((long)(t + 3 * FIN_G) * 2 + 1) * C + c];
            s += (double)a0; s += (double)a1; s += (double)a2; s += (double)a3;
            q += (double)b0; q += (double)b1; q += (double)b2; q += (double)b3;
        }
        for (; t < tiles_m; t += FIN_G) {
            s += (double)partial[((long)t * 2 + 0) * C + c];
            q += (double)partial[((long)t * 2 + 1) * C + c];
        }
    }
    ss[rg][cl] = s; sq[rg][cl] = q;
    __syncthreads();
    if (rg == 0 && c < C) {
        double mean, var;
        if (training) {
            double S = 0.0, Q = 0.0;
#pragma unroll
            for (int g = 0; g < FIN_G; ++g) { S += ss[g][cl]; Q += sq[g][cl]; }
            mean = S / count;
            var = Q / count - mean * mean;
            if (var < 0.0) var = 0.0;
            if (running_mean) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * (double)(float)mean);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (double)(float)unbiased);
            }
        } else {
            mean = running_mean[c];
            var = running_var[c];
        }
        const float invstd = 1.0f / sqrtf((float)var + eps);
        const float sc = gamma[c] * invstd;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
    }
}

// per-tile slabs -> fixed-point integer accumulators: workgroup (x, y) sums tiles [128y, 128y+128) of channels
// [32x, 32x+32) in f64 (fixed order) and adds the two totals with 64-bit INTEGER atomics (order-independent)
constexpr int SLAB_TILES_PER_WG = 128;
__global__ __launch_bounds__(1024) void bn_slab_to_acc_kernel(const float* __restrict__ partial, int tiles_m, int C,
                                                              long long* __restrict__ acc) {
    __shared__ double ss[FIN_G][32], sq[FIN_G][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    partial += (long)blockIdx.z * tiles_m * 2 * C;           // grouped program: group = blockIdx.z, [G][tiles_m][2][C] slabs,
    acc += (long)blockIdx.z * 4 * C;                         // [G][2 parities][2][C] accumulators
    const int t0 = blockIdx.y * SLAB_TILES_PER_WG;
    const int t1 = (t0 + SLAB_TILES_PER_WG < tiles_m) ? t0 + SLAB_TILES_PER_WG : tiles_m;
    double s = 0.0, q = 0.0;
    if (c < C) {
        float a[SLAB_TILES_PER_WG / FIN_G], b[SLAB_TILES_PER_WG / FIN_G];
#pragma unroll
        for (int u = 0; u < SLAB_TILES_PER_WG / FIN_G; ++u) {        // all of a thread's loads in flight together
            const int t = t0 + rg + u * FIN_G;
            a[u] = t < t1 ? partial[((long)t * 2 + 0) * C + c] : 0.0f;
            b[u] = t < t1 ? partial[((long)t * 2 + 1) * C + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < SLAB_TILES_PER_WG / FIN_G; ++u) {
            s += (double)a[u];
            q += (double)b[u];
        }
    }
    ss[rg][cl] = s;
    sq[rg][cl] = q;
    __syncthreads();
    if (rg == 0 && c < C) {
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int g = 0; g < FIN_G; ++g) {
            S += ss[g][cl];
            Q += sq[g][cl];
        }
        atomicAdd((unsigned long long*)(acc + c), (unsigned long long)__double2ll_rn(S * SAT_STAT_SCALE));
        atomicAdd((unsigned long long*)(acc + C + c), (unsigned long long)__double2ll_rn(Q * SAT_STAT_SCALE));
    }
}

// eval mode: (scale, shift) of EVERY BatchNorm of the stack from its running statistics, one workgroup per layer
// (same float arithmetic as bn_finalize_kernel's eval branch)
__global__ __launch_bounds__(256) void bn_eval_batch_kernel(const sat_bn_eval_item* __restrict__ items, float eps) {
    const sat_bn_eval_item it = items[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += 256) {
        const float invstd = 1.0f / sqrtf(it.running_var[c] + eps);
        const float sc = it.gamma[c] * invstd;
        it.scale_out[c] = sc;
        it.shift_out[c] = it.beta[c] - it.running_mean[c] * sc;
    }
}

// Deferred running-statistics update: a program run with momentum 1 into private buffers leaves every layer's batch (mean,
// unbiased var) there as f32; this applies them to the model's running statistics with the real momentum -- the same expression
// on the same f32 inputs as the in-kernel update, so bit-identical to it -- one workgroup per layer, in the caller's batch order.
__global__ __launch_bounds__(256) void bn_running_apply_kernel(const sat_bn_running_item* __restrict__ items, float momentum) {
    const sat_bn_running_item it = items[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += 256) {
        it.running_mean[c] = (float)((1.0 - momentum) * it.running_mean[c] + momentum * (double)it.batch_mean[c]);
        it.running_var[c] = (float)((1.0 - momentum) * it.running_var[c] + momentum * (double)it.batch_var[c]);
    }
}

// Where a BatchNorm's (scale, shift) comes from: either a table the finalize kernel wrote, or -- `acc` set -- the
// fixed-point integer sums the producing conv accumulated (sat_conv_glds.hip): then every workgroup derives the
// table itself into LDS (a few KB of loads, f64 arithmetic identical to bn_finalize_kernel), and workgroup 0 also
// updates the running statistics and clears the OTHER step-parity's accumulators for the next step.
struct BnSrc {
    const float* scale;
    const float* shift;
    const long long* acc;     // [2][C] (this step's parity)
    long long* acc_clear;     // [2][C] (other parity) or NULL
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
};

// grouped program (sat_op.groups): group g's statistics live g blocks further on -- accumulators [G][2 parities][2][C], the
// deferred running-statistics log [G][2][C]
__device__ __forceinline__ BnSrc bn_group(BnSrc b, long g, int C) {
    if (g) {
        if (b.acc) b.acc += g * 4 * C;
        if (b.acc_clear) b.acc_clear += g * 4 * C;
        if (b.running_mean) { b.running_mean += g * 2 * C; b.running_var += g * 2 * C; }
    }
    return b;
}

__device__ __forceinline__ void bn_table_from_acc(const BnSrc& b, int C, double count, float momentum, float eps,
                                                  float* sc, float* sh) {
    const double inv = 1.0 / (SAT_STAT_SCALE * count);      // one f64 division per thread, none per channel
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const long long s1 = b.acc[c], s2 = b.acc[C + c];
        const double mean = (double)s1 * inv;
        double var = (double)s2 * inv - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = 1.0f / sqrtf((float)var + eps);
        const float s = b.gamma[c] * invstd;
        sc[c] = s;
        sh[c] = b.beta[c] - (float)mean * s;
        if (blockIdx.x == 0) {
            if (b.running_mean) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                b.running_mean[c] = (float)((1.0 - momentum) * b.running_mean[c] + momentum * (double)(float)mean);
                b.running_var[c] = (float)((1.0 - momentum) * b.running_var[c] + momentum * (double)(float)unbiased);
            }
            if (b.acc_clear) { b.acc_clear[c] = 0; b.acc_clear[C + c] = 0; }
        }
    }
}

// out = relu(in0*s0 + t0)  /  out = relu(in0*s0 + t0 + (in1*s1 + t1 | in1)); NHWC, C % chunk == 0.
// The launcher makes the total thread count a multiple of the chunks per pixel, so a thread's channel chunk never
// changes along its grid-stride walk: scale/shift live in registers and the loop is pure 16-byte streaming.
// `out` may alias `in0` (in-place normalise: every chunk is read before it is written, by the thread that writes it).
template <typename T, bool ADD, bool NT = false>
__global__ void bn_act_kernel(const T* in0, const T* __restrict__ in1, T* out,
                              const BnSrc b0_, const BnSrc b1_, int has_b1, double count, float momentum, float eps,
                              long nchunks, int C) {
    constexpr int V = Vec<T>::N;
    // grouped program: group = blockIdx.y works on its own batch (nchunks chunks further on) with its own statistics
    const long grp = blockIdx.y;
    in0 += grp * nchunks * V;
    if constexpr (ADD) in1 += grp * nchunks * V;
    out += grp * nchunks * V;
    const BnSrc b0 = bn_group(b0_, grp, C), b1 = bn_group(b1_, grp, C);
    extern __shared__ __attribute__((aligned(16))) float tab[];      // [4][C] when a table is derived here
    const float* s0 = b0.scale;
    const float* t0 = b0.shift;
    const float* s1 = b1.scale;
    const float* t1 = b1.shift;
    const long stride = (long)gridDim.x * blockDim.x;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cch = C / V;
    // 32-bit index math in the prologue (a 64-bit modulo costs more than the whole streaming loop of a small tensor)
    const bool fast = ((unsigned)stride % (unsigned)cch) == 0;
    constexpr int U = ADD ? 2 : 4;             // 4 independent 16-byte loads per lane and stage
    u32x4 rx[U], rz[U];
    if (fast) {     // first stage of the stream goes in flight BEFORE the table is derived: it does not depend on it
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long iu = i0 + u * stride;
            if (iu < nchunks) {
                rx[u] = load_in<T, NT>(in0 + iu * V);
                if constexpr (ADD) rz[u] = load_in<T, NT>(in1 + iu * V);
            }
        }
    }
    bool derived = false;
    if (b0.acc) {
        bn_table_from_acc(b0, C, count, momentum, eps, tab, tab + C);
        s0 = tab; t0 = tab + C; derived = true;
    }
    if (ADD && has_b1 && b1.acc) {
        bn_table_from_acc(b1, C, count, momentum, eps, tab + 2 * C, tab + 3 * C);
        s1 = tab + 2 * C; t1 = tab + 3 * C; derived = true;
    }
    if (derived) __syncthreads();
    if (fast) {
        const int c0 = (int)((unsigned)i0 % (unsigned)cch) * V;
        float sc0[V], sh0[V], sc1[V], sh1[V];
#pragma unroll
        for (int k = 0; k < V; k += 4) {
            const f32x4 a = *(const f32x4*)(s0 + c0 + k), b = *(const f32x4*)(t0 + c0 + k);
            f32x4 c = {1.f, 1.f, 1.f, 1.f}, d = {0.f, 0.f, 0.f, 0.f};
            if (ADD && has_b1) { c = *(const f32x4*)(s1 + c0 + k); d = *(const f32x4*)(t1 + c0 + k); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc0[k + e] = a[e]; sh0[k + e] = b[e]; sc1[k + e] = c[e]; sh1[k + e] = d[e]; }
        }
        for (long i = i0; i < nchunks;) {
            const long inext = i + stride * U;
            u32x4 nx[U], nz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {          // next stage in flight while this one is computed and stored
                const long iu = inext + u * stride;
                if (iu < nchunks) {
                    nx[u] = load_in<T, NT>(in0 + iu * V);
                    if constexpr (ADD) nz[u] = load_in<T, NT>(in1 + iu * V);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long iu = i + u * stride;
                if (iu < nchunks) {
                    float x[V], y[V];
                    unpack_chunk<T>(rx[u], x);
                    if constexpr (ADD) {
                        float z[V];
                        unpack_chunk<T>(rz[u], z);
#pragma unroll
                        for (int k = 0; k < V; ++k) y[k] = fmaxf(x[k] * sc0[k] + sh0[k] + (z[k] * sc1[k] + sh1[k]), 0.0f);
                    } else {
#pragma unroll
                        for (int k = 0; k < V; ++k) y[k] = fmaxf(x[k] * sc0[k] + sh0[k], 0.0f);
                    }
                    store_chunk<T>(out + iu * V, y);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rx[u] = nx[u];
                if constexpr (ADD) rz[u] = nz[u];
            }
            i = inext;
        }
        return;
    }
    for (long i = i0; i < nchunks; i += stride) {
        const int c0 = (int)((i * V) % C);
        float x[V], y[V];
        load_chunk<T>(in0 + i * V, x);
#pragma unroll
        for (int k = 0; k < V; ++k) y[k] = x[k] * s0[c0 + k] + t0[c0 + k];
        if constexpr (ADD) {
            float z[V];
            load_chunk<T>(in1 + i * V, z);
            if (has_b1) {
#pragma unroll
                for (int k = 0; k < V; ++k) y[k] += z[k] * s1[c0 + k] + t1[c0 + k];
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) y[k] += z[k];
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) y[k] = fmaxf(y[k], 0.0f);
        store_chunk<T>(out + i * V, y);
    }
}

// out[pixel * ldo + c] = relu(in[pixel * C + c] * s[c] + t[c]): the activation of a conv lands in a channel slice of a wider
// (concatenated) NHWC tensor -- Inception blocks (BASELINE configs[3]).  One thread per 16-byte chunk.
template <typename T>
__global__ void bn_relu_strided_kernel(const T* __restrict__ in, T* __restrict__ out, long ldo, const BnSrc b_, double count,
                                       float momentum, float eps, long npix, int C) {
    constexpr int V = Vec<T>::N;
    const long grp = blockIdx.y;                             // grouped program: npix = the pixels of ONE group's batch
    in += grp * npix * C;
    out += grp * npix * ldo;
    const BnSrc b = bn_group(b_, grp, C);
    extern __shared__ __attribute__((aligned(16))) float tab[];
    const float* s = b.scale;
    const float* t = b.shift;
    if (b.acc) {
        bn_table_from_acc(b, C, count, momentum, eps, tab, tab + C);
        __syncthreads();
        s = tab;
        t = tab + C;
    }
    const int cch = C / V;
    const long total = npix * cch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cch;
        const int c0 = (int)(i - pix * cch) * V;
        float x[V], y[V];
        load_chunk<T>(in + pix * C + c0, x);
#pragma unroll
        for (int k = 0; k < V; ++k) y[k] = fmaxf(x[k] * s[c0 + k] + t[c0 + k], 0.0f);
        store_chunk<T>(out + pix * ldo + c0, y);
    }
}

// out[n][ho][wo][c] = max_{3x3, stride 2, pad 1} relu(in*s + t)
template <typename T>
__global__ void bn_relu_maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, const BnSrc b_, double count,
                                       float momentum, float eps, int N, int Hin, int Win, int C, int Hout, int Wout) {
    constexpr int V = Vec<T>::N;
    const long grp = blockIdx.y;                             // grouped program: group = blockIdx.y, N = the batch of ONE group
    in += grp * N * Hin * Win * C;
    out += grp * N * Hout * Wout * C;
    const BnSrc b = bn_group(b_, grp, C);
    extern __shared__ __attribute__((aligned(16))) float tab[];      // [2][C] when the table is derived here
    const float* s = b.scale;
    const float* t = b.shift;
    if (b.acc) {                // statistics arrive as integer sums: derive (scale, shift) like bn_act_kernel does
        bn_table_from_acc(b, C, count, momentum, eps, tab, tab + C);
        __syncthreads();
        s = tab;
        t = tab + C;
    }
    const int cch = C / V;
    const long total = (long)N * Hout * Wout * cch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // 32-bit index math (the launcher guarantees total < 2^31): 64-bit div/mod would cost more than the 9 loads
        const unsigned iu = (unsigned)i;
        const unsigned cc = iu % (unsigned)cch;
        unsigned r = iu / (unsigned)cch;
        const int wo = (int)(r % (unsigned)Wout); r /= (unsigned)Wout;
        const int ho = (int)(r % (unsigned)Hout);
        const int n = (int)(r / (unsigned)Hout);
        const int c0 = (int)cc * V;
        float sc[V], sh[V], best[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { sc[k] = s[c0 + k]; sh[k] = t[c0 + k]; best[k] = 0.0f; }  // relu >= 0
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = ho * 2 - 1 + dh;
            if ((unsigned)hi >= (unsigned)Hin) continue;
            for (int dw = 0; dw < 3; ++dw) {
                const int wi = wo * 2 - 1 + dw;
                if ((unsigned)wi >= (unsigned)Win) continue;
                float x[V];
                load_chunk<T>(in + (((long)n * Hin + hi) * Win + wi) * C + c0, x);
#pragma unroll
                for (int k = 0; k < V; ++k) best[k] = fmaxf(best[k], x[k] * sc[k] + sh[k]);
            }
        }
        store_chunk<T>(out + i * V, best);
    }
}

// global average pool: in [N][HW][C] -> out f32 [N][C]
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ in, float* __restrict__ out, int N, int HW, int C) {
    constexpr int V = Vec<T>::N;
    const int cch = C / V;
    const long total = (long)N * cch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cch);
        const int n = (int)(i / cch);
        // 4 independent partial sums (pixels p, p+1, p+2, p+3 of every group of 4) keep 4 loads in flight; combined in
        // a fixed order
        float part[4][V], acc[V];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < V; ++k) part[u][k] = 0.0f;
        const T* base = in + (long)n * HW * C + cc * V;
        int p = 0;
        for (; p + 3 < HW; p += 4) {
            float x[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_chunk<T>(base + (long)(p + u) * C, x[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < V; ++k) part[u][k] += x[u][k];
        }
        for (; p < HW; ++p) {
            float x[V];
            load_chunk<T>(base + (long)p * C, x);
#pragma unroll
            for (int k = 0; k < V; ++k) part[0][k] += x[k];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = (part[0][k] + part[1][k]) + (part[2][k] + part[3][k]);
        const float inv = 1.0f / (float)HW;
#pragma unroll
        for (int k = 0; k < V; ++k) out[(long)n * C + cc * V + k] = acc[k] * inv;
    }
}

// ------------------------------------------------------------------------------------------------------
// embedding
__device__ __forceinline__ int find_step(const int* prefix, int T, int row) {
    int t = 0;
    while (t + 1 < T && row >= prefix[t + 1]) ++t;
    return t;
}

__global__ __launch_bounds__(256) void embed_concat_fwd_kernel(const float* __restrict__ features,
                                                               const float* __restrict__ embed,
                                                               const int64_t* __restrict__ captions, long cap_stride,
                                                               const int* __restrict__ prefix, int T, int E, int V,
                                                               float* __restrict__ X) {
    const int row = blockIdx.x;
    const int t = find_step(prefix, T, row);
    const int b = row - prefix[t];
    const float* src;
    if (t == 0) {
        src = features + (long)b * E;
    } else {
        long tok = captions[(long)b * cap_stride + (t - 1)];
        tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
        src = embed + tok * E;
    }
    float* dst = X + (long)row * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) dst[e] = src[e];
}

__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ embed, const int64_t* __restrict__ ids,
                                                         long ids_stride, int E, int V, float* __restrict__ out) {
    const int b = blockIdx.x;
    long tok = ids[(long)b * ids_stride];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    for (int e = threadIdx.x; e < E; e += blockDim.x) out[(long)b * E + e] = embed[tok * E + e];
}

// deterministic scatter-add: the block of the FIRST occurrence of a token sums all its rows in row order.
__global__ __launch_bounds__(256) void embed_concat_bwd_kernel(const float* __restrict__ dX,
                                                               const int64_t* __restrict__ captions, long cap_stride,
                                                               const int* __restrict__ prefix, int T, int B, int E, int V,
                                                               int Nrows, float* __restrict__ d_embed,
                                                               float* __restrict__ d_features) {
    extern __shared__ __attribute__((aligned(16))) int tok[];   // [Nrows]; -1 for step-0 rows
    const int row = blockIdx.x;
    // packed token list, walked step by step (rows of step t are contiguous: no per-row search of the prefix array)
    for (int i = threadIdx.x; i < prefix[1]; i += blockDim.x) tok[i] = -1;
    for (int t = 1; t < T; ++t) {
        const int p0 = prefix[t], n = prefix[t + 1] - p0;
        for (int b = threadIdx.x; b < n; b += blockDim.x) {
            const long x = captions[(long)b * cap_stride + (t - 1)];
            tok[p0 + b] = (int)(x < 0 ? 0 : (x >= V ? V - 1 : x));     // memory safety only: sat_validate_ids reports bad ids
        }
    }
    __syncthreads();
    const int v = tok[row];
    if (v < 0) {   // feature rows (t == 0): b == row
        for (int e = threadIdx.x; e < E; e += blockDim.x) d_features[(long)row * E + e] = dX[(long)row * E + e];
        return;
    }
    int dup = 0;
    for (int i = threadIdx.x; i < row; i += blockDim.x) dup |= (tok[i] == v);
    if (__syncthreads_or(dup)) return;
    // later occurrences of the same token: parallel scan, gathered as a bitmap so the sum order is by row index
    __shared__ unsigned match_bits[1280];                  // 40960 rows
    const int nwords = (Nrows + 31) >> 5;
    const bool fits = nwords <= 1280;
    int any = 0;
    if (fits) {
        for (int w = threadIdx.x; w < nwords; w += blockDim.x) {
            unsigned bits = 0u;
            const int base = w << 5;
            for (int b = 0; b < 32; ++b) {
                const int i = base + b;
                if (i > row && i < Nrows && tok[i] == v) bits |= 1u << b;
            }
            match_bits[w] = bits;
            any |= (bits != 0u);
        }
    }
    any = __syncthreads_or(any);
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float acc = dX[(long)row * E + e];
        if (!fits) {
            for (int i = row + 1; i < Nrows; ++i)
                if (tok[i] == v) acc += dX[(long)i * E + e];
        } else if (any) {
            for (int w = row >> 5; w < nwords; ++w) {
                unsigned bits = match_bits[w];
                while (bits) {
                    const int b = __ffs(bits) - 1;
                    bits &= bits - 1;
                    acc += dX[(long)((w << 5) + b) * E + e];
                }
            }
        }
        d_embed[(long)v * E + e] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------
// row-wise softmax cross entropy, optionally overwriting the row with its gradient
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
    return r;
}

__global__ __launch_bounds__(256) void ce_rows_kernel(float* __restrict__ logits, long ldl, const int64_t* __restrict__ targets,
                                                      int V, float inv_denom, int write_grad, float* __restrict__ row_loss) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    float* x = logits + (long)row * ldl;
    constexpr int RC = 12;                       // 16-byte chunks of the row a thread keeps in registers
    const int nq = V >> 2;
    if ((ldl & 3) == 0 && nq <= RC * 256 && (((uintptr_t)logits) & 15) == 0) {
        // V <= 12288: the row is read ONCE into registers; max, sum and the gradient come from there (one read + one
        // write of the logits instead of three reads + one write)
        const int tid = threadIdx.x;
        f32x4 rc[RC];
#pragma unroll
        for (int c = 0; c < RC; ++c)
            if (tid + c * 256 < nq) rc[c] = *(const f32x4*)(x + 4 * (tid + c * 256));
        const int tail = (nq << 2) + tid;
        const float xt = tail < V ? x[tail] : -INFINITY;
        float m = xt;
#pragma unroll
        for (int c = 0; c < RC; ++c)
            if (tid + c * 256 < nq) m = fmaxf(fmaxf(fmaxf(m, rc[c][0]), fmaxf(rc[c][1], rc[c][2])), rc[c][3]);
        m = block_reduce(m, true, sh);
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < RC; ++c)
            if (tid + c * 256 < nq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s += expf(rc[c][e] - m);
            }
        if (tail < V) s += expf(xt - m);
        s = block_reduce(s, false, sh);
        const float lse = m + logf(s);
        long tgt = targets[row];
        tgt = tgt < 0 ? 0 : (tgt >= V ? V - 1 : tgt);
        const int tq = (int)(tgt >> 2), te = (int)(tgt & 3);
        // the thread that holds the target logit reports the row loss
#pragma unroll
        for (int c = 0; c < RC; ++c)
            if (tid + c * 256 == tq && tq < nq) row_loss[row] = lse - rc[c][te];
        if (tail == (int)tgt && tq >= nq) row_loss[row] = lse - xt;
        if (write_grad) {
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                const int q = tid + c * 256;
                if (q < nq) {
                    f32x4 g;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        g[e] = (expf(rc[c][e] - lse) - ((4 * q + e) == (int)tgt ? 1.0f : 0.0f)) * inv_denom;
                    *(f32x4*)(x + 4 * q) = g;
                }
            }
            if (tail < V) x[tail] = (expf(xt - lse) - (tail == (int)tgt ? 1.0f : 0.0f)) * inv_denom;
        }
        return;
    }
    float m = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 256) m = fmaxf(m, x[i]);
    m = block_reduce(m, true, sh);
    float s = 0.0f;
    for (int i = threadIdx.x; i < V; i += 256) s += expf(x[i] - m);
    s = block_reduce(s, false, sh);
    const float lse = m + logf(s);
    long tgt = targets[row];
    tgt = tgt < 0 ? 0 : (tgt >= V ? V - 1 : tgt);
    if (threadIdx.x == 0) row_loss[row] = lse - x[tgt];
    if (write_grad) {
        __syncthreads();   // x[tgt] read above before any overwrite
        for (int i = threadIdx.x; i < V; i += 256) {
            const float pr = expf(x[i] - lse);
            x[i] = (pr - (i == (int)tgt ? 1.0f : 0.0f)) * inv_denom;
        }
    }
}

__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ v, int n, float scale, float* out) {
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

// out[c] = sum_r x[r*ld + c]; block = 32 columns x 8 row groups, fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, int rows, int cols,
                                                     float* __restrict__ out) {
    __shared__ float sh[8][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.0f;
    if (c < cols) {
        int r = rg;
        for (; r + 24 < rows; r += 32) {          // 4 independent loads in flight; the add order stays fixed
            const float a0 = x[(long)r * ld + c], a1 = x[(long)(r + 8) * ld + c];
            const float a2 = x[(long)(r + 16) * ld + c], a3 = x[(long)(r + 24) * ld + c];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; r < rows; r += 8) s += x[(long)r * ld + c];
    }
    sh[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float t = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += sh[g][cl];
        out[c] = t;
    }
}

// ------------------------------------------------------------------------------------------------------
// fused clamp + Adam (torch.optim.Adam single-tensor arithmetic, train.py:88-91,146)
// `skip`: optional device word (f32); non-zero = some kernel of this step reported a fault (sat_step_fault_flag): the whole
// update is dropped -- parameters, moments and the gradient buffer stay bit for bit what they were
__global__ void clamp_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, long n, float beta1, float beta2, float eps, float clip,
                                  float step_size, float bc2_sqrt, const float* __restrict__ skip) {
    if (skip && *skip != 0.0f) return;
    const float w1 = 1.0f - beta1, w2 = 1.0f - beta2;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pp = ((f32x4*)p)[i], gg = ((f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gk = gg[k];
            if (clip > 0.0f) gk = fminf(fmaxf(gk, -clip), clip);
            gg[k] = gk;
            mm[k] = mm[k] + w1 * (gk - mm[k]);
            vv[k] = vv[k] * beta2 + (w2 * gk) * gk;
            const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
            pp[k] = pp[k] + (-step_size * mm[k]) / denom;
        }
        ((f32x4*)p)[i] = pp; ((f32x4*)g)[i] = gg; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
    }
    // tail (n % 4)
    const long base = n4 << 2;
    const long i = base + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float gk = g[i];
        if (clip > 0.0f) gk = fminf(fmaxf(gk, -clip), clip);
        g[i] = gk;
        const float mk = m[i] + w1 * (gk - m[i]);
        const float vk = v[i] * beta2 + (w2 * gk) * gk;
        m[i] = mk; v[i] = vk;
        p[i] = p[i] + (-step_size * mk) / (sqrtf(vk) / bc2_sqrt + eps);
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (bf16_t)in[i];
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}

// ------------------------------------------------------------------------------------------------------
// LSTM backward, pointwise part of step t (rows b < n = batch_sizes[t]):
//   dh = dHS[row] + sum_z dh_part[z][b]  (rows b < n_next only)      dc = dh*o*(1-tc^2) + dc_state[b]
//   DG[row] = (di*i*(1-i), df*f*(1-f), dg*(1-g^2), do*o*(1-o));      dc_state[b] = dc*f
__global__ void lstm_bwd_point_kernel(const float* __restrict__ dHS, const float* __restrict__ dh_part, int nz,
                                      long slab_stride, int n_next, const float* __restrict__ GA,
                                      const float* __restrict__ CS, const float* __restrict__ CS_prev,
                                      float* __restrict__ dc_state, float* __restrict__ DG, int n, int H) {
    const long total = (long)n * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % H);
        const int b = (int)(i / H);
        float dh = dHS[i];
        float dcn = 0.0f;
        if (b < n_next) {
            for (int z = 0; z < nz; ++z) dh += dh_part[(long)z * slab_stride + i];
            dcn = dc_state[i];
        }
        const float* ga = GA + (long)b * 4 * H;
        const float gi = ga[j], gf = ga[H + j], gg = ga[2 * H + j], go = ga[3 * H + j];
        const float tc = sat_tanh(CS[i]);
        const float c_prev = CS_prev ? CS_prev[i] : 0.0f;
        const float d_o = dh * tc;
        const float dc = dh * go * (1.0f - tc * tc) + dcn;
        float* dg = DG + (long)b * 4 * H;
        dg[j] = dc * gg * gi * (1.0f - gi);
        dg[H + j] = dc * c_prev * gf * (1.0f - gf);
        dg[2 * H + j] = dc * gi * (1.0f - gg * gg);
        dg[3 * H + j] = d_o * go * (1.0f - go);
        dc_state[i] = dc * gf;
    }
}

// ------------------------------------------------------------------------------------------------------
// BatchNorm1d head: block = 32 features x 8 row groups
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ part, int nz, long slab_stride,
                                                       const float* __restrict__ b_fc, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* running_mean,
                                                       float* running_var, float momentum, float eps, int training,
                                                       int B, int E, float* __restrict__ zbuf, float* __restrict__ feats,
                                                       float* __restrict__ xhat, float* __restrict__ rstd_out) {
    __shared__ float sh[8][32];
    __shared__ float bc[2][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + cl;
    const bool ok = e < E;
    float s = 0.0f;
    if (ok)
        for (int b = rg; b < B; b += 8) {
            float z = b_fc[e];
            for (int k = 0; k < nz; ++k) z += part[(long)k * slab_stride + (long)b * E + e];
            zbuf[(long)b * E + e] = z;
            s += z;
        }
    sh[rg][cl] = s;
    __syncthreads();
    if (rg == 0) {
        float t = 0.0f;
        for (int g = 0; g < 8; ++g) t += sh[g][cl];
        bc[0][cl] = t / (float)B;
    }
    __syncthreads();
    float mean = bc[0][cl];
    float q = 0.0f;
    if (ok)
        for (int b = rg; b < B; b += 8) {
            const float d = zbuf[(long)b * E + e] - mean;
            q += d * d;
        }
    __syncthreads();
    sh[rg][cl] = q;
    __syncthreads();
    if (rg == 0 && ok) {
        float t = 0.0f;
        for (int g = 0; g < 8; ++g) t += sh[g][cl];
        float var = t / (float)B;
        if (training) {
            const float unb = B > 1 ? t / (float)(B - 1) : var;
            running_mean[e] = (1.0f - momentum) * running_mean[e] + momentum * mean;
            running_var[e] = (1.0f - momentum) * running_var[e] + momentum * unb;
        } else {
            mean = running_mean[e];
            var = running_var[e];
        }
        const float rs = 1.0f / sqrtf(var + eps);
        bc[0][cl] = mean;
        bc[1][cl] = rs;
        rstd_out[e] = rs;
    }
    __syncthreads();
    if (ok) {
        const float mu = bc[0][cl], rs = bc[1][cl], ga = gamma[e], be = beta[e];
        for (int b = rg; b < B; b += 8) {
            const float xh = (zbuf[(long)b * E + e] - mu) * rs;
            xhat[(long)b * E + e] = xh;
            feats[(long)b * E + e] = xh * ga + be;
        }
    }
}

__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       int B, int E, float* __restrict__ dz, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, float* __restrict__ db_fc) {
    __shared__ float s1[8][32], s2[8][32];
    __shared__ float bc[2][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + cl;
    const bool ok = e < E;
    float a = 0.0f, c = 0.0f;
    if (ok)
        for (int b = rg; b < B; b += 8) {
            const float d = dy[(long)b * E + e];
            a += d * xhat[(long)b * E + e];
            c += d;
        }
    s1[rg][cl] = a; s2[rg][cl] = c;
    __syncthreads();
    if (rg == 0) {
        float ta = 0.0f, tc = 0.0f;
        for (int g = 0; g < 8; ++g) { ta += s1[g][cl]; tc += s2[g][cl]; }
        bc[0][cl] = ta; bc[1][cl] = tc;
        if (ok) { dgamma[e] = ta; dbeta[e] = tc; }
    }
    __syncthreads();
    float zs = 0.0f;
    if (ok) {
        const float dga = bc[0][cl], dbe = bc[1][cl];
        const float k = gamma[e] * rstd[e] / (float)B;
        for (int b = rg; b < B; b += 8) {
            const float v = k * ((float)B * dy[(long)b * E + e] - dbe - xhat[(long)b * E + e] * dga);
            dz[(long)b * E + e] = v;
            zs += v;
        }
    }
    __syncthreads();
    s1[rg][cl] = zs;
    __syncthreads();
    if (rg == 0 && ok) {
        float t = 0.0f;
        for (int g = 0; g < 8; ++g) t += s1[g][cl];
        db_fc[e] = t;
    }
}

}  // namespace

// ======================================================================================================
// host launchers
int sat_image_prep_launch(const sat_op* op, hipStream_t s) {
    if (!op->in0 || !op->out) return SAT_ERR_ARG;
    const long total = (long)op->N * op->Hin * op->Win;
    if (op->Cout != 0 && op->Cout != 4 && !(op->Cout == 8 && op->dtype == SAT_BF16)) return SAT_ERR_ARG;
    if (op->dtype == SAT_BF16 && op->Cout == 8)       // NHWC8: a 16-byte chunk per pixel for 3x3 stems (VGG conv1_1)
        hipLaunchKernelGGL((image_prep_kernel<bf16_t, 8>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0, s, (const float*)op->in0,
                           (bf16_t*)op->out, op->N, op->Hin, op->Win, op->Hout, op->Wout, op->pad);
    else if (op->dtype == SAT_BF16)
        hipLaunchKernelGGL(image_prep_kernel<bf16_t>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, s, (const float*)op->in0,
                           (bf16_t*)op->out, op->N, op->Hin, op->Win, op->Hout, op->Wout, op->pad);
    else
        hipLaunchKernelGGL(image_prep_kernel<float>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, s, (const float*)op->in0,
                           (float*)op->out, op->N, op->Hin, op->Win, op->Hout, op->Wout, op->pad);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_bn_running_apply(const sat_bn_running_item* items, int n_items, float momentum, sat_stream_t stream) {
    if (!items || n_items < 0 || !(momentum >= 0.0f && momentum <= 1.0f)) return SAT_ERR_ARG;
    if (n_items == 0) return SAT_OK;
    hipLaunchKernelGGL(bn_running_apply_kernel, dim3(n_items), dim3(256), 0, (hipStream_t)stream, items, momentum);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// nn.BatchNorm*.num_batches_tracked += 1 for every BatchNorm of a stack (one flat int64 tensor behind all of them): the last integer
// add of the product path that was a torch operator
__global__ void counter_add_kernel(long long* p, int n, long long v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += v;
}
extern "C" int sat_counter_add(int64_t* counters, int n, int64_t value, sat_stream_t stream) {
    if (!counters || n < 0) return SAT_ERR_ARG;
    if (n == 0) return SAT_OK;
    hipLaunchKernelGGL(counter_add_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (long long*)counters, n, (long long)value);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

int sat_bn_eval_batch_launch(const sat_op* op, hipStream_t s) {
    if (!op->in0 || op->count < 1 || op->count > 65535) return SAT_ERR_ARG;
    hipLaunchKernelGGL(bn_eval_batch_kernel, dim3((int)op->count), dim3(256), 0, s, (const sat_bn_eval_item*)op->in0, op->eps);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

int sat_bn_finalize_launch(const sat_op* op, int parity, hipStream_t s) {
    if (op->stat_acc) {
        // "acc mode": many-tile layers.  The per-tile slabs are reduced by MANY small workgroups (128 tiles each) into
        // the same fixed-point integer accumulators the few-tile convs feed directly, and the consuming kernel derives
        // (scale, shift) itself: ~3 us of wide parallel work instead of a 5-11 us latency-bound tail of 2-16 workgroups.
        if (!op->stat_partial || op->tiles_m < 1 || op->Cout < 1) return SAT_ERR_ARG;
        long long* acc = (long long*)op->stat_acc + (long)parity * 2 * op->Cout;
        const int groups = op->groups > 1 ? op->groups : 1;
        hipLaunchKernelGGL(bn_slab_to_acc_kernel, dim3(sat_cdiv(op->Cout, 32), sat_cdiv(op->tiles_m, SLAB_TILES_PER_WG), groups),
                           dim3(1024), 0, s, op->stat_partial, op->tiles_m, op->Cout, acc);
        SAT_LAUNCH_CHECK();
        return SAT_OK;
    }
    if (op->groups > 1) return SAT_ERR_UNSUPPORTED;          // grouped programs keep their statistics as integer sums
    if (!op->gamma || !op->beta || !op->scale_out || !op->shift_out) return SAT_ERR_ARG;
    if (op->training && !op->stat_partial) return SAT_ERR_ARG;
    if (!op->training && (!op->running_mean || !op->running_var)) return SAT_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(sat_cdiv(op->Cout, 32)), dim3(1024), 0, s, op->stat_partial, op->tiles_m,
                       op->Cout, (double)op->count, op->gamma, op->beta, op->running_mean, op->running_var,
                       op->momentum, op->eps, op->training, op->scale_out, op->shift_out);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

template <typename T>
static int bn_act_launch_t(const sat_op* op, bool add, int parity, hipStream_t s) {
    constexpr int V = Vec<T>::N;
    const int C = op->Cout;
    if (C % V) return SAT_ERR_ARG;
    const long n = (long)op->N * op->Hout * op->Wout * C;
    const long nch = n / V;
    BnSrc b0 = {}, b1 = {};
    b0.scale = op->scale0; b0.shift = op->shift0;
    b1.scale = op->scale1; b1.shift = op->shift1;
    int has_b1 = 0;
    size_t lds = 0;
    if (op->stat_acc) {              // statistics arrive as integer sums: derive the table in the kernel
        if (!op->gamma || !op->beta || op->count < 1) return SAT_ERR_ARG;
        long long* base = (long long*)op->stat_acc;          // [2 parities][2][C]
        b0.acc = base + (long)parity * 2 * C;
        b0.acc_clear = base + (long)(1 - parity) * 2 * C;
        b0.gamma = op->gamma; b0.beta = op->beta; b0.running_mean = op->running_mean; b0.running_var = op->running_var;
        lds = (size_t)4 * C * sizeof(float);
    } else if (!op->scale0 || !op->shift0) {
        return SAT_ERR_ARG;
    }
    if (add) {
        if (op->stat_acc1) {
            if (!op->gamma1 || !op->beta1 || op->count < 1) return SAT_ERR_ARG;
            long long* base = (long long*)op->stat_acc1;
            b1.acc = base + (long)parity * 2 * C;
            b1.acc_clear = base + (long)(1 - parity) * 2 * C;
            b1.gamma = op->gamma1; b1.beta = op->beta1; b1.running_mean = op->running_mean1; b1.running_var = op->running_var1;
            lds = (size_t)4 * C * sizeof(float);
            has_b1 = 1;
        } else if (op->scale1) {
            if (!op->shift1) return SAT_ERR_ARG;
            has_b1 = 1;
        }
    }
    if (lds > 64 * 1024) return SAT_ERR_UNSUPPORTED;
    // thread count a multiple of the chunks per pixel (channel chunk invariant per thread)
    const int cch = C / V;
    int grid = ew_grid(nch);
    if (lds && grid > 2048) grid = 2048;   // every workgroup derives the affine table first: fewer, fatter workgroups amortise that prologue
    if (op->groups > 1 && grid >= 2 * op->groups) grid /= op->groups;      // grouped: the same number of workgroups in total (ew_grid's cap is about wave slots per CU)
    if (cch > EW_BLOCK && (cch % EW_BLOCK) == 0) {
        const int g0 = cch / EW_BLOCK;
        grid = grid / g0 * g0;
        if (grid < g0) grid = g0;
    }
    // Workgroup size of the table-deriving launches: every workgroup turns the integer sums of ALL C channels into (scale, shift)
    // first -- f64 arithmetic, ~90 vector instructions per channel -- so at 256 threads a 1024-channel normalise+add spends ~360
    // instructions per thread there, on every one of its 768 workgroups, before it streams.  The same threads in 2x fewer, 2x
    // larger workgroups derive each channel 2x less often.  Measured (round 3, bench.py): strictly sequential steps 6.14 -> 6.02 ms
    // at 512 (6.05 at 1024); with three stacks in flight the prologue hides under the neighbours' work either way.
    constexpr int kBnBlock = 512;
    int block = EW_BLOCK;
    if (lds && (kBnBlock % cch) == 0 && grid >= kBnBlock / EW_BLOCK) {
        block = kBnBlock;
        grid = grid / (kBnBlock / EW_BLOCK);
    }
    const int groups = op->groups > 1 ? op->groups : 1;
    if (groups > 1 && !op->stat_acc) return SAT_ERR_UNSUPPORTED;       // a grouped program has per-group batch statistics
    if (groups > 1 && ((b0.running_mean && b0.running_var != b0.running_mean + C) ||
                       (b1.running_mean && b1.running_var != b1.running_mean + C))) return SAT_ERR_ARG;
    const double count = (double)op->count;
    // The normalise+add kernel reads its two operands with NON-TEMPORAL loads: both are dead after it (the raw conv3 tensor, and
    // the block input that y replaces), so they need not displace the other stacks' live tensors from the Infinity Cache.
    // Measured under look-ahead (round 3): +1.2-2.9 % at every grid size (13.55 -> 13.75 k at 2048 workgroups, 14.0 -> 14.27 k at 768).
    if (add)
        hipLaunchKernelGGL((bn_act_kernel<T, true, true>), dim3(grid, groups), dim3(block), lds, s, (const T*)op->in0, (const T*)op->in1,
                           (T*)op->out, b0, b1, has_b1, count, op->momentum, op->eps, nch, C);
    else
        hipLaunchKernelGGL((bn_act_kernel<T, false>), dim3(grid, groups), dim3(block), lds, s, (const T*)op->in0, (const T*)nullptr,
                           (T*)op->out, b0, b1, 0, count, op->momentum, op->eps, nch, C);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

template <typename T>
static int bn_relu_strided_launch_t(const sat_op* op, int parity, hipStream_t s) {
    constexpr int V = Vec<T>::N;
    const int C = op->Cout;
    if ((C % V) || (op->ldc % V) || op->ldc < C) return SAT_ERR_ARG;
    BnSrc b = {};
    b.scale = op->scale0; b.shift = op->shift0;
    size_t lds = 0;
    if (op->stat_acc) {
        if (!op->gamma || !op->beta || op->count < 1) return SAT_ERR_ARG;
        long long* base = (long long*)op->stat_acc;
        b.acc = base + (long)parity * 2 * C;
        b.acc_clear = base + (long)(1 - parity) * 2 * C;
        b.gamma = op->gamma; b.beta = op->beta; b.running_mean = op->running_mean; b.running_var = op->running_var;
        lds = (size_t)2 * C * sizeof(float);
    } else if (!op->scale0 || !op->shift0) {
        return SAT_ERR_ARG;
    }
    const long npix = (long)op->N * op->Hout * op->Wout;
    int grid = ew_grid(npix * (C / V));
    if (lds && grid > 2048) grid = 2048;
    const int groups = op->groups > 1 ? op->groups : 1;
    if (groups > 1 && (!op->stat_acc || (b.running_mean && b.running_var != b.running_mean + C))) return SAT_ERR_ARG;
    hipLaunchKernelGGL(bn_relu_strided_kernel<T>, dim3(grid, groups), dim3(EW_BLOCK), lds, s, (const T*)op->in0, (T*)op->out, (long)op->ldc, b,
                       (double)op->count, op->momentum, op->eps, npix, C);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

int sat_bn_act_launch(const sat_op* op, bool add, int parity, hipStream_t s) {
    if (!op->in0 || !op->out || (add && !op->in1)) return SAT_ERR_ARG;
    if (!add && op->ldc && op->ldc != op->Cout)        // activation written into a channel slice of a wider tensor
        return op->dtype == SAT_BF16 ? bn_relu_strided_launch_t<bf16_t>(op, parity, s) : bn_relu_strided_launch_t<float>(op, parity, s);
    return op->dtype == SAT_BF16 ? bn_act_launch_t<bf16_t>(op, add, parity, s) : bn_act_launch_t<float>(op, add, parity, s);
}

int sat_bn_relu_maxpool_launch(const sat_op* op, int parity, hipStream_t s) {
    if (!op->in0 || !op->out) return SAT_ERR_ARG;
    const int C = op->Cout;
    BnSrc b = {};
    b.scale = op->scale0; b.shift = op->shift0;
    size_t lds = 0;
    if (op->stat_acc) {
        if (!op->gamma || !op->beta || op->count < 1) return SAT_ERR_ARG;
        long long* base = (long long*)op->stat_acc;
        b.acc = base + (long)parity * 2 * C;
        b.acc_clear = base + (long)(1 - parity) * 2 * C;
        b.gamma = op->gamma; b.beta = op->beta; b.running_mean = op->running_mean; b.running_var = op->running_var;
        lds = (size_t)2 * C * sizeof(float);
        if (lds > 64 * 1024) return SAT_ERR_UNSUPPORTED;
    } else if (!op->scale0 || !op->shift0) {
        return SAT_ERR_ARG;
    }
    const double count = (double)op->count;
    const int groups = op->groups > 1 ? op->groups : 1;
    if (groups > 1 && (!op->stat_acc || (b.running_mean && b.running_var != b.running_mean + C))) return SAT_ERR_ARG;
    if (op->dtype == SAT_BF16) {
        if (C % 8) return SAT_ERR_ARG;
        const long total = (long)op->N * op->Hout * op->Wout * (C / 8);
        if (total >= (1L << 31)) return SAT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(bn_relu_maxpool_kernel<bf16_t>, dim3(ew_grid(total), groups), dim3(EW_BLOCK), lds, s, (const bf16_t*)op->in0,
                           (bf16_t*)op->out, b, count, op->momentum, op->eps, op->N, op->Hin, op->Win, C, op->Hout, op->Wout);
    } else {
        if (C % 4) return SAT_ERR_ARG;
        const long total = (long)op->N * op->Hout * op->Wout * (C / 4);
        if (total >= (1L << 31)) return SAT_ERR_UNSUPPORTED;
        if (groups > 1) return SAT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, dim3(ew_grid(total)), dim3(EW_BLOCK), lds, s, (const float*)op->in0,
                           (float*)op->out, b, count, op->momentum, op->eps, op->N, op->Hin, op->Win, C, op->Hout, op->Wout);
    }
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

int sat_avgpool_launch(const sat_op* op, hipStream_t s) {
    if (!op->in0 || !op->out) return SAT_ERR_ARG;
    const int C = op->Cout, HW = op->Hin * op->Win;
    if (op->dtype == SAT_BF16) {
        if (C % 8) return SAT_ERR_ARG;
        hipLaunchKernelGGL(avgpool_kernel<bf16_t>, dim3(ew_grid((long)op->N * C / 8)), dim3(EW_BLOCK), 0, s,
                           (const bf16_t*)op->in0, (float*)op->out, op->N, HW, C);
    } else {
        if (C % 4) return SAT_ERR_ARG;
        hipLaunchKernelGGL(avgpool_kernel<float>, dim3(ew_grid((long)op->N * C / 4)), dim3(EW_BLOCK), 0, s,
                           (const float*)op->in0, (float*)op->out, op->N, HW, C);
    }
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

int sat_lstm_bwd_point_launch(const float* dHS, const float* dh_part, int nz, long slab_stride, int n_next,
                              const float* GA, const float* CS, const float* CS_prev, float* dc_state, float* DG,
                              int n, int H, hipStream_t s) {
    hipLaunchKernelGGL(lstm_bwd_point_kernel, dim3(ew_grid((long)n * H)), dim3(EW_BLOCK), 0, s, dHS, dh_part, nz,
                       slab_stride, n_next, GA, CS, CS_prev, dc_state, DG, n, H);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

int sat_bn1d_fwd_launch(const float* part, int nz, long slab_stride, const float* b_fc, const float* gamma,
                        const float* beta, float* rm, float* rv, float momentum, float eps, int training, int B, int E,
                        float* zbuf, float* feats, float* xhat, float* rstd, hipStream_t s) {
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3(sat_cdiv(E, 32)), dim3(256), 0, s, part, nz, slab_stride, b_fc, gamma, beta,
                       rm, rv, momentum, eps, training, B, E, zbuf, feats, xhat, rstd);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// dW[e][f] = sum_b dz[b][e] * pooled[b][f] for a batch of at most a few hundred rows (the encoder head's fc gradient,
// models.py:16): one thread per (e, 4 consecutive f), an fmaf chain over b in row order (what the exact-f32 MFMA GEMM computes,
// which spends 35 us on this K = 64 product)
__global__ __launch_bounds__(256) void outer_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x, int B, int E, int F,
                                                          float* __restrict__ dw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int f4 = F >> 2;
    if (i >= (long)E * f4) return;
    const int e = (int)(i / f4), f = (int)(i - (long)e * f4) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        const float d = dz[(long)b * E + e];
        const f32x4 v = *(const f32x4*)(x + (long)b * F + f);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaf(d, v[k], acc[k]);
    }
    *(f32x4*)(dw + (long)e * F + f) = acc;
}

int sat_outer_wgrad_launch(const float* dz, const float* x, int B, int E, int F, float* dw, hipStream_t s) {
    const long n = (long)E * (F >> 2);
    hipLaunchKernelGGL(outer_wgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dz, x, B, E, F, dw);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

int sat_bn1d_bwd_launch(const float* dy, const float* xhat, const float* rstd, const float* gamma, int B, int E,
                        float* dz, float* dgamma, float* dbeta, float* db_fc, hipStream_t s) {
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3(sat_cdiv(E, 32)), dim3(256), 0, s, dy, xhat, rstd, gamma, B, E, dz, dgamma,
                       dbeta, db_fc);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_embed_concat_fwd(const float* features, const float* embed, const int64_t* captions,
                                    int64_t cap_stride, const int32_t* prefix, int T, int N, int B, int E, int V,
                                    float* X, sat_stream_t stream) {
    if (!features || !embed || !prefix || !X || T < 1 || B < 1 || N < B) return SAT_ERR_ARG;
    if (T > 1 && !captions) return SAT_ERR_ARG;
    hipLaunchKernelGGL(embed_concat_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, features, embed, captions,
                       (long)cap_stride, prefix, T, E, V, X);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_embed_concat_bwd(const float* dX, const int64_t* captions, int64_t cap_stride,
                                    const int32_t* prefix, int T, int N, int B, int E, int V, float* d_embed,
                                    float* d_features, sat_stream_t stream) {
    if (!dX || !prefix || !d_embed || !d_features || T < 1 || B < 1 || N < B) return SAT_ERR_ARG;
    if (T > 1 && !captions) return SAT_ERR_ARG;
    // the packed token list is LDS resident: 4 B per row next to the kernel's 5 KB of static LDS
    const size_t dyn = (size_t)N * 4;
    if (dyn > 150 * 1024) return SAT_ERR_UNSUPPORTED;
    if (dyn > 48 * 1024) {      // beyond the default dynamic-LDS limit the kernel has to opt in
        hipError_t ea = hipFuncSetAttribute((const void*)embed_concat_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (ea != hipSuccess) return (int)ea;
    }
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d_embed, 0, (size_t)V * E * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(embed_concat_bwd_kernel, dim3(N), dim3(256), (size_t)N * 4, s, dX, captions, (long)cap_stride,
                       prefix, T, B, E, V, N, d_embed, d_features);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_embed_rows(const float* embed, const int64_t* ids, int64_t ids_stride, int B, int E, int V,
                              float* out, sat_stream_t stream) {
    if (!embed || !ids || !out) return SAT_ERR_ARG;
    hipLaunchKernelGGL(embed_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, embed, ids, (long)ids_stride, E, V, out);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_ce_rows(float* logits, int64_t ldl, const int64_t* targets, int N, int V, float inv_denom,
                           int write_grad, float* row_loss, float* loss_out, sat_stream_t stream) {
    if (!logits || !targets || !row_loss || N < 1 || V < 1 || ldl < V) return SAT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_rows_kernel, dim3(N), dim3(256), 0, s, logits, (long)ldl, targets, V, inv_denom, write_grad, row_loss);
    SAT_LAUNCH_CHECK();
    if (loss_out) {
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, s, row_loss, N, inv_denom, loss_out);
        SAT_LAUNCH_CHECK();
    }
    return SAT_OK;
}

extern "C" int sat_colsum_f32(const float* x, int64_t ld, int rows, int cols, float* out, sat_stream_t stream) {
    if (!x || !out || rows < 1 || cols < 1) return SAT_ERR_ARG;
    hipLaunchKernelGGL(colsum_kernel, dim3(sat_cdiv(cols, 32)), dim3(256), 0, (hipStream_t)stream, x, (long)ld, rows, cols, out);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// ---- a step's fault flag: OR of the status words its residency-dependent kernels may have set ----
struct FaultWords { const unsigned* w[8]; int n; };
__global__ void step_fault_flag_kernel(const FaultWords fw, float* __restrict__ sticky, float* __restrict__ slot) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float f = sticky ? *sticky : 0.0f;
    for (int i = 0; i < fw.n; ++i)
        if (__hip_atomic_load(fw.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) f = 1.0f;
    if (sticky) *sticky = f;
    *slot = f;
}

extern "C" int sat_step_fault_flag(const void* const* status_words, int n_words, float* sticky, float* slot, sat_stream_t stream) {
    if (!slot || n_words < 0 || n_words > 8 || (n_words > 0 && !status_words)) return SAT_ERR_ARG;
    FaultWords fw = {};
    fw.n = n_words;
    for (int i = 0; i < n_words; ++i) {
        if (!status_words[i]) return SAT_ERR_ARG;
        fw.w[i] = (const unsigned*)status_words[i];
    }
    hipLaunchKernelGGL(step_fault_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, fw, sticky, slot);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_clamp_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                   float beta2, float eps, float clip, int step, sat_stream_t stream) {
    return sat_clamp_adam_step_guarded(p, g, m, v, n, lr, beta1, beta2, eps, clip, step, nullptr, stream);
}

extern "C" int sat_clamp_adam_step_guarded(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                           float beta2, float eps, float clip, int step, const float* skip_if_nonzero,
                                           sat_stream_t stream) {
    if (!p || !g || !m || !v || n < 1 || step < 1) return SAT_ERR_ARG;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return SAT_ERR_ARG;
    // bias corrections in double on the host, exactly as torch.optim.Adam does with python floats
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(clamp_adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, (hipStream_t)stream, p, g, m, v,
                       (long)n, beta1, beta2, eps, clip, step_size, bc2_sqrt, skip_if_nonzero);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_cast_f32_bf16(const float* in, void* out, int64_t n, sat_stream_t stream) {
    if (!in || !out || n < 1) return SAT_ERR_ARG;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, in, (bf16_t*)out, (long)n);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}
extern "C" int sat_cast_bf16_f32(const void* in, float* out, int64_t n, sat_stream_t stream) {
    if (!in || !out || n < 1) return SAT_ERR_ARG;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)in, out, (long)n);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// Range check of a caption / id matrix: nn.Embedding and nn.CrossEntropyLoss raise on an id outside [0, V)
// (models.py:49, train.py:143); the gather / CE kernels here only clamp for memory safety, so the wrappers launch
// this first and turn a set status word into an exception.
namespace {
__global__ __launch_bounds__(256) void validate_ids_kernel(const int64_t* __restrict__ ids, long row_stride, int rows, int cols,
                                                           long lo, long hi, int* __restrict__ status) {
    const long total = (long)rows * cols;
    int bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols, c = i - r * cols;
        const long v = ids[r * row_stride + c];
        bad |= (v < lo || v >= hi) ? 1 : 0;
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(status, 1);
}
}  // namespace

extern "C" int sat_validate_ids(const int64_t* ids, int64_t row_stride, int rows, int cols, int64_t lo, int64_t hi,
                                int32_t* status, sat_stream_t stream) {
    if (!ids || !status || rows < 0 || cols < 0 || row_stride < cols) return SAT_ERR_ARG;
    if (rows == 0 || cols == 0) return SAT_OK;
    const long total = (long)rows * cols;
    const int grid = (int)(total < 256L * 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(validate_ids_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, (long)row_stride, rows, cols,
                       (long)lo, (long)hi, (int*)status);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// targets = pack_padded_sequence(captions[:,1:], lengths-1).data (train.py:134-135): out[row(t,b)] = captions[b][t+1]
namespace {
__global__ void pack_targets_kernel(const int64_t* __restrict__ captions, long cap_stride, const int* __restrict__ prefix,
                                    int T, int N, int64_t* __restrict__ out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    const int t = find_step(prefix, T, row);
    out[row] = captions[(long)(row - prefix[t]) * cap_stride + t + 1];
}
}  // namespace

extern "C" int sat_pack_targets(const int64_t* captions, int64_t cap_stride, const int32_t* prefix, int T, int N,
                                int64_t* targets, sat_stream_t stream) {
    if (!captions || !prefix || !targets || T < 1 || N < 1) return SAT_ERR_ARG;
    hipLaunchKernelGGL(pack_targets_kernel, dim3(sat_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, captions,
                       (long)cap_stride, prefix, T, N, targets);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// out[i] = sum_z in[z*slab_stride + i] (fixed order): combines split-K partial slabs
namespace {
__global__ void sum_slabs_kernel(const float* __restrict__ in, int nslab, long slab_stride, long n4, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 acc = ((const f32x4*)in)[i];
        for (int z = 1; z < nslab; ++z) acc += ((const f32x4*)(in + (long)z * slab_stride))[i];
        ((f32x4*)out)[i] = acc;
    }
}
}  // namespace

extern "C" int sat_sum_slabs_f32(const float* in, int nslab, int64_t slab_stride, int64_t n, float* out, sat_stream_t stream) {
    if (!in || !out || nslab < 1 || n < 1 || (n & 3) || (slab_stride & 3)) return SAT_ERR_ARG;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(ew_grid(n / 4)), dim3(EW_BLOCK), 0, (hipStream_t)stream, in, nslab,
                       (long)slab_stride, (long)(n / 4), out);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}
