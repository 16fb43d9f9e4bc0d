// Beam decode helpers (the reference has a stub only, model2.py:113-114; greedy loop models.py:56-67).
// One decode step per image = log-softmax of K rows of logits + top-K over the K*V candidates + LSTM state
// re-ordering by parent; the sequences are recovered at the end by walking the (parent, token) back-pointers.
// HBM/L2-bound byte/compare work, two launches per step:
//   beam_row_kernel   one workgroup per (image, hypothesis) row, the row read ONCE into registers: log-sum-exp and the
//                     row's K best continuations (thread-local sorted lists in registers, K rounds of block arg-best);
//   beam_merge_kernel one wave per image: the K best of its K*K row candidates.
// The union of per-row top-K lists contains the global top-K, and both kernels order candidates by (score desc, flat
// index k*V+v asc), so the result is the exact top-K with a deterministic tie rule.
#include "sat_internal.h"
#include <limits.h>

namespace {

constexpr int KMAX = 8;   // beam width limit (thread-local candidate lists live in registers; K*K <= one wave)
constexpr int RC = 12;    // 16-byte chunks of a logits row a thread keeps in registers (V <= 12288 rows are read once)

// a better than b: higher score, ties -> lower flat candidate index (k*V + v)
__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

// wave-wide arg-best of (value, index) under `better`, the result in every lane: four DPP row rotations (8, 4, 2, 1: an all-reduce
// inside each row of 16 lanes -- the operation is commutative and idempotent) and the four row results by readlane
template <int CTRL>
__device__ __forceinline__ void argbest_step(float& v, int& i) {
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
    const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xf, 0xf, false);
    if (better(ov, oi, v, i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void wave_argbest(float& v, int& i) {
    argbest_step<0x128>(v, i);      // row_ror:8
    argbest_step<0x124>(v, i);      // row_ror:4
    argbest_step<0x122>(v, i);      // row_ror:2
    argbest_step<0x121>(v, i);      // row_ror:1
    float bv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    int bi = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const float qv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * q));
        const int qi = __builtin_amdgcn_readlane(i, 16 * q);
        if (better(qv, qi, bv, bi)) { bv = qv; bi = qi; }
    }
    v = bv;
    i = bi;
}

template <int NT>
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];      // wave order: fixed
    return r;
}

// insert (cv, ci) into a list sorted by (value desc, index asc); the caller visits candidates in increasing index
// order, so an equal value goes BEHIND the ones already stored
__device__ __forceinline__ void list_insert(float (&val)[KMAX], int (&idx)[KMAX], float cv, int ci) {
    bool shifted = false;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        const bool sw = shifted || cv > val[j];
        const float tv = val[j];
        const int ti = idx[j];
        if (sw) {
            val[j] = cv;
            idx[j] = ci;
            cv = tv;
            ci = ti;
        }
        shifted = sw;
    }
}

// One workgroup of NT threads per (image, hypothesis) row.  Round 5: the selection no longer keeps a sorted top-K list per thread
// (an insert per element that beats the list's tail: 11 k instructions per wave at V = 10 000, the kernel was instruction-bound at
// 33 us per 320 rows).  Now: a thread keeps only its single best candidate; K rounds of block arg-best; the wave that owned the
// winner strikes it out and finds its threads' next best -- the other waves skip that.  Exactly the same result: the K best
// candidates of the row under (score desc, flat index asc).
template <int NT>
__global__ __launch_bounds__(NT) void beam_row_kernel(const float* __restrict__ logits, long ldl,
                                                      const float* __restrict__ scores_in,
                                                      const int64_t* __restrict__ last_tok, long end_id, int K, int V,
                                                      float* __restrict__ cand_val, int* __restrict__ cand_idx) {
    constexpr int NW = NT / 64, RCN = RC * 256 / NT;          // waves; 16-byte chunks of the row a thread keeps in registers
    __shared__ float sh[NW];
    __shared__ float s_rv[NW];
    __shared__ int s_ri[NW];
    __shared__ float s_cv[NW * KMAX];
    __shared__ int s_ci[NW * KMAX];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int k = row % K;
    const float base = scores_in[row];
    const bool frozen = last_tok != nullptr && end_id >= 0 && last_tok[row] == end_id;
    float* cv_out = cand_val + (long)row * K;
    int* ci_out = cand_idx + (long)row * K;
    if (base == -INFINITY || frozen) {              // uniform across the block: a dead or a finished hypothesis
        if (tid < K) {
            // finished: one continuation, end_id again at unchanged score; dead: nothing
            const bool live = frozen && base != -INFINITY && tid == 0 && end_id < V;
            cv_out[tid] = live ? base : -INFINITY;
            ci_out[tid] = live ? k * V + (int)end_id : INT_MAX;
        }
        return;
    }
    const float* x = logits + (long)row * ldl;
    const int nq = V >> 2;                                   // whole 16-byte chunks (rows are 16-byte aligned: ldl % 4 == 0)
    if ((ldl & 3) == 0 && nq <= RCN * NT) {
        // the row lives in registers: ONE pass over memory (every load unconditional, chunk indexes past the row clamped and
        // ignored below: a load under a runtime condition makes the compiler branch around each one and wait for it alone)
        f32x4 rc[RCN];
#pragma unroll
        for (int c = 0; c < RCN; ++c) {
            const int q = tid + c * NT;
            rc[c] = *(const f32x4*)(x + 4 * (q < nq ? q : nq - 1));
        }
        const int tail = (nq << 2) + tid;                    // V % 4 leftover elements, one per thread
        float xt = tail < V ? x[tail] : -INFINITY;
#pragma unroll
        for (int c = 0; c < RCN; ++c)
            if (tid + c * NT >= nq) rc[c] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};       // (a select on the VALUE)
        float m = xt;
#pragma unroll
        for (int c = 0; c < RCN; ++c) m = fmaxf(fmaxf(fmaxf(m, rc[c][0]), fmaxf(rc[c][1], rc[c][2])), rc[c][3]);
        m = block_reduce<NT>(m, true, sh);
        // exp(x - m) as ONE fma and ONE v_exp_f32 per element (2^(x log2e - m log2e); 1 ulp, arguments <= 0, exp(-inf) = 0 for the
        // padding): libm's expf is ~8 instructions per element with its range fix-ups, and this loop is a third of the kernel's
        // instructions
        const float ml = m * 1.44269504088896340736f;
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < RCN; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s += __builtin_amdgcn_exp2f(__builtin_fmaf(rc[c][e], 1.44269504088896340736f, -ml));
        }
        s += __builtin_amdgcn_exp2f(__builtin_fmaf(xt, 1.44269504088896340736f, -ml));
        s = block_reduce<NT>(s, false, sh);
        const float lse = m + logf(s);
        // candidate scores in place (the f32 expression the step-by-step oracle rounds: base + (x - lse)); -inf stays -inf
#pragma unroll
        for (int c = 0; c < RCN; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) rc[c][e] = base + (rc[c][e] - lse);
        xt = base + (xt - lse);
        // this thread's best TWO (elements in increasing index order: strict > keeps the lower index among equals in front)
        float v1 = -INFINITY, v2 = -INFINITY;
        int i1 = INT_MAX, i2 = INT_MAX;
        auto offer = [&](float xv, int xi) {
            const bool a = xv > v1, b2 = xv > v2;
            v2 = a ? v1 : (b2 ? xv : v2);
            i2 = a ? i1 : (b2 ? xi : i2);
            v1 = a ? xv : v1;
            i1 = a ? xi : i1;
        };
#pragma unroll
        for (int c = 0; c < RCN; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) offer(rc[c][e], k * V + 4 * (tid + c * NT) + e);
        offer(xt, k * V + tail);
        if (v1 == -INFINITY) i1 = INT_MAX;                   // (padding / -inf logits are no candidates)
        if (v2 == -INFINITY) i2 = INT_MAX;
        // the best element of this thread that comes AFTER (pv, pi) in the order (score desc, index asc): the rare third pop
        auto local_next = [&](float pv, int pi, float& bv, int& bi) {
            bv = -INFINITY;
            bi = INT_MAX;
            auto see = [&](float xv, int xi) {
                const bool after = xv < pv || (xv == pv && xi > pi);
                if (after && xv > bv) { bv = xv; bi = xi; }
            };
#pragma unroll
            for (int c = 0; c < RCN; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) see(rc[c][e], k * V + 4 * (tid + c * NT) + e);
            see(xt, k * V + tail);
        };
        // Selection without block barriers in the rounds (round 5, second form): every WAVE extracts the K best of its own quarter of
        // the row -- K rounds of a wave-wide arg-best over the threads' current heads on DPP row rotations + four readlanes (no LDS
        // crossbar, no barrier); the owner lane pops its head (its second best is at hand; a third pop rescans, rarely) -- then ONE
        // barrier and wave 0 picks the K best of the NW * K survivors the same way.  The union of the waves' lists contains the row's
        // K best, and every comparison uses (score desc, flat index asc): the result is exactly the one of block-wide rounds.
        const int lane = tid & 63, wave = tid >> 6;
        int pops = 0;
        for (int r = 0; r < K; ++r) {
            float bv = v1;
            int bi = i1;
            wave_argbest(bv, bi);
            if (lane == 0) { s_cv[wave * KMAX + r] = bv; s_ci[wave * KMAX + r] = bi; }
            if (i1 == bi && bi != INT_MAX) {                 // the owner lane (candidate indexes are unique)
                if (pops == 0) {
                    v1 = v2;
                    i1 = i2;
                } else {
                    float nv;
                    int ni;
                    local_next(bv, bi, nv, ni);
                    v1 = nv;
                    i1 = ni;
                }
                ++pops;
            }
        }
        __syncthreads();
        if (wave == 0) {
            const bool has = lane < NW * K;
            float cv = has ? s_cv[(lane / K) * KMAX + lane % K] : -INFINITY;
            int ci = has ? s_ci[(lane / K) * KMAX + lane % K] : INT_MAX;
            for (int r = 0; r < K; ++r) {
                float bv = cv;
                int bi = ci;
                wave_argbest(bv, bi);
                if (lane == 0) { cv_out[r] = bv; ci_out[r] = bi; }
                if (ci == bi) { cv = -INFINITY; ci = INT_MAX; }
            }
        }
        return;
    }
    // rows too long for the registers: thread-local sorted lists over three passes
    float val[KMAX];
    int idx[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        val[j] = -INFINITY;
        idx[j] = INT_MAX;
    }
    {
        float m = -INFINITY;
        for (int i = tid; i < V; i += NT) m = fmaxf(m, x[i]);
        m = block_reduce<NT>(m, true, sh);
        float s = 0.0f;
        for (int i = tid; i < V; i += NT) s += expf(x[i] - m);
        s = block_reduce<NT>(s, false, sh);
        const float lse = m + logf(s);
        for (int i = tid; i < V; i += NT) {
            const float cv = base + (x[i] - lse);
            if (cv > val[KMAX - 1]) list_insert(val, idx, cv, k * V + i);
        }
    }
    // K rounds of block arg-best over the list heads; the winner pops its head
    for (int r = 0; r < K; ++r) {
        float bv = val[0];
        int bi = idx[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (better(ov, oi, bv, bi)) {
                bv = ov;
                bi = oi;
            }
        }
        __syncthreads();
        if ((tid & 63) == 0) {
            s_rv[tid >> 6] = bv;
            s_ri[tid >> 6] = bi;
        }
        __syncthreads();
        bv = s_rv[0];
        bi = s_ri[0];
#pragma unroll
        for (int w = 1; w < NW; ++w)
            if (better(s_rv[w], s_ri[w], bv, bi)) {
                bv = s_rv[w];
                bi = s_ri[w];
            }
        if (idx[0] == bi && bi != INT_MAX) {          // candidate indices are unique across threads
#pragma unroll
            for (int j = 0; j + 1 < KMAX; ++j) {
                val[j] = val[j + 1];
                idx[j] = idx[j + 1];
            }
            val[KMAX - 1] = -INFINITY;
            idx[KMAX - 1] = INT_MAX;
        }
        if (tid == 0) {
            cv_out[r] = bv;
            ci_out[r] = bi;
        }
    }
}

// One workgroup of 256 threads per image: wave 0 merges (K rounds of a wave-wide arg-best over the image's K * K row candidates), then
// -- one barrier -- all four waves move the rows of the next step: embedding row of the chosen token and, in the wide decode path, the
// (h, c) rows of the hypothesis it extends, every row by its own wave with all its loads in flight at once (the copies used to sit
// inside the merge rounds, one dependent round trip per hypothesis: 6.2 + 4.8 us per step as two launches, 8.0 as one wave)
__global__ __launch_bounds__(256) void beam_merge_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                         int K, int V, int* __restrict__ parent,
                                                         int64_t* __restrict__ token, float* __restrict__ scores_out,
                                                         const float* __restrict__ embed, int E, float* __restrict__ x_next, long ldx,
                                                         const float* __restrict__ h_src, const float* __restrict__ c_src, int H,
                                                         float* __restrict__ h_dst, long ldh, float* __restrict__ c_dst) {
    __shared__ int s_par[KMAX], s_tok[KMAX];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        float v = -INFINITY;
        int ix = INT_MAX;
        if (lane < K * K) {
            v = cand_val[(long)b * K * K + lane];
            ix = cand_idx[(long)b * K * K + lane];
        }
        for (int r = 0; r < K; ++r) {
            float bv = v;
            int bi = ix;
            wave_argbest(bv, bi);                          // (bv, bi) are the same in every lane after the reduction
            if (ix == bi && bi != INT_MAX) {             // the winner leaves the pool
                v = -INFINITY;
                ix = INT_MAX;
            }
            const bool ok = bi != INT_MAX;
            const int par = ok ? bi / V : 0, tok = ok ? bi % V : 0;
            if (lane == 0) {
                parent[b * K + r] = par;
                token[b * K + r] = tok;
                scores_out[b * K + r] = ok ? bv : -INFINITY;
                s_par[r] = par;
                s_tok[r] = tok;
            }
        }
    }
    if (!embed && !h_src) return;                          // (uniform)
    __syncthreads();
    for (int r = wave; r < K; r += 4) {
        const int par = s_par[r], tok = s_tok[r];
        const long to = (long)(b * K + r);
        if (embed) {                                       // the next step's input row: embed(token) (models.py:64)
            const float* src = embed + (long)tok * E;
            for (int e = lane * 4; e < E; e += 256) *(f32x4*)(x_next + to * ldx + e) = *(const f32x4*)(src + e);
        }
        if (h_src) {                                       // ... and the LSTM state of the hypothesis it extends
            const long from = (long)(b * K + par) * H;
            for (int e = lane * 4; e < H; e += 256) {
                const f32x4 hv = *(const f32x4*)(h_src + from + e), cv = *(const f32x4*)(c_src + from + e);
                *(f32x4*)(h_dst + to * ldh + e) = hv;
                *(f32x4*)(c_dst + to * H + e) = cv;
            }
        }
    }
}

// x0 row (b, k) = features[b]; scores: hypothesis 0 of every image live at 0, the others dead
__global__ __launch_bounds__(256) void beam_init_kernel(const float* __restrict__ features, int K, int E, long total,
                                                        float* __restrict__ x0, long ldx, float* __restrict__ scores, int R) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / E;
        x0[row * ldx + (i - row * E)] = features[(row / K) * E + (i - row * E)];
    }
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < R) scores[t] = (t % K) == 0 ? 0.0f : -INFINITY;
}

// the (h, c) rows of one layer re-ordered by parent in ONE launch: dst row (b, k) = src row (b, parent[b, k])
__global__ __launch_bounds__(256) void beam_gather2_kernel(const float* __restrict__ h_src, const float* __restrict__ c_src,
                                                           const int* __restrict__ parent, int K, int W, long total,
                                                           float* __restrict__ h_dst, long ldh, float* __restrict__ c_dst) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 2 * total; i += (long)gridDim.x * 256) {
        const bool second = i >= total;
        const long j = second ? i - total : i;
        const long row = j / W;
        const int col = (int)(j - row * W);
        const long b = row / K;
        const long from = (b * K + parent[row]) * W + col;
        if (second) c_dst[j] = c_src[from];
        else h_dst[row * ldh + col] = h_src[from];
    }
}

// wide decode steps (many rows, one LSTM layer): W_cat [4H][In + H] = [W_ih | W_hh], so that the step's gates are ONE GEMM over the
// concatenated input row [x | h]
__global__ __launch_bounds__(256) void beam_wcat_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, int In, int H,
                                                        long total, float* __restrict__ wcat) {
    const int ld = In + H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / ld;
        const int col = (int)(i - row * ld);
        wcat[i] = col < In ? w_ih[row * In + col] : w_hh[row * H + (col - In)];
    }
}

// gates [R][4H] (i, f, g, o pre-activations, both biases in) -> c (in place), h_new: the LSTM cell's pointwise half (models.py:52)
// (nslab > 1: the gates arrive as K-split partial sums, slab z `slab` floats further on; summed here in slab order)
__global__ __launch_bounds__(256) void beam_lstm_point_kernel(const float* __restrict__ gates, float* __restrict__ c,
                                                              float* __restrict__ h_new, int H, long total, int nslab, long slab) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / H;
        const int u = (int)(i - row * H);
        const float* g = gates + row * 4 * H;
        float a0 = g[u], a1 = g[H + u], a2 = g[2 * H + u], a3 = g[3 * H + u];
        for (int z = 1; z < nslab; ++z) {
            const float* gz = g + z * slab;
            a0 += gz[u]; a1 += gz[H + u]; a2 += gz[2 * H + u]; a3 += gz[3 * H + u];
        }
        const float gi = sat_sigmoid(a0), gf = sat_sigmoid(a1), gg = sat_tanh(a2), go = sat_sigmoid(a3);
        const float cn = gf * c[i] + gi * gg;
        c[i] = cn;
        h_new[i] = go * sat_tanh(cn);
    }
}

// dst row (b,k) = src row (b, parent[b,k]); W floats per row
__global__ __launch_bounds__(256) void beam_gather_kernel(const float* __restrict__ src, const int* __restrict__ parent, int K,
                                                          int W, long total, float* __restrict__ dst) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / W;
        const int col = (int)(i - row * W);
        const long b = row / K;
        dst[i] = src[(b * K + parent[row]) * W + col];
    }
}

__global__ __launch_bounds__(256) void beam_backtrack_kernel(const int* __restrict__ parents, const int64_t* __restrict__ tokens,
                                                             int T, int BK, int K, int64_t* __restrict__ ids) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= BK) return;
    const int b = row / K;
    int cur = row - b * K;
    for (int t = T - 1; t >= 0; --t) {
        const long at = (long)t * BK + b * K + cur;
        ids[(long)row * T + t] = tokens[at];
        cur = parents[at];
    }
}

// eval.py:103-109: the id -> word loop of `evaluation` stops at the first '<end>'.  One thread per caption row: the number of
// ids in front of the first end_id (T when there is none) -- the host then reads ids[b][:kept[b]] only.
__global__ __launch_bounds__(256) void kept_tokens_kernel(const int64_t* __restrict__ ids, long stride, int B, int T, int64_t end_id,
                                                          int32_t* __restrict__ kept) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int64_t* row = ids + (long)b * stride;
    int n = 0;
    while (n < T && row[n] != end_id) ++n;
    kept[b] = n;
}

}  // namespace

extern "C" int sat_kept_tokens(const int64_t* ids, int64_t stride, int B, int T, int64_t end_id, int32_t* kept,
                               sat_stream_t stream) {
    if (!ids || !kept || B <= 0 || T <= 0 || stride < T) return SAT_ERR_ARG;
    hipLaunchKernelGGL(kept_tokens_kernel, dim3(sat_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, ids, (long)stride, B, T,
                       end_id, kept);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int64_t sat_beam_step_ws_bytes(int B, int K) { return (int64_t)B * K * K * 8; }

extern "C" int sat_beam_step(const float* logits, int64_t ldl, const float* scores_in, const int64_t* last_tokens,
                             int64_t end_id, int B, int K, int V, int32_t* parent, int64_t* token, float* scores_out,
                             void* workspace, int64_t ws_bytes, sat_stream_t stream) {
    if (!logits || !scores_in || !parent || !token || !scores_out || !workspace) return SAT_ERR_ARG;
    if (B <= 0 || K <= 0 || V <= 0 || ldl < V) return SAT_ERR_ARG;
    if (K > KMAX || (long)K * V > INT_MAX - 1) return SAT_ERR_UNSUPPORTED;
    if (ws_bytes < sat_beam_step_ws_bytes(B, K)) return SAT_ERR_WORKSPACE;
    float* cand_val = (float*)workspace;                       // [B*K rows][K]
    int* cand_idx = (int*)(cand_val + (long)B * K * K);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(beam_row_kernel<256>, dim3(B * K), dim3(256), 0, s, logits, (long)ldl, scores_in, last_tokens,
                       (long)end_id, K, V, cand_val, cand_idx);
    SAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(256), 0, s, cand_val, cand_idx, K, V, parent, token, scores_out,
                       (const float*)nullptr, 0, (float*)nullptr, 0L, (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0L,
                       (float*)nullptr);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// ---- the whole beam decode of one batch as ONE call (eval.py:99's `model.sample` loop, models.py:56-67, widened to a beam; the
//      reference's own sample_beam is a stub, model2.py:113-114).  Round 4 drove the 20 steps from Python: seven launches per step
//      through ctypes, 2.44 ms per 20 steps of which the GPU was busy ~1.3 ms -- the loop was host-bound.  Here the host enqueues
//      5 launches per step (one LSTM layer) from C with no Python in between: LSTM step, exact-f32 vocab projection, per-row
//      log-softmax + top-K, per-image merge (which also gathers the next step's embedding rows), ONE (h, c) re-ordering launch per
//      layer.  Same kernels, same order, same arithmetic as the step-by-step entry points: bit-identical ids and scores. ----
static int64_t al256(int64_t n) { return (n + 255) / 256 * 256; }
constexpr int kBeamKSplitMax = 8;      // K-split slices of the wide LSTM step's gate GEMM (the workspace holds that many [R][4H] slabs)

extern "C" int64_t sat_beam_decode_ws_bytes(int B, int K, int E, int H, int V, int num_layers, int steps) {
    if (B <= 0 || K <= 0 || K > KMAX || E <= 0 || H <= 0 || V <= 0 || num_layers < 1 || steps < 1) return 0;
    const int64_t R = (int64_t)B * K, ldl = (V + 3) / 4 * 4;
    // (+ the wide path's buffers: two [R][E + H] input rows, W_cat [4H][E + H], gates [R][4H])
    // (+ the vocab projection's weights split three ways for the bf16 pipe, when the wide path runs it: sat_gemm_f32x3)
    return al256(sat_gemm_f32x3_packed_bytes(V, H)) +
           al256(4 * (int64_t)num_layers * R * H * 4) + al256(R * ldl * 4) + 2 * al256(R * 4) + al256(sat_beam_step_ws_bytes(B, K)) +
           al256((int64_t)steps * R * 4) + al256((int64_t)steps * R * 8) + 2 * al256(R * E * 4) +
           2 * al256(R * (int64_t)(E + H) * 4) + al256(4 * (int64_t)H * (E + H) * 4) + al256(kBeamKSplitMax * R * 4 * (int64_t)H * 4);
}

extern "C" int sat_beam_decode(const float* features, const float* embed, const float* const* lstm_w, int num_layers,
                               const float* lin_w, const float* lin_b, int B, int K, int E, int H, int V, int steps,
                               int64_t end_id, int64_t* ids, float* scores_out, void* workspace, int64_t ws_bytes,
                               sat_stream_t stream) {
    if (!features || !embed || !lstm_w || !lin_w || !lin_b || !ids || !workspace) return SAT_ERR_ARG;
    const int64_t need = sat_beam_decode_ws_bytes(B, K, E, H, V, num_layers, steps);
    if (need <= 0) return SAT_ERR_ARG;
    if ((E & 3) || (long)K * V > INT_MAX - 1) return SAT_ERR_UNSUPPORTED;
    if (ws_bytes < need || (((uintptr_t)workspace) & 255)) return SAT_ERR_WORKSPACE;
    for (int i = 0; i < 4 * num_layers; ++i)
        if (!lstm_w[i]) return SAT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t R = (int64_t)B * K, ldl = (V + 3) / 4 * 4;
    char* w = (char*)workspace;
    float* hc = (float*)w;                                     // [layer][h0, h1, c0, c1][R][H]
    w += al256(4 * (int64_t)num_layers * R * H * 4);
    float* logits = (float*)w; w += al256(R * ldl * 4);
    float* sc[2];
    sc[0] = (float*)w; w += al256(R * 4);
    sc[1] = (float*)w; w += al256(R * 4);
    void* cand = w; w += al256(sat_beam_step_ws_bytes(B, K));
    int32_t* parents = (int32_t*)w; w += al256((int64_t)steps * R * 4);
    int64_t* tokens = (int64_t*)w; w += al256((int64_t)steps * R * 8);
    float* x0 = (float*)w; w += al256(R * E * 4);
    float* xe = (float*)w; w += al256(R * E * 4);
    float* xh[2];
    xh[0] = (float*)w; w += al256(R * (int64_t)(E + H) * 4);
    xh[1] = (float*)w; w += al256(R * (int64_t)(E + H) * 4);
    float* wcat = (float*)w; w += al256(4 * (int64_t)H * (E + H) * 4);
    float* gates = (float*)w; w += al256(kBeamKSplitMax * R * 4 * (int64_t)H * 4);
    void* lin_split = w;                                       // [V128][H] x 3 bf16, fragment ordered (sat_gemm_f32x3_pack)
    hipError_t e = hipMemsetAsync(hc, 0, (size_t)(4 * (int64_t)num_layers * R * H * 4), s);      // h = c = 0
    if (e != hipSuccess) return (int)e;
    // WIDE steps (round 5): with one LSTM layer and >= 128 rows the step's gates are ONE LDS-tiled exact-f32 MFMA GEMM over the
    // concatenated row [x | h] (sat_gemm_f32: ~0.5 of the f32 peak) + a pointwise launch, instead of the few-rows kernel, which
    // re-reads the 6 MB of weights once per 64-row chunk (34 -> ~17 us per step at 320 rows)
    const bool wide = num_layers == 1 && R >= 128 && !(H & 3);
    const long ldx = E + H;
    // the gate GEMM of a step is small ([R][E + H] x [E + H][4H]: 160 tiles of 64 x 64 at 320 rows): its K axis is dealt over enough
    // slices to fill the chip's ~512 workgroup slots, the pointwise launch sums the slices in order (measured at 320 rows, 20 steps:
    // 1 slice 1.82 ms, 2 1.76, 3 1.685, 4 1.73, 6 1.75, 8 1.85 -- profiles/r05_decode_loop.txt); SAT_BEAM_KSPLIT overrides
    static const int ks_env = [] { const char* e = getenv("SAT_BEAM_KSPLIT"); return e ? atoi(e) : 0; }();
    int ksplit = 1;
    if (wide) {
        const long tiles = (long)sat_cdiv((int)R, 64) * sat_cdiv(4 * H, 64);
        ksplit = (int)(512 / (tiles > 0 ? tiles : 1));
        const int nk = sat_cdiv((int)ldx, 32);                 // K-steps of the f32 kernel
        if (ksplit > nk / 4) ksplit = nk / 4;
        if (ksplit > kBeamKSplitMax) ksplit = kBeamKSplitMax;
        if (ksplit < 1) ksplit = 1;
        if (ks_env >= 1 && ks_env <= kBeamKSplitMax) ksplit = ks_env;
    }
    // WIDE projection (round 5): with >= 128 rows the vocab projection -- half of a step on the exact-f32 pipe -- runs on the bf16
    // pipe from a three-way split of both operands (sat_gemm_x3.hip: f32 accuracy, not the f32 pipe's bit pattern; the weights are
    // split once per call).  SAT_BEAM_X3=0: the exact-f32 pipe
    static const bool x3_env = [] { const char* v = getenv("SAT_BEAM_X3"); return !(v && v[0] == '0'); }();
    const bool x3 = wide && x3_env && sat_gemm_f32x3_packed_bytes(V, H) > 0 && !(V & 3);
    if (x3) SAT_TRY(sat_gemm_f32x3_pack(lin_w, V, H, lin_split, stream));
    if (wide) {
        e = hipMemsetAsync(xh[0], 0, (size_t)(R * ldx * 4), s);                                  // h_{-1} = 0
        if (e != hipSuccess) return (int)e;
        const long total = 4L * H * ldx;
        int grid = sat_cdiv(total, 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(beam_wcat_kernel, dim3(grid), dim3(256), 0, s, lstm_w[0], lstm_w[1], E, H, total, wcat);
        SAT_LAUNCH_CHECK();
    }
    if (ldl > V) {                                             // pad columns of the logits rows: read by nobody, but keep them defined
        e = hipMemsetAsync(logits, 0, (size_t)(R * ldl * 4), s);
        if (e != hipSuccess) return (int)e;
    }
    {
        const long total = R * E;                              // (>= R: the scores, one per thread, are covered by the same grid)
        int grid = sat_cdiv(total, 256);
        if (grid > 2048) grid = 2048;
        if ((long)grid * 256 < R) grid = sat_cdiv(R, 256);
        hipLaunchKernelGGL(beam_init_kernel, dim3(grid), dim3(256), 0, s, features, K, E, total, wide ? xh[0] : x0, wide ? ldx : (long)E,
                           sc[0], (int)R);
        SAT_LAUNCH_CHECK();
    }
    float* cand_val = (float*)cand;
    int* cand_idx = (int*)(cand_val + (long)B * K * K);
    int cur_h[8] = {0}, cur_c[8] = {0};                        // which of the two buffers holds a layer's live h / c
    if (num_layers > 8) return SAT_ERR_UNSUPPORTED;
    const float* x = x0;
    int si = 0, xi = 0;
    for (int i = 0; i < steps; ++i) {
        const float* inp = x;
        if (wide) {
            float* c = hc + (long)(2 + cur_c[0]) * R * H;
            float* h_new = hc + (long)cur_h[0] * R * H;
            SAT_TRY(sat_gemm_f32_splitk(0, 0, xh[xi], ldx, wcat, ldx, gates, 4L * H, lstm_w[2], lstm_w[3], (int)R, 4 * H, (int)ldx, ksplit,
                                        R * 4L * H, stream));
            {
                const long total = R * H;
                int grid = sat_cdiv(total, 256);
                if (grid > 2048) grid = 2048;
                hipLaunchKernelGGL(beam_lstm_point_kernel, dim3(grid), dim3(256), 0, s, gates, c, h_new, H, total, ksplit, R * 4L * H);
                SAT_LAUNCH_CHECK();
            }
            inp = h_new;
        }
        for (int l = 0; l < num_layers && !wide; ++l) {
            float* base = hc + (long)l * 4 * R * H;
            float* h_in = base + (long)cur_h[l] * R * H;
            float* h_out = base + (long)(1 - cur_h[l]) * R * H;
            float* c = base + (long)(2 + cur_c[l]) * R * H;
            const int In = l == 0 ? E : H;
            SAT_TRY(sat_lstm_step(inp, h_in, c, lstm_w[4 * l], lstm_w[4 * l + 1], lstm_w[4 * l + 2], lstm_w[4 * l + 3], (int)R, In, H,
                                  h_out, stream));
            cur_h[l] = 1 - cur_h[l];
            inp = h_out;
        }
        if (x3) SAT_TRY(sat_gemm_f32x3(inp, H, lin_split, lin_b, logits, ldl, (int)R, V, H, stream));
        else SAT_TRY(sat_vocab_logits_fwd(inp, lin_w, lin_b, (int)R, H, V, logits, ldl, stream));
        int32_t* par = parents + (long)i * R;
        int64_t* tok = tokens + (long)i * R;
        const int64_t* last = (i > 0 && end_id >= 0) ? tok - R : (const int64_t*)nullptr;
        hipLaunchKernelGGL(beam_row_kernel<256>, dim3((unsigned)R), dim3(256), 0, s, logits, (long)ldl, sc[si], last, (long)end_id, K, V,
                           cand_val, cand_idx);
        SAT_LAUNCH_CHECK();
        if (wide) {
            // ... the merge wave also moves (h, c) by parent: h straight into the next input row's h half (one launch fewer per step)
            hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(256), 0, s, cand_val, cand_idx, K, V, par, tok, sc[1 - si], embed, E,
                               xh[1 - xi], ldx, (const float*)(hc + (long)cur_h[0] * R * H), (const float*)(hc + (long)(2 + cur_c[0]) * R * H), H,
                               xh[1 - xi] + E, ldx, hc + (long)(2 + 1 - cur_c[0]) * R * H);
            SAT_LAUNCH_CHECK();
            si = 1 - si;
            cur_c[0] = 1 - cur_c[0];
            xi = 1 - xi;
            x = xe;
            continue;
        }
        hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(256), 0, s, cand_val, cand_idx, K, V, par, tok, sc[1 - si], embed, E, xe, (long)E,
                           (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0L, (float*)nullptr);
        SAT_LAUNCH_CHECK();
        si = 1 - si;
        if (K > 1) {
            for (int l = 0; l < num_layers; ++l) {
                float* base = hc + (long)l * 4 * R * H;
                const long total = R * H;
                int grid = sat_cdiv(2 * total, 256);
                if (grid > 2048) grid = 2048;
                hipLaunchKernelGGL(beam_gather2_kernel, dim3(grid), dim3(256), 0, s, base + (long)cur_h[l] * R * H,
                                   base + (long)(2 + cur_c[l]) * R * H, par, K, H, total, base + (long)(1 - cur_h[l]) * R * H, (long)H,
                                   base + (long)(2 + 1 - cur_c[l]) * R * H);
                SAT_LAUNCH_CHECK();
                cur_h[l] = 1 - cur_h[l];
                cur_c[l] = 1 - cur_c[l];
            }
        }
        x = xe;
    }
    hipLaunchKernelGGL(beam_backtrack_kernel, dim3(sat_cdiv(R, 256)), dim3(256), 0, s, parents, tokens, steps, (int)R, K, ids);
    SAT_LAUNCH_CHECK();
    if (scores_out) {
        e = hipMemcpyAsync(scores_out, sc[si], (size_t)(R * 4), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
    }
    return SAT_OK;
}

extern "C" int sat_beam_gather_rows(const float* src, const int32_t* parent, int B, int K, int width, float* dst,
                                    sat_stream_t stream) {
    if (!src || !parent || !dst || src == dst) return SAT_ERR_ARG;
    if (B <= 0 || K <= 0 || width <= 0) return SAT_ERR_ARG;
    const long total = (long)B * K * width;
    int grid = sat_cdiv(total, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(beam_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, parent, K, width, total, dst);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_beam_backtrack(const int32_t* parents, const int64_t* tokens, int T, int B, int K, int64_t* ids,
                                  sat_stream_t stream) {
    if (!parents || !tokens || !ids) return SAT_ERR_ARG;
    if (T <= 0 || B <= 0 || K <= 0) return SAT_ERR_ARG;
    hipLaunchKernelGGL(beam_backtrack_kernel, dim3(sat_cdiv((long)B * K, 256)), dim3(256), 0, (hipStream_t)stream, parents,
                       tokens, T, B * K, K, ids);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// ---- greedy decode (models.py:56-67 `DecoderRNN.sample`) as ONE call: steps x (LSTM step per layer, vocab projection + arg-max,
//      embedding row of the chosen id), enqueued from C.  h / c: [num_layers][B][H] initial state IN, final state OUT (updated in
//      place; h_tmp: scratch of the same size).  ids [B][ids_stride] i64, column i = step i. ----
extern "C" int sat_greedy_decode(const float* features, const float* embed, const float* const* lstm_w, int num_layers,
                                 const float* lin_w, const float* lin_b, int B, int E, int H, int V, int steps, float* h, float* c,
                                 float* h_tmp, float* x_tmp /*[B][E]*/, int64_t* ids, int64_t ids_stride, float* workspace,
                                 int64_t ws_bytes, sat_stream_t stream) {
    if (!features || !embed || !lstm_w || !lin_w || !lin_b || !h || !c || !h_tmp || !x_tmp || !ids || !workspace) return SAT_ERR_ARG;
    if (B <= 0 || E <= 0 || H <= 0 || V <= 0 || num_layers < 1 || num_layers > 8 || steps < 1 || ids_stride < steps) return SAT_ERR_ARG;
    if (ws_bytes < sat_vocab_argmax_ws_bytes(B, V)) return SAT_ERR_WORKSPACE;
    float* hb[8][2];
    for (int l = 0; l < num_layers; ++l) { hb[l][0] = h + (long)l * B * H; hb[l][1] = h_tmp + (long)l * B * H; }
    int cur[8] = {0};
    const float* x = features;
    for (int i = 0; i < steps; ++i) {
        const float* inp = x;
        for (int l = 0; l < num_layers; ++l) {
            SAT_TRY(sat_lstm_step(inp, hb[l][cur[l]], c + (long)l * B * H, lstm_w[4 * l], lstm_w[4 * l + 1], lstm_w[4 * l + 2],
                                  lstm_w[4 * l + 3], B, l == 0 ? E : H, H, hb[l][1 - cur[l]], stream));
            cur[l] = 1 - cur[l];
            inp = hb[l][cur[l]];
        }
        SAT_TRY(sat_vocab_argmax(inp, lin_w, lin_b, B, H, V, ids + i, ids_stride, workspace, ws_bytes, stream));
        SAT_TRY(sat_embed_rows(embed, ids + i, ids_stride, B, E, V, x_tmp, stream));
        x = x_tmp;
    }
    for (int l = 0; l < num_layers; ++l)
        if (cur[l]) {                                          // an odd number of steps: the live hidden state sits in the scratch
            hipError_t e = hipMemcpyAsync(hb[l][0], hb[l][1], (size_t)B * H * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
    return SAT_OK;
}
