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

__device__ __forceinline__ float block_reduce256(float v, bool is_max, float* sh) {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
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

__global__ __launch_bounds__(256) void beam_row_kernel(const float* __restrict__ logits, long ldl,
                                                       const float* __restrict__ scores_in,
                                                       const int64_t* __restrict__ last_tok, long end_id, int K, int V,
                                                       float* __restrict__ cand_val, int* __restrict__ cand_idx) {
    __shared__ float sh[4];
    __shared__ float s_rv[4];
    __shared__ int s_ri[4];
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
    // thread-local best KMAX continuations of this row
    float val[KMAX];
    int idx[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        val[j] = -INFINITY;
        idx[j] = INT_MAX;
    }
    const int nq = V >> 2;                                   // whole 16-byte chunks (rows are 16-byte aligned: ldl % 4 == 0)
    if ((ldl & 3) == 0 && nq <= RC * 256) {
        // the row lives in registers: ONE pass over memory, then max, sum and selection from registers.
        // A thread's elements are visited in increasing index order (chunk tid, tid+256, ...; tail last).
        f32x4 rc[RC];
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            const int q = tid + c * 256;
            if (q < nq) rc[c] = *(const f32x4*)(x + 4 * q);
        }
        const int tail = (nq << 2) + tid;                    // V % 4 leftover elements, one per thread
        const float xt = tail < V ? x[tail] : -INFINITY;
        float m = xt;
#pragma unroll
        for (int c = 0; c < RC; ++c)
            if (tid + c * 256 < nq) m = fmaxf(fmaxf(fmaxf(m, rc[c][0]), fmaxf(rc[c][1], rc[c][2])), rc[c][3]);
        m = block_reduce256(m, true, sh);
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < RC; ++c)
            if (tid + c * 256 < nq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s += expf(rc[c][e] - m);
            }
        if (tail < V) s += expf(xt - m);
        s = block_reduce256(s, false, sh);
        const float lse = m + logf(s);
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            const int q = tid + c * 256;
            if (q < nq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float cv = base + (rc[c][e] - lse);
                    if (cv > val[KMAX - 1]) list_insert(val, idx, cv, k * V + 4 * q + e);
                }
            }
        }
        if (tail < V) {
            const float cv = base + (xt - lse);
            if (cv > val[KMAX - 1]) list_insert(val, idx, cv, k * V + tail);
        }
    } else {
        float m = -INFINITY;
        for (int i = tid; i < V; i += 256) m = fmaxf(m, x[i]);
        m = block_reduce256(m, true, sh);
        float s = 0.0f;
        for (int i = tid; i < V; i += 256) s += expf(x[i] - m);
        s = block_reduce256(s, false, sh);
        const float lse = m + logf(s);
        for (int i = tid; i < V; i += 256) {
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
        for (int w = 1; w < 4; ++w)
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

__global__ __launch_bounds__(64) void beam_merge_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                        int K, int V, int* __restrict__ parent,
                                                        int64_t* __restrict__ token, float* __restrict__ scores_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float v = -INFINITY;
    int ix = INT_MAX;
    if (lane < K * K) {
        v = cand_val[(long)b * K * K + lane];
        ix = cand_idx[(long)b * K * K + lane];
    }
    for (int r = 0; r < K; ++r) {
        float bv = v;
        int bi = ix;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (better(ov, oi, bv, bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (ix == bi && bi != INT_MAX) {             // the winner leaves the pool
            v = -INFINITY;
            ix = INT_MAX;
        }
        if (lane == 0) {
            const bool ok = bi != INT_MAX;
            parent[b * K + r] = ok ? bi / V : 0;
            token[b * K + r] = ok ? bi % V : 0;
            scores_out[b * K + r] = ok ? bv : -INFINITY;
        }
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
    hipLaunchKernelGGL(beam_row_kernel, dim3(B * K), dim3(256), 0, s, logits, (long)ldl, scores_in, last_tokens,
                       (long)end_id, K, V, cand_val, cand_idx);
    SAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(64), 0, s, cand_val, cand_idx, K, V, parent, token, scores_out);
    SAT_LAUNCH_CHECK();
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
