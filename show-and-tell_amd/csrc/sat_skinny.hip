// Skinny (M <= 64 rows per workgroup) f32 GEMMs for the serial parts of the decoder: the recurrent LSTM
// step (models.py:52), its backward dh chain, the encoder fc (models.py:27) and the greedy-decode vocab
// projection + argmax (models.py:61-63).
//
// A weight element is used by exactly one workgroup, once, so weights and activations go straight from
// L2 to VGPRs in MFMA operand layout (no LDS round trip): v_mfma_f32_16x16x4_f32 (exact f32), lane
// (n = l&15, kg = l>>4) loads 16 contiguous bytes of K and feeds MFMA e with k = 16*kb + 4*kg + e (the same
// K permutation on both operands).  One workgroup = 16 output columns x 64 rows; its 4 waves split K and
// reduce through LDS.  The LSTM epilogue applies the i,f,g,o gate math on the reduced tile, so h_t and
// c_t never leave the kernel un-activated: the "16 columns" of a workgroup are the 4 gates of 4 hidden units.
#include "sat_internal.h"

namespace {

enum { EPI_STORE = 0, EPI_LSTM = 1, EPI_ARGMAX = 2, EPI_LSTM_BWD = 3 };

struct SkinnyArgs {
    // operand pair 1 (required) and 2 (optional): out += A[M,K] * Wsel[16,K]^T
    const float* A; long lda; const float* W; long ldw; int K;
    const float* A2; long lda2; const float* W2; long ldw2; int K2;
    int M;          // valid rows
    int N;          // EPI_STORE/ARGMAX: output columns;  EPI_LSTM: hidden size H
    int nz;         // split-K slices (gridDim.z)
    // EPI_STORE
    float* out; long ldo; long slab_stride;   // out[z*slab_stride + row*ldo + col]
    const float* bias; const float* bias2;    // indexed by weight row
    // EPI_LSTM
    const float* xg; long ldxg;               // x-gates rows (may be NULL when pair 2 + biases are used)
    float* c_state;                           // [M,H] in place
    float* ga; long ldga;                     // activated gates out (nullable)
    float* cs;                                // [M,H] cell-state tape (nullable)
    float* h_out;                             // [M,H]
    float* h_out2; int m2;                    // second copy for rows < m2 (next step's h_prev rows), nullable
    // EPI_ARGMAX
    float* pmax; int* pidx;                   // [M][gridDim.x]
    // EPI_LSTM_BWD: out columns are hidden units j; acc = (DG_{t+1} W_hh)[row][j] for rows < m2 (the rows that have a step
    // t+1); the epilogue is the pointwise LSTM backward of step t for (row, j):
    //   dh = dhs[row][j] + acc;  dc = dh*o*(1-tanh(c)^2) + dc_state (rows < m2);  DG_t[row][g*H+j];  dc_state = dc*f
    const float* dhs;                         // [M][H]   d(loss)/d(h_t) from above
    const float* cs_prev;                     // [M][H] c_{t-1} or NULL (zeros)
    float* dg;                                // [M][4H] out
};

constexpr int NWV = 8;   // waves per workgroup: K is split 8 ways (x gridDim.z), partial tiles reduced through LDS

template <bool WKM, int RT = 4>
__device__ __forceinline__ void load_operands(const float* __restrict__ A, long lda, const float* __restrict__ W, long ldw,
                                              int K, int M, int row_chunk0, long wrow, bool wrow_ok, int kb, int lane,
                                              f32x4& b, f32x4 (&a)[RT]) {
    const int n16 = lane & 15, kg = lane >> 4;
    const int k = kb * 16 + kg * 4;
    const bool kok = k < K;   // K % 4 == 0
    b = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (kok && wrow_ok) {
        if constexpr (!WKM) {
            b = *(const f32x4*)(W + wrow * ldw + k);
        } else {
            b[0] = W[(long)(k + 0) * ldw + wrow];
            b[1] = W[(long)(k + 1) * ldw + wrow];
            b[2] = W[(long)(k + 2) * ldw + wrow];
            b[3] = W[(long)(k + 3) * ldw + wrow];
        }
    }
#pragma unroll
    for (int mt = 0; mt < RT; ++mt) {
        const int row = row_chunk0 + mt * 16 + n16;
        a[mt] = (kok && row < M) ? *(const f32x4*)(A + (long)row * lda + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// software-pipelined: the operands of K-block kb+step are in flight while the MFMAs of K-block kb issue
template <bool WKM, int RT = 4>
__device__ __forceinline__ void accumulate_pair(const float* __restrict__ A, long lda, const float* __restrict__ W,
                                                long ldw, int K, int M, int row_chunk0, long wrow, bool wrow_ok,
                                                int kb0, int kbstep, int lane, f32x4 (&acc)[RT]) {
    const int nkb = (K + 15) >> 4;
    if (kb0 >= nkb) return;
    f32x4 b, a[RT], bn, an[RT];
    load_operands<WKM, RT>(A, lda, W, ldw, K, M, row_chunk0, wrow, wrow_ok, kb0, lane, b, a);
    for (int kb = kb0; kb < nkb; kb += kbstep) {
        const bool more = kb + kbstep < nkb;
        if (more) load_operands<WKM, RT>(A, lda, W, ldw, K, M, row_chunk0, wrow, wrow_ok, kb + kbstep, lane, bn, an);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < RT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][e], b[e], acc[mt], 0, 0, 0);
        if (more) {
            b = bn;
#pragma unroll
            for (int mt = 0; mt < RT; ++mt) a[mt] = an[mt];
        }
    }
}

// One step of the LSTM backward chain (train.py:144 through models.py:52) as ONE launch: dh_t = dHS_t + DG_{t+1} W_hh for this
// workgroup's 16 hidden units x 16 batch rows (K = 4H over the 8 waves, reduced through LDS in a fixed order), then the
// pointwise gate backward of step t on the reduced tile -- the split-K slabs and the separate pointwise launch of the
// two-kernel form disappear.  Grid = (H/16, ceil(n/16)): 128 workgroups at H = 512, batch 64.
__global__ __launch_bounds__(NWV * 64) void lstm_bwd_step_kernel(const SkinnyArgs p) {
    __shared__ __attribute__((aligned(16))) float red[NWV][16][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = blockIdx.x, rc0 = blockIdx.y * 16;
    const int n16 = lane & 15;
    const int H = p.N;
    const long wrow = (long)cg * 16 + n16;                 // hidden unit fed by MFMA column n16 (W_hh is K-major here)
    const bool wrow_ok = wrow < H;
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    if (p.A) accumulate_pair<true, 1>(p.A, p.lda, p.W, p.ldw, p.K, p.m2, rc0, wrow, wrow_ok, wave, NWV, lane, acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][(lane >> 4) * 4 + e][n16] = acc[0][e];
    __syncthreads();
    if (tid >= 256) return;
    const int row = tid >> 4, jj = tid & 15;
    const int grow = rc0 + row, j = cg * 16 + jj;
    if (grow >= p.M || j >= H) return;
    float dh = p.dhs[(long)grow * H + j];
    float dcn = 0.0f;
    if (grow < p.m2) {
#pragma unroll
        for (int w = 0; w < NWV; ++w) dh += red[w][row][jj];
        dcn = p.c_state[(long)grow * H + j];
    }
    const float* ga = p.ga + (long)grow * p.ldga;
    const float gi = ga[j], gf = ga[H + j], gg = ga[2 * H + j], go = ga[3 * H + j];
    const float tc = sat_tanh(p.cs[(long)grow * H + j]);
    const float c_prev = p.cs_prev ? p.cs_prev[(long)grow * H + j] : 0.0f;
    const float d_o = dh * tc;
    const float dc = dh * go * (1.0f - tc * tc) + dcn;
    float* dg = p.dg + (long)grow * 4 * H;
    dg[j] = dc * gg * gi * (1.0f - gi);
    dg[H + j] = dc * c_prev * gf * (1.0f - gf);
    dg[2 * H + j] = dc * gi * (1.0f - gg * gg);
    dg[3 * H + j] = d_o * go * (1.0f - go);
    p.c_state[(long)grow * H + j] = dc * gf;
}

template <int EPI, bool WKM>
__global__ __launch_bounds__(NWV * 64) void skinny_kernel(const SkinnyArgs p) {
    __shared__ __attribute__((aligned(16))) float red[NWV][64][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = blockIdx.x, rc0 = blockIdx.y * 64, z = blockIdx.z;
    const int n16 = lane & 15;

    // weight row feeding output column n16 of this workgroup
    long wrow;
    bool wrow_ok;
    if constexpr (EPI == EPI_LSTM) {
        wrow = (long)(n16 >> 2) * p.N + cg * 4 + (n16 & 3);   // gate (n16>>2), hidden unit cg*4 + (n16&3)
        wrow_ok = (cg * 4 + (n16 & 3)) < p.N;
    } else {
        wrow = (long)cg * 16 + n16;
        wrow_ok = wrow < p.N;
    }

    // EPI_LSTM: the epilogue's own operands (x-gates, c_{t-1}) are fetched now, under the K loop
    float pre_xg[4] = {0.f, 0.f, 0.f, 0.f}, pre_c = 0.f;
    if constexpr (EPI == EPI_LSTM) {
        const int prow = rc0 + (tid >> 2), pj = cg * 4 + (tid & 3);
        if (tid < 256 && prow < p.M && pj < p.N) {
            pre_c = p.c_state[(long)prow * p.N + pj];
            if (p.xg) {
#pragma unroll
                for (int g = 0; g < 4; ++g) pre_xg[g] = p.xg[(long)prow * p.ldxg + (long)g * p.N + pj];
            }
        }
    }

    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accumulate_pair<WKM>(p.A, p.lda, p.W, p.ldw, p.K, p.M, rc0, wrow, wrow_ok, z * NWV + wave, NWV * p.nz, lane, acc);
    // the second operand pair has the first one's weight layout (EPI_LSTM is only built with WKM = false)
    if (p.A2) accumulate_pair<WKM>(p.A2, p.lda2, p.W2, p.ldw2, p.K2, p.M, rc0, wrow, wrow_ok, z * NWV + wave, NWV * p.nz, lane, acc);

    // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][mt * 16 + (lane >> 4) * 4 + e][n16] = acc[mt][e];
    __syncthreads();
    if (tid >= 256) return;                  // the epilogue is 64 rows x 4 column quads = 256 lanes

    const int row = tid >> 2, q = tid & 3;
    const int grow = rc0 + row;
    if constexpr (EPI == EPI_STORE) {
        f32x4 v = *(const f32x4*)&red[0][row][q * 4];
#pragma unroll
        for (int w = 1; w < NWV; ++w) {
            const f32x4 t = *(const f32x4*)&red[w][row][q * 4];
            v += t;
        }
        if (grow < p.M) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = cg * 16 + q * 4 + e;
                if (col < p.N) {
                    float o = v[e];
                    if (z == 0) {
                        if (p.bias) o += p.bias[col];
                        if (p.bias2) o += p.bias2[col];
                    }
                    p.out[(long)z * p.slab_stride + (long)grow * p.ldo + col] = o;
                }
            }
        }
    } else if constexpr (EPI == EPI_LSTM) {
        const int H = p.N;
        const int j = cg * 4 + q;
        if (grow < p.M && j < H) {
            float g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) s += red[w][row][g * 4 + q];
                const long wr = (long)g * H + j;
                s += pre_xg[g];
                if (p.bias) s += p.bias[wr];
                if (p.bias2) s += p.bias2[wr];
                g4[g] = s;
            }
            const float gi = sat_sigmoid(g4[0]), gf = sat_sigmoid(g4[1]), gg = sat_tanh(g4[2]), go = sat_sigmoid(g4[3]);
            const float c_prev = pre_c;
            const float c_new = gf * c_prev + gi * gg;
            const float h_new = go * sat_tanh(c_new);
            p.c_state[(long)grow * H + j] = c_new;
            if (p.ga) {
                float* ga = p.ga + (long)grow * p.ldga;
                ga[j] = gi; ga[H + j] = gf; ga[2 * H + j] = gg; ga[3 * H + j] = go;
            }
            if (p.cs) p.cs[(long)grow * H + j] = c_new;
            p.h_out[(long)grow * H + j] = h_new;
            if (p.h_out2 && grow < p.m2) p.h_out2[(long)grow * H + j] = h_new;
        }
    } else {  // EPI_ARGMAX: first maximal column of this 16-column group per row
        float best = -INFINITY;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = cg * 16 + q * 4 + e;
            if (col < p.N) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) s += red[w][row][q * 4 + e];
                if (p.bias) s += p.bias[col];
                if (s > best) { best = s; bidx = col; }
            }
        }
#pragma unroll
        for (int o = 1; o < 4; o <<= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bidx, o, 64);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        if (q == 0 && grow < p.M) {
            p.pmax[(long)grow * gridDim.x + cg] = best;
            p.pidx[(long)grow * gridDim.x + cg] = bidx;
        }
    }
}

__global__ __launch_bounds__(256) void argmax_reduce_kernel(const float* pmax, const int* pidx, int ncg,
                                                            int64_t* ids, int64_t ids_stride) {
    __shared__ float sb[256];
    __shared__ int si[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int c = tid; c < ncg; c += 256) {
        const float v = pmax[(long)row * ncg + c];
        const int i = pidx[(long)row * ncg + c];
        if (v > best || (v == best && i < bidx)) { best = v; bidx = i; }
    }
    sb[tid] = best; si[tid] = bidx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float ob = sb[tid + s];
            const int oi = si[tid + s];
            if (ob > sb[tid] || (ob == sb[tid] && oi < si[tid])) { sb[tid] = ob; si[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) ids[(long)row * ids_stride] = (int64_t)si[0];
}

}  // namespace

// ---- host-side launchers shared with sat_decoder.hip -------------------------------------------------
int sat_skinny_store(const float* A, long lda, const float* W, long ldw, int wkm, int M, int N, int K, int nz,
                     float* out, long ldo, long slab_stride, const float* bias, hipStream_t s) {
    if ((K & 3) || (lda & 3) || (!wkm && (ldw & 3))) return SAT_ERR_ARG;
    SkinnyArgs a = {};
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.K = K; a.M = M; a.N = N; a.nz = nz;
    a.out = out; a.ldo = ldo; a.slab_stride = slab_stride; a.bias = bias;
    dim3 grid(sat_cdiv(N, 16), sat_cdiv(M, 64), nz);
    if (wkm) hipLaunchKernelGGL((skinny_kernel<EPI_STORE, true>), grid, dim3(NWV * 64), 0, s, a);
    else hipLaunchKernelGGL((skinny_kernel<EPI_STORE, false>), grid, dim3(NWV * 64), 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

int sat_skinny_lstm(const float* h_prev, const float* w_hh, const float* x, const float* w_ih, int In,
                    const float* bias, const float* bias2, const float* xg, long ldxg, int M, int H,
                    float* c_state, float* ga, long ldga, float* cs, float* h_out, float* h_out2, int m2,
                    hipStream_t s) {
    if ((H & 3) || (In & 3)) return SAT_ERR_ARG;
    SkinnyArgs a = {};
    a.A = h_prev; a.lda = H; a.W = w_hh; a.ldw = H; a.K = H;
    a.A2 = x; a.lda2 = In; a.W2 = w_ih; a.ldw2 = In; a.K2 = In;
    a.M = M; a.N = H; a.nz = 1;
    a.bias = bias; a.bias2 = bias2; a.xg = xg; a.ldxg = ldxg;
    a.c_state = c_state; a.ga = ga; a.ldga = ldga; a.cs = cs; a.h_out = h_out; a.h_out2 = h_out2; a.m2 = m2;
    dim3 grid(sat_cdiv(H, 4), sat_cdiv(M, 64), 1);
    hipLaunchKernelGGL((skinny_kernel<EPI_LSTM, false>), grid, dim3(NWV * 64), 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int64_t sat_vocab_argmax_ws_bytes(int B, int V) { return (int64_t)B * sat_cdiv(V, 16) * 8; }

extern "C" int sat_vocab_argmax(const float* h, const float* w, const float* b, int B, int H, int V,
                                int64_t* ids, int64_t ids_stride, float* workspace, int64_t ws_bytes,
                                sat_stream_t stream) {
    if (!h || !w || !ids || !workspace || (H & 3)) return SAT_ERR_ARG;
    if (ws_bytes < sat_vocab_argmax_ws_bytes(B, V)) return SAT_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int ncg = sat_cdiv(V, 16);
    SkinnyArgs a = {};
    a.A = h; a.lda = H; a.W = w; a.ldw = H; a.K = H; a.M = B; a.N = V; a.nz = 1; a.bias = b;
    a.pmax = workspace; a.pidx = (int*)(workspace + (long)B * ncg);
    dim3 grid(ncg, sat_cdiv(B, 64), 1);
    hipLaunchKernelGGL((skinny_kernel<EPI_ARGMAX, false>), grid, dim3(NWV * 64), 0, s, a);
    SAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(argmax_reduce_kernel, dim3(B), dim3(256), 0, s, a.pmax, a.pidx, ncg, ids, ids_stride);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

extern "C" int sat_lstm_step(const float* x, const float* h_in, float* c, const float* w_ih, const float* w_hh,
                             const float* b_ih, const float* b_hh, int B, int In, int H, float* h_out,
                             sat_stream_t stream) {
    if (!x || !h_in || !c || !w_ih || !w_hh || !h_out) return SAT_ERR_ARG;
    return sat_skinny_lstm(h_in, w_hh, x, w_ih, In, b_ih, b_hh, nullptr, 0, B, H, c, nullptr, 0, nullptr, h_out,
                           nullptr, 0, (hipStream_t)stream);
}

// one nn.LSTMCell step (model2.py:58 `self.lstmcell(rnn_input, (hidden, c))`): like sat_lstm_step, plus the tapes the
// backward needs (activated gates i,f,g,o and the new cell state), both optional
extern "C" int sat_lstmcell_fwd(const float* x, const float* h_in, float* c, const float* w_ih, const float* w_hh,
                                const float* b_ih, const float* b_hh, int B, int In, int H, float* h_out,
                                float* gates /*[B,4H] or NULL*/, float* c_tape /*[B,H] or NULL*/, sat_stream_t stream) {
    if (!x || !h_in || !c || !w_ih || !w_hh || !h_out || B < 1) return SAT_ERR_ARG;
    return sat_skinny_lstm(h_in, w_hh, x, w_ih, In, b_ih, b_hh, nullptr, 0, B, H, c, gates, 4L * H, c_tape, h_out,
                           nullptr, 0, (hipStream_t)stream);
}

// one fused step of the LSTM backward chain (see lstm_bwd_step_kernel); DG_next NULL / n_next 0 for the last time step
int sat_lstm_bwd_step(const float* dHS, const float* DG_next, int n_next, const float* w_hh, const float* GA, const float* CS,
                      const float* CS_prev, float* dc_state, float* DG, int n, int H, hipStream_t s) {
    if ((H & 3) || n < 1) return SAT_ERR_ARG;
    SkinnyArgs a = {};
    a.A = n_next > 0 ? DG_next : nullptr; a.lda = 4L * H; a.W = w_hh; a.ldw = H; a.K = 4 * H;
    a.M = n; a.N = H; a.m2 = n_next; a.nz = 1;
    a.dhs = dHS; a.ga = const_cast<float*>(GA); a.ldga = 4L * H; a.cs = const_cast<float*>(CS); a.cs_prev = CS_prev;
    a.c_state = dc_state; a.dg = DG;
    dim3 grid(sat_cdiv(H, 16), sat_cdiv(n, 16), 1);
    hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, dim3(NWV * 64), 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// ------------------------------------------------------------------------------------------------------
// General M <= 64-rows-per-chunk GEMM on the skinny kernel: out[M,N] = A[M,K] * op(W) + bias, with the K range split over
// the 8 waves of a workgroup AND over grid.z slices so that a 64-row problem still fills the chip (a 64 x 64-tile GEMM would
// run on N/64 workgroups).  The per-step GEMMs of the Show-Attend-Tell decoder (model2.py:54-62: weight_hh projection,
// LSTMCell input / hidden gradients) have exactly this shape.  w_kmajor: 0 = W[n*ldw + k] ("NT"), 1 = W[k*ldw + n].
static int skinny_nz(int M, int N, int K) {
    const long wgs = (long)sat_cdiv(N, 16) * sat_cdiv(M, 64);
    int nz = (int)(384 / (wgs > 0 ? wgs : 1));
    const int max_by_k = K / (16 * NWV * 2);                 // at least two 16-wide K blocks per wave and slice
    if (nz > max_by_k) nz = max_by_k;
    return nz < 1 ? 1 : (nz > 16 ? 16 : nz);
}

extern "C" int64_t sat_skinny_gemm_ws_bytes(int M, int N, int K) {
    const int nz = skinny_nz(M, N, K);
    return nz > 1 ? (int64_t)nz * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int sat_sum_slabs_f32(const float* in, int nslab, int64_t slab_stride, int64_t n, float* out, sat_stream_t stream);

static int skinny_store2(const float* A, long lda, const float* W, long ldw, int K, const float* A2, long lda2, const float* W2,
                         long ldw2, int K2, int wkm, int M, int N, int nz, float* out, long ldo, long slab_stride,
                         const float* bias, hipStream_t s) {
    SkinnyArgs a = {};
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.K = K;
    a.A2 = A2; a.lda2 = lda2; a.W2 = W2; a.ldw2 = ldw2; a.K2 = K2;
    a.M = M; a.N = N; a.nz = nz;
    a.out = out; a.ldo = ldo; a.slab_stride = slab_stride; a.bias = bias;
    dim3 grid(sat_cdiv(N, 16), sat_cdiv(M, 64), nz);
    if (wkm) hipLaunchKernelGGL((skinny_kernel<EPI_STORE, true>), grid, dim3(NWV * 64), 0, s, a);
    else hipLaunchKernelGGL((skinny_kernel<EPI_STORE, false>), grid, dim3(NWV * 64), 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// out = A W (+ A2 W2) + bias: the optional second operand pair shares the output tile, so a sum of two products (dh_{t-1} through
// the LSTMCell and through the attention projection, model2.py:58 / 74) is one launch
extern "C" int sat_skinny_gemm2_f32(const float* A, int64_t lda, const float* W, int64_t ldw, int K, const float* A2, int64_t lda2,
                                    const float* W2, int64_t ldw2, int K2, int w_kmajor, int M, int N, const float* bias,
                                    float* out, int64_t ldo, float* workspace, int64_t ws_bytes, sat_stream_t stream) {
    if (!A || !W || !out || M < 1 || N < 1 || K < 4 || (K & 3) || (lda & 3) || ldo < N) return SAT_ERR_ARG;
    if (!w_kmajor && (ldw & 3)) return SAT_ERR_ARG;
    if (A2 && (!W2 || K2 < 4 || (K2 & 3) || (lda2 & 3) || (!w_kmajor && (ldw2 & 3)))) return SAT_ERR_ARG;
    if (!A2) K2 = 0;
    int nz = skinny_nz(M, N, K);
    if (ldo != N || (((long)M * N) & 3)) nz = 1;             // the slab sum wants a dense, 16-byte-granular result
    hipStream_t s = (hipStream_t)stream;
    if (nz == 1) return skinny_store2(A, lda, W, ldw, K, A2, lda2, W2, ldw2, K2, w_kmajor, M, N, 1, out, ldo, 0, bias, s);
    if (!workspace || ws_bytes < sat_skinny_gemm_ws_bytes(M, N, K)) return SAT_ERR_WORKSPACE;
    SAT_TRY(skinny_store2(A, lda, W, ldw, K, A2, lda2, W2, ldw2, K2, w_kmajor, M, N, nz, workspace, N, (long)M * N, bias, s));
    return sat_sum_slabs_f32(workspace, nz, (int64_t)M * N, (int64_t)M * N, out, stream);
}

extern "C" int sat_skinny_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, int w_kmajor, int M, int N, int K,
                                   const float* bias, float* out, int64_t ldo, float* workspace, int64_t ws_bytes,
                                   sat_stream_t stream) {
    return sat_skinny_gemm2_f32(A, lda, W, ldw, K, nullptr, 0, nullptr, 0, 0, w_kmajor, M, N, bias, out, ldo, workspace, ws_bytes,
                                stream);
}
