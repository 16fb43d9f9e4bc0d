// bf16 throughput mode of the decoder's three big GEMMs (models.py:53 `self.linear(hiddens[0])` and its backward,
// train.py:143-144): vocab projection, d(linear.weight) and d(hiddens) on v_mfma_f32_32x32x16_bf16 with f32 accumulate.
//
//   logits[N,V] = Hs[N,H]  W[V,H]^T + b          M = N rows, N = V, K = H
//   dW[V,H]     = G^T[V,N] Hs^T[H,N]^T           M = V,      N = H, K = N rows      (G = d loss / d logits)
//   dHs[N,H]    = G[N,V]   W^T[H,V]^T            M = N rows, N = H, K = V  (split-K slabs, summed in fixed order)
//
// In the f32 parity mode these run as exact-f32 MFMA GEMMs (sat_gemm.hip: 0.57 of a 157 TFLOP/s pipe, 0.39 ms per step at
// BASELINE configs[1]); the configuration names bf16, whose pipe is 16x wider.  All three are "NT" products (both operands
// K-contiguous), so ONE kernel serves them: the LDS-DMA ring of the conv kernel's 1x1 path (global_load_lds_dwordx4 into an
// XOR-swizzled image, counted s_waitcnt vmcnt(N), one raw s_barrier per K-step) with an f32 epilogue staged through LDS so the
// output leaves as 16-byte stores.  The operands are bf16 COPIES made per step by cast / tiled-transpose kernels (K padded to
// a multiple of 64 with zeros, so the K loop has no tail); the f32 master weights, the CE arithmetic (exp / log in f32 from f32
// logits), the bias gradient and every accumulation stay f32.  d loss / d logits is written ONCE, as bf16, by the CE kernel.
#include "sat_internal.h"

namespace {

__device__ u32x4 g_zero16_b;   // zero-initialised: the source of every out-of-range chunk

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct GemmArgs {
    const bf16_t* A;      // [M][lda], K contiguous
    const bf16_t* B;      // [N][ldb], K contiguous
    float* C;             // [M][ldc] (+ z * slab for split-K slice z)
    const float* bias;    // [N] or NULL (ignored when ksplit > 1)
    const float* bias2;   // second bias added like the first (nn.LSTM's b_ih + b_hh), or NULL
    int M, N, K;          // K: multiple of 64
    long lda, ldb, ldc, slab;
    int tiles_n, ksteps;  // K-steps of 64 per split slice
};

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N == 0 || N == 4 || N == 6 || N == 8 || N == 12, "add the vmcnt literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// 128 x BN tile, K-step 64, S ring stages, 8 waves as 2 (M) x 4 (N): each wave owns 64 x BN/4 of the tile
template <int BN, int S>
__global__ __launch_bounds__(512) void gemm_bf16_nt_kernel(const GemmArgs p) {
    constexpr int BM = 128, NT = 512;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int WGM = 2, WGN = 4, WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int NAI = BM / 64, NBI = BN / 64, LPW = NAI + NBI;       // 8-row x 128-B LDS-DMA pieces per wave per stage
    constexpr int D = S - 1;
    constexpr int CROWF = BN + 4;                                      // f32 epilogue row stride (floats)
    static_assert(64 * CROWF * 4 <= S * STAGE, "half of the f32 output tile must fit the ring");
    __shared__ __attribute__((aligned(16))) char smem[S * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x, bid = blockIdx.x, z = blockIdx.y;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_m = swz / p.tiles_n, tile_n = swz - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = z * p.ksteps * 64;
    int nk = (p.K - k_begin) / 64;
    if (nk > p.ksteps) nk = p.ksteps;
    if (nk < 0) nk = 0;

    const bf16_t* zero = (const bf16_t*)&g_zero16_b;
    const bf16_t* a_ptr[NAI];
    const bf16_t* b_ptr[NBI];
    int a_step[NAI], b_step[NBI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int row = wave * (NAI * 8) + i * 8 + (lane >> 3);
        const int c = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        const bool ok = m0 + row < p.M;
        a_ptr[i] = ok ? p.A + ((long)(m0 + row) * p.lda + k_begin + c) : zero;
        a_step[i] = ok ? 64 : 0;
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int row = wave * (NBI * 8) + i * 8 + (lane >> 3);
        const int c = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        const bool ok = n0 + row < p.N;
        b_ptr[i] = ok ? p.B + ((long)(n0 + row) * p.ldb + k_begin + c) : zero;
        b_step[i] = ok ? 64 : 0;
    }
    int issued = 0;
    auto issue = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
        const bool live = issued < nk;                  // past the K range: zero-source dummies keep the counts uniform
        ++issued;
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(live ? a_ptr[i] : zero), (lptr_t)(sA + (wave * (NAI * 8) + i * 8) * 128), 16, 0, 0);
            a_ptr[i] += a_step[i];
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(live ? b_ptr[i] : zero), (lptr_t)(sB + (wave * (NBI * 8) + i * 8) * 128), 16, 0, 0);
            b_ptr[i] += b_step[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    int a_off[TM][4], b_off[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * WM + i * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a_off[i][ks] = row * 128 + (((2 * ks + h) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * WN + j * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b_off[j][ks] = A_BYTES + row * 128 + (((2 * ks + h) ^ ((row >> 1) & 7)) << 4);
    }

#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        wait_vm<LPW*(D - 1)>();                       // my pieces of K-step kt have landed ...
        __builtin_amdgcn_s_barrier();                 // ... and everybody's; everybody is done reading K-step kt-1
        asm volatile("" ::: "memory");
        int nbuf = buf + D;
        if (nbuf >= S) nbuf -= S;
        issue(nbuf);
        const char* st = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(st + a_off[i][ks]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *(const bf16x8*)(st + b_off[j][ks]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        buf = (buf + 1 == S) ? 0 : buf + 1;
    }
    wait_vm<0>();
    __syncthreads();

    // ---- epilogue: f32 tile through LDS, 64 rows (one wave row) at a time; C/D map: col = lane&31, row = (e&3)+8*(e>>2)+4*(lane>>5)
    float* ct = (float*)smem;
    float* C = p.C + (long)z * p.slab;
    const bool add_bias = p.bias != nullptr && gridDim.y == 1;
#pragma unroll
    for (int hf = 0; hf < WGM; ++hf) {
        if (wm == hf) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * WN + j * 32 + r;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ct[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CROWF + col] = acc[i][j][e];
            }
        }
        __syncthreads();
        constexpr int CPR = BN / 4;                    // 16-byte chunks per row
#pragma unroll
        for (int it = 0; it < 64 * CPR / NT; ++it) {
            const int q = tid + it * NT;
            const int row = q / CPR, c4 = q - row * CPR;
            const int grow = m0 + hf * 64 + row, gcol = n0 + c4 * 4;
            if (grow < p.M && gcol < p.N) {             // N % 4 == 0: a chunk is all in or all out
                f32x4 v = *(const f32x4*)(ct + row * CROWF + c4 * 4);
                if (add_bias) {
                    v += *(const f32x4*)(p.bias + gcol);
                    if (p.bias2) v += *(const f32x4*)(p.bias2 + gcol);
                }
                *(f32x4*)(C + (long)grow * p.ldc + gcol) = v;
            }
        }
        __syncthreads();
    }
}

// ---- operand preparation -------------------------------------------------------------------------------------------------
// f32 [R][ldi] -> bf16 [R][ldo], columns [C, ldo) zero (ldo % 8 == 0)
__global__ __launch_bounds__(256) void cast_rows_bf16_kernel(const float* __restrict__ in, long ldi, int R, int C, bf16_t* __restrict__ out,
                                                             long ldo) {
    const long per_row = ldo >> 3, total = (long)R * per_row;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long row = q / per_row;
        const int c0 = (int)(q - row * per_row) * 8;
        const float* src = in + row * ldi + c0;
        bf16x8 o;
        if (c0 + 8 <= C && ((((uintptr_t)src) & 15) == 0)) {
            const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)a[e]; o[e + 4] = (bf16_t)b[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(c0 + e < C ? src[e] : 0.0f);
        }
        *(bf16x8*)(out + row * ldo + c0) = o;
    }
}

template <typename T> __device__ __forceinline__ float ld_f(const T* p);
template <> __device__ __forceinline__ float ld_f<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f<bf16_t>(const bf16_t* p) { return (float)*p; }

// in [R][ldi] (f32 or bf16) -> out bf16 [C][ldo] = in^T, columns [R, ldo) zero; 64 x 64 tiles through LDS
// psum (optional): psum[blockIdx.y][c] = f32 sum over this tile's 64 input rows of column c (fixed order) -- summed over the
// row tiles by the caller, that is the column sum of `in` (the bias gradient, when `in` is d loss / d logits)
template <typename Tin>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const Tin* __restrict__ in, long ldi, int R, int C, bf16_t* __restrict__ out,
                                                             long ldo, float* __restrict__ psum) {
    __shared__ bf16_t tile[64][66];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, t = threadIdx.x;
    {
        const int cc = (t & 15) * 4, rr = t >> 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = rr + pass * 16;
            const long grow = r0 + row;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gc = c0 + cc + e;
                tile[row][cc + e] = (bf16_t)((grow < R && gc < C) ? ld_f<Tin>(in + grow * ldi + gc) : 0.0f);
            }
        }
    }
    __syncthreads();
    if (psum && t < 64 && c0 + t < C) {
        float a = 0.0f;
#pragma unroll 8
        for (int rr2 = 0; rr2 < 64; ++rr2) a += (float)tile[rr2][t];
        psum[(long)blockIdx.y * C + c0 + t] = a;
    }
    {
        const int orr = (t & 15) * 4, oc = t >> 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int c = oc + pass * 16;
            if (c0 + c < C && r0 + orr < ldo) {        // ldo % 4 == 0: a 4-element group is all in or all out
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tile[orr + e][c];
                *(bf16x4*)(out + (long)(c0 + c) * ldo + r0 + orr) = v;
            }
        }
    }
}

// column sums of a bf16 matrix in f32, fixed order: block = 64 columns, 8 row groups
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long ld, int rows, int cols, float* __restrict__ out) {
    __shared__ float sh[8][64];
    const int c2 = (threadIdx.x & 31) * 2, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 64 + c2;
    float s0 = 0.0f, s1 = 0.0f;
    if (c < cols) {
        for (int r = rg; r < rows; r += 8) {
            const unsigned u = *(const unsigned*)(x + (long)r * ld + c);       // two bf16 (ld, c even)
            s0 += __uint_as_float(u << 16);
            s1 += __uint_as_float(u & 0xffff0000u);
        }
    }
    sh[rg][c2] = s0;
    sh[rg][c2 + 1] = s1;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = blockIdx.x * 64 + threadIdx.x;
        if (cc < cols) {
            float t = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) t += sh[g][threadIdx.x];
            out[cc] = t;
        }
    }
}

__device__ __forceinline__ float blk_reduce(float v, bool is_max, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
    return r;
}

// Row-wise softmax cross entropy (train.py:53,143) from f32 logits, arithmetic identical to ce_rows_kernel's register path;
// d loss / d logits = (softmax - onehot) * inv_denom leaves ONCE, rounded to bf16, into grad[row][0..ldg) (pad columns zero);
// the logits are not modified.  V % 4 == 0, V <= 12288.
__global__ __launch_bounds__(256) void ce_rows_bf16grad_kernel(const float* __restrict__ logits, long ldl, const int64_t* __restrict__ targets,
                                                               int V, float inv_denom, float* __restrict__ row_loss,
                                                               bf16_t* __restrict__ grad, long ldg) {
    __shared__ float sh[4];
    constexpr int RC = 12;
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (long)row * ldl;
    const int nq = V >> 2;
    f32x4 rc[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c)
        if (tid + c * 256 < nq) rc[c] = *(const f32x4*)(x + 4 * (tid + c * 256));
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < RC; ++c)
        if (tid + c * 256 < nq) m = fmaxf(fmaxf(fmaxf(m, rc[c][0]), fmaxf(rc[c][1], rc[c][2])), rc[c][3]);
    m = blk_reduce(m, true, sh);
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < RC; ++c)
        if (tid + c * 256 < nq) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s += expf(rc[c][e] - m);
        }
    s = blk_reduce(s, false, sh);
    const float lse = m + logf(s);
    long tgt = targets[row];
    tgt = tgt < 0 ? 0 : (tgt >= V ? V - 1 : tgt);
    const int tq = (int)(tgt >> 2), te = (int)(tgt & 3);
    bf16_t* g = grad + (long)row * ldg;
#pragma unroll
    for (int c = 0; c < RC; ++c) {
        const int q = tid + c * 256;
        if (q < nq) {
            if (q == tq) row_loss[row] = lse - rc[c][te];
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)((expf(rc[c][e] - lse) - ((4 * q + e) == (int)tgt ? 1.0f : 0.0f)) * inv_denom);
            *(bf16x4*)(g + 4 * q) = o;
        } else if (4 * q < ldg) {                       // pad columns [V, ldg): zeros (the K axis of the dHs product)
            bf16x4 o = {(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};
            *(bf16x4*)(g + 4 * q) = o;
        }
    }
}

__global__ __launch_bounds__(256) void sum_scale_b_kernel(const float* __restrict__ v, int n, float scale, float* out) {
    __shared__ float sh[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = blk_reduce(s, false, sh);
    if (threadIdx.x == 0) out[0] = s * scale;
}

int pad64(int x) { return (x + 63) / 64 * 64; }

int launch_gemm(const bf16_t* A, long lda, const bf16_t* B, long ldb, float* C, long ldc, const float* bias, int M, int N, int K,
                int ksplit, long slab, hipStream_t s, const float* bias2 = nullptr) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 64 || (K % 64) || (N % 4) || (lda % 8) || (ldb % 8) || (ldc % 4) || ksplit < 1)
        return SAT_ERR_ARG;
    GemmArgs a = {};
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.bias2 = bias2; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.slab = slab;
    const int nk = K / 64;
    a.ksteps = sat_cdiv(nk, ksplit);
    const int tm = sat_cdiv(M, 128), tn = sat_cdiv(N, 128);
    a.tiles_n = tn;
    const dim3 grid(tm * tn, sat_cdiv(nk, a.ksteps)), block(512);
    // ring depth, measured at cfg 2 (tools/microbench.py gemm16): more workgroups than CUs -> 2 stages (64 KB of LDS: two
    // workgroups per CU, one's f32 epilogue under the other's K loop: logits 29.1 vs 36.3 us, dW 26.4 vs 30.1 us); one round
    // of long-K workgroups -> 3 stages (dHs, split-K 6: 23.5 vs 26.5 us); a 4th stage never paid
    const int S = (long)grid.x * grid.y > 256 ? 2 : 3;
    if (S == 2) hipLaunchKernelGGL((gemm_bf16_nt_kernel<128, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_bf16_nt_kernel<128, 3>), grid, block, 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// workspace layout of the bf16 vocab path (all offsets 256-byte aligned)
struct VocabWs {
    int64_t hsb, wb, wtb, hstb, gb, gtb, slabs, psum, total;
    int Npad, Vpad, ks;
};
VocabWs vocab_ws(int N, int H, int V) {
    VocabWs w = {};
    w.Npad = pad64(N); w.Vpad = pad64(V);
    auto al = [](int64_t x) { return (x + 255) / 256 * 256; };
    int64_t off = 0;
    w.hsb = off;  off = al(off + (int64_t)N * H * 2);            // Hs bf16 [N][H]
    w.wb = off;   off = al(off + (int64_t)V * H * 2);            // W bf16 [V][H]
    w.wtb = off;  off = al(off + (int64_t)H * w.Vpad * 2);       // W^T bf16 [H][Vpad]
    w.hstb = off; off = al(off + (int64_t)H * w.Npad * 2);       // Hs^T bf16 [H][Npad]
    w.gb = off;   off = al(off + (int64_t)N * w.Vpad * 2);       // G bf16 [N][Vpad]
    w.gtb = off;  off = al(off + (int64_t)V * w.Npad * 2);       // G^T bf16 [V][Npad]
    // dHs: K = Vpad is long and M x N small -> deal K over enough slices to fill the chip
    const long tiles = (long)sat_cdiv(N, 128) * sat_cdiv(H, 128);
    int ks = (int)(256 / (tiles > 0 ? tiles : 1));
    const int nk = w.Vpad / 64;
    if (ks > nk / 8) ks = nk / 8;
    w.ks = ks < 1 ? 1 : (ks > 16 ? 16 : ks);
    w.slabs = off; off = al(off + (w.ks > 1 ? (int64_t)w.ks * N * H * 4 : 0));
    w.psum = off; off = al(off + (int64_t)(w.Npad / 64) * V * 4);  // per-row-tile column sums of G (bias gradient)
    w.total = off;
    return w;
}

bool vocab_bf16_ok(int N, int H, int V) { return N >= 1 && H >= 64 && (H % 64) == 0 && V >= 4 && (V % 4) == 0 && V <= 12288; }

}  // namespace

extern "C" int sat_sum_slabs_f32(const float* in, int nslab, int64_t slab_stride, int64_t n, float* out, sat_stream_t stream);

// C[M,N] f32 = A[M,K] B[N,K]^T (+ bias[N]): bf16 operands, both K-contiguous, K % 64 == 0 (pad with zeros), N % 4 == 0.
// ksplit > 1: slice z of the K range writes C + z * slab_stride (no bias); the caller sums the slabs (sat_sum_slabs_f32).
extern "C" int sat_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                                int M, int N, int K, int ksplit, int64_t slab_stride, sat_stream_t stream) {
    return launch_gemm((const bf16_t*)A, (long)lda, (const bf16_t*)B, (long)ldb, C, (long)ldc, bias, M, N, K, ksplit, (long)slab_stride,
                       (hipStream_t)stream);
}

// out bf16 [C][ldo] = in^T for in f32 [R][ldi]; columns [R, ldo) zero-filled (ldo % 4 == 0, ldo >= R)
extern "C" int sat_transpose_f32_bf16(const float* in, int64_t ldi, int R, int C, void* out, int64_t ldo, sat_stream_t stream) {
    if (!in || !out || R < 1 || C < 1 || ldo < R || (ldo % 4) || ldi < C) return SAT_ERR_ARG;
    hipLaunchKernelGGL((transpose_bf16_kernel<float>), dim3(sat_cdiv(C, 64), sat_cdiv(ldo, 64)), dim3(256), 0, (hipStream_t)stream, in,
                       (long)ldi, R, C, (bf16_t*)out, (long)ldo, (float*)nullptr);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// C[M,N] f32 = op(A) op(B)^T (+ bias + bias2) on the bf16 matrix pipe from F32 operands: each operand is cast (row-major
// [rows][K]) or transposed ([K][rows], "k-major") into a bf16 [rows][Kpad] copy in `scratch` (Kpad = K rounded up to 64, zero
// padded), then the NT kernel runs.  scratch: (M + N) * Kpad * 2 bytes, 256-byte aligned.  N % 4 == 0, lda / ldb % 4 == 0.
int64_t sat_gemm_mixed_scratch_bytes(int M, int N, int K) {
    const int64_t kp = pad64(K);
    return (((int64_t)M * kp * 2 + 255) / 256 * 256) + (((int64_t)N * kp * 2 + 255) / 256 * 256);
}
int sat_gemm_mixed_nt(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor, float* C, long ldc,
                      const float* bias, const float* bias2, int M, int N, int K, void* scratch, int64_t scratch_bytes, hipStream_t s) {
    if (!A || !B || !C || !scratch || M < 1 || N < 1 || K < 1) return SAT_ERR_ARG;
    if ((N % 4) || (ldc % 4)) return SAT_ERR_UNSUPPORTED;
    if (scratch_bytes < sat_gemm_mixed_scratch_bytes(M, N, K) || (((uintptr_t)scratch) & 255)) return SAT_ERR_WORKSPACE;
    const int kp = pad64(K);
    bf16_t* ab = (bf16_t*)scratch;
    bf16_t* bb = (bf16_t*)((char*)scratch + (((int64_t)M * kp * 2 + 255) / 256 * 256));
    auto prep = [&](const float* src, long ld, int kmajor, int rows, bf16_t* dst) {
        if (kmajor) {       // src [K][rows] -> dst [rows][kp]
            hipLaunchKernelGGL((transpose_bf16_kernel<float>), dim3(sat_cdiv(rows, 64), sat_cdiv(kp, 64)), dim3(256), 0, s, src, ld, K, rows,
                               dst, (long)kp, (float*)nullptr);
        } else {            // src [rows][K] -> dst [rows][kp]
            long g = ((long)rows * kp / 8 + 255) / 256;
            hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3((unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g))), dim3(256), 0, s, src, ld, rows, K,
                               dst, (long)kp);
        }
    };
    prep(A, lda, a_kmajor, M, ab);
    SAT_LAUNCH_CHECK();
    prep(B, ldb, b_kmajor, N, bb);
    SAT_LAUNCH_CHECK();
    return launch_gemm(ab, kp, bb, kp, C, ldc, bias, M, N, kp, 1, 0, s, bias2);
}

extern "C" int64_t sat_vocab_bf16_ws_bytes(int N, int H, int V) { return vocab_bf16_ok(N, H, V) ? vocab_ws(N, H, V).total : 0; }

// models.py:53 + train.py:143 in the bf16 throughput mode: logits (f32, from bf16 operands) + mean CE; d loss / d logits is
// left in the workspace as bf16 for sat_vocab_ce_bwd_bf16, next to the bf16 operand copies both calls share.
extern "C" int sat_vocab_ce_fwd_bf16(const float* Hs, const float* w, const float* b, const int64_t* targets, int N, int H, int V,
                                     float inv_denom, float* logits, int64_t ldl, float* row_loss, float* loss_out, void* workspace,
                                     int64_t ws_bytes, sat_stream_t stream) {
    if (!Hs || !w || !b || !targets || !logits || !row_loss || !workspace || ldl < V || (ldl % 4)) return SAT_ERR_ARG;
    if (!vocab_bf16_ok(N, H, V)) return SAT_ERR_UNSUPPORTED;
    const VocabWs L = vocab_ws(N, H, V);
    if (ws_bytes < L.total) return SAT_ERR_WORKSPACE;
    if ((((uintptr_t)workspace) & 255) || (((uintptr_t)logits) & 15)) return SAT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    bf16_t* hsb = (bf16_t*)(ws + L.hsb); bf16_t* wb = (bf16_t*)(ws + L.wb); bf16_t* wtb = (bf16_t*)(ws + L.wtb);
    bf16_t* hstb = (bf16_t*)(ws + L.hstb); bf16_t* gb = (bf16_t*)(ws + L.gb);
    auto egrid = [](long n8) { long g = (n8 + 255) / 256; return dim3((unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g))); };
    hipLaunchKernelGGL(cast_rows_bf16_kernel, egrid((long)N * H / 8), dim3(256), 0, s, Hs, (long)H, N, H, hsb, (long)H);
    SAT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cast_rows_bf16_kernel, egrid((long)V * H / 8), dim3(256), 0, s, w, (long)H, V, H, wb, (long)H);
    SAT_LAUNCH_CHECK();
    hipLaunchKernelGGL((transpose_bf16_kernel<float>), dim3(sat_cdiv(H, 64), sat_cdiv(L.Vpad, 64)), dim3(256), 0, s, w, (long)H, V, H, wtb,
                       (long)L.Vpad, (float*)nullptr);
    SAT_LAUNCH_CHECK();
    hipLaunchKernelGGL((transpose_bf16_kernel<float>), dim3(sat_cdiv(H, 64), sat_cdiv(L.Npad, 64)), dim3(256), 0, s, Hs, (long)H, N, H, hstb,
                       (long)L.Npad, (float*)nullptr);
    SAT_LAUNCH_CHECK();
    SAT_TRY(launch_gemm(hsb, H, wb, H, logits, (long)ldl, b, N, V, H, 1, 0, s));
    hipLaunchKernelGGL(ce_rows_bf16grad_kernel, dim3(N), dim3(256), 0, s, logits, (long)ldl, targets, V, inv_denom, row_loss, gb,
                       (long)L.Vpad);
    SAT_LAUNCH_CHECK();
    if (loss_out) {
        hipLaunchKernelGGL(sum_scale_b_kernel, dim3(1), dim3(256), 0, s, row_loss, N, inv_denom, loss_out);
        SAT_LAUNCH_CHECK();
    }
    return SAT_OK;
}

// train.py:144 for the vocab projection: dW[V,H], db[V], dHs[N,H] (all f32) from the workspace sat_vocab_ce_fwd_bf16 left
extern "C" int sat_vocab_ce_bwd_bf16(int N, int H, int V, float* dw, float* db, float* dHs, void* workspace, int64_t ws_bytes,
                                     sat_stream_t stream) {
    if (!dw || !db || !dHs || !workspace) return SAT_ERR_ARG;
    if (!vocab_bf16_ok(N, H, V)) return SAT_ERR_UNSUPPORTED;
    const VocabWs L = vocab_ws(N, H, V);
    if (ws_bytes < L.total) return SAT_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    bf16_t* wtb = (bf16_t*)(ws + L.wtb); bf16_t* hstb = (bf16_t*)(ws + L.hstb);
    bf16_t* gb = (bf16_t*)(ws + L.gb); bf16_t* gtb = (bf16_t*)(ws + L.gtb);
    float* slabs = (float*)(ws + L.slabs);
    // G -> G^T; the same pass leaves per-row-tile column sums of G, whose sum over the tiles is the bias gradient
    float* psum = (float*)(ws + L.psum);
    const int rt = sat_cdiv(L.Npad, 64);
    hipLaunchKernelGGL((transpose_bf16_kernel<bf16_t>), dim3(sat_cdiv(V, 64), rt), dim3(256), 0, s, gb, (long)L.Vpad, N, V, gtb,
                       (long)L.Npad, psum);
    SAT_LAUNCH_CHECK();
    SAT_TRY(sat_sum_slabs_f32(psum, rt, (int64_t)V, (int64_t)V, db, stream));
    SAT_TRY(launch_gemm(gtb, L.Npad, hstb, L.Npad, dw, H, nullptr, V, H, L.Npad, 1, 0, s));               // dW = G^T Hs
    if (L.ks == 1) return launch_gemm(gb, L.Vpad, wtb, L.Vpad, dHs, H, nullptr, N, H, L.Vpad, 1, 0, s);     // dHs = G W
    SAT_TRY(launch_gemm(gb, L.Vpad, wtb, L.Vpad, slabs, H, nullptr, N, H, L.Vpad, L.ks, (long)N * H, s));
    const int nslab = sat_cdiv(L.Vpad / 64, sat_cdiv(L.Vpad / 64, L.ks));
    return sat_sum_slabs_f32(slabs, nslab, (int64_t)N * H, (int64_t)N * H, dHs, stream);
}
