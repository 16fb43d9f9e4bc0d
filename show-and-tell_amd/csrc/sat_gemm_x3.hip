// f32 GEMM on the bf16 matrix pipe by a THREE-WAY SPLIT of both operands (beam decode: the vocab projection `self.linear(hiddens)`,
// models.py:53 / :63, of 320 hypothesis rows per step -- half of the decode step's time on the exact-f32 pipe).
//
//   C[M,N] = A[M,K] * W[N,K]^T + bias,   A, W, C f32
//
// Every f32 operand x is written as hi + mid + lo with three bf16 terms (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); the
// two subtractions are exact in f32, so |x - hi - mid - lo| <= 2^-25 |x|).  A product of two bf16 numbers is exact in f32, and
// v_mfma_f32_32x32x16_bf16 accumulates in f32, so with the six products of weight >= 2^-16
//     hi*hi  +  (hi*mid + mid*hi + hi*lo + lo*hi + mid*mid)
// the result carries the accuracy of an f32 GEMM (the dropped products are below 2^-24 of |a||b|) at 6 / 16 of the f32 pipe's cost
// per product: the bf16 pipe is 16 x the f32 one.  hi*hi and the five corrections go to SEPARATE accumulators (added once, at the end),
// so that the small terms are not rounded away against the large sum.  Not the bit pattern of the f32-pipe GEMM (another summation
// order, like any two f32 GEMMs), hence used where ids / scores are checked against an oracle to a tolerance: the beam decode's
// projection (tests/test_gpu_gemm_x3.py: against f64, relative to sum |a||b|; tests/test_gpu_parity.py: decode ids bit-exact).
//
// Kernel = conv_aw_kernel's structure (sat_conv_aw.inc): 128 x 128 tiles, four waves, wave w owns 32 columns x 128 rows; the WEIGHTS are
// split and put in MFMA fragment order ONCE per decode call (sat_gemm_f32x3_pack: 3 x [N][K] bf16) and stream straight into registers
// (12 KB per wave and K-step of 64, two K-steps ahead); the activations go global (f32) -> registers -> split -> three XOR-swizzled LDS
// row images per K-step (two buffers, one raw barrier per K-step); 96 MFMAs per wave and K-step.  Column tiles are the slow index of the
// XCD-aware tile map: the row tiles that share a weight slice run on one XCD and hit in its L2.
#include "sat_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct X3Args {
    const float* A;
    long lda;
    const char* Bp;          // packed split weights: [N128 / 32][K / 64][3 splits][4 ks][64 lanes][8 bf16]
    const float* bias;
    float* C;
    long ldc;
    int M, N, K, tiles_m;
    long a_bytes;
};

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// (x0, x1) -> the three bf16 pairs
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    f32x2 v = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 q = {r[0] - __uint_as_float(mid << 16), r[1] - __uint_as_float(mid & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

// W f32 [N][K] -> packed[nb][cb][s][ks][lane = h*32 + r][e] = split_s(W[32 nb + r][64 cb + 16 ks + 8 h + e]); columns >= N are zeros
__global__ __launch_bounds__(256) void x3_pack_kernel(const float* __restrict__ W, u32x4* __restrict__ P, int N, int K, long total) {
    const int ncb = K >> 6;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 3);
        const long blk = idx >> 8;
        const int cb = (int)(blk % ncb);
        const long nb = blk / ncb;
        const long col = 32 * nb + (lane & 31);
        const int k0 = 64 * cb + 16 * ks + 8 * (lane >> 5);
        u32x4 hi = {0u, 0u, 0u, 0u}, mid = hi, lo = hi;
        if (col < N) {
            const f32x4 a = *(const f32x4*)(W + col * K + k0), b = *(const f32x4*)(W + col * K + k0 + 4);
            unsigned th, tm, tl;
            split2(a[0], a[1], th, tm, tl); hi[0] = th; mid[0] = tm; lo[0] = tl;
            split2(a[2], a[3], th, tm, tl); hi[1] = th; mid[1] = tm; lo[1] = tl;
            split2(b[0], b[1], th, tm, tl); hi[2] = th; mid[2] = tm; lo[2] = tl;
            split2(b[2], b[3], th, tm, tl); hi[3] = th; mid[3] = tm; lo[3] = tl;
        }
        u32x4* dst = P + ((blk * 3) * 4 + ks) * 64 + lane;      // split s: + s * 4 * 64 chunks
        dst[0] = hi;
        dst[4 * 64] = mid;
        dst[8 * 64] = lo;
    }
}

// BM = 128: one workgroup per CU (96 KB of LDS, 128 + 234 registers); BM = 64: two per CU (48 KB, <= 256 registers) -- twice the
// weight bytes through L2 -> CU per flop, but two waves per SIMD cover each other's barriers and split work, and 320 rows are exactly 5 tiles
template <int BM>
__global__ __launch_bounds__(256, BM == 128 ? 1 : 2) void gemm_x3_kernel(const X3Args p) {
    constexpr int BN = 128, NT = 256, TI = BM / 32;
    constexpr int ABUF = BM * 128;                          // one K-step of one split image: BM rows x 64 bf16
    constexpr int NJ = BM / 32, RJ = 32;                    // 8-element chunks of a stage per thread, rows RJ apart
    constexpr int CROW = BN * 4 + 16;                       // epilogue: f32 tile rows
    constexpr int SMEM = (6 * ABUF > BM * CROW) ? 6 * ABUF : BM * CROW;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_n = swz / p.tiles_m, tile_m = swz - tile_n * p.tiles_m;      // the row tiles of a column tile share an XCD's L2
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K >> 6;                                  // K-steps of 64: even, >= 4 (the launcher checks)

    // ---- this thread's share of an A stage: chunk slot lc (8 k-elements = 32 bytes of f32) of rows r0 + 32 j; it fetches logical
    //      chunk lc ^ ((row >> 1) & 7), so each split image is the ring kernel's XOR-swizzled row image.  Rows past M: an offset past
    //      the buffer (reads zeros) ----
    const int lc = tid & 7, r0 = tid >> 3;
    const int sw0 = (r0 >> 1) & 7;
    const __amdgpu_buffer_rsrc_t asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    int a_voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int grow = m0 + r0 + RJ * j;
        a_voff[j] = grow < p.M ? (int)(((long)grow * p.lda + ((lc ^ sw0) << 3)) * 4) : 0x7fffffe0;
    }
    u32x4 ax[2][NJ][2];
    auto load_stage = [&](int g, u32x4 (&d)[NJ][2]) {
        const int gs = g * 256;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            d[j][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff[j], gs, 0));
            d[j][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff[j] + 16, gs, 0));
        }
    };
    // ---- this wave's weight stream: nk K-steps of 12 KB (3 splits x 4 substeps x 1 KB), two K-steps ahead ----
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.Bp + (long)((n0 >> 5) + wave) * nk * 12288), 0, nk * 12288, 0x00020000);
    const int wlane = lane * 16;
    auto load_b = [&](int g, int s, int ks) {
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wsrc, wlane, g * 12288 + s * 4096 + ks * 1024, 0));
    };

    load_stage(0, ax[0]);
    load_stage(1, ax[1]);
    bf16x8 bq[2][3][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bq[g][s][ks] = load_b(g, s, ks);

    f32x16 acc0[TI], acc1[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc0[i][e] = 0.0f;
            acc1[i][e] = 0.0f;
        }

    // a landed stage: split into the three bf16 terms, 16 bytes per row and image into slot lc of rows r0 + 32 j of buffer `buf`
    auto store_stage = [&](int buf, u32x4 (&sx)[NJ][2]) {
        char* at0 = smem + buf * 3 * ABUF + r0 * 128 + (lc << 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            u32x4 hi, mid, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 w = sx[j][q >> 1];
                unsigned th, tm, tl;
                split2(__uint_as_float(w[2 * (q & 1)]), __uint_as_float(w[2 * (q & 1) + 1]), th, tm, tl);
                hi[q] = th;
                mid[q] = tm;
                lo[q] = tl;
            }
            *(u32x4*)(at0 + j * (RJ * 128)) = hi;
            *(u32x4*)(at0 + ABUF + j * (RJ * 128)) = mid;
            *(u32x4*)(at0 + 2 * ABUF + j * (RJ * 128)) = lo;
        }
    };

    int a_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_off[ks] = r * 128 + (((2 * ks + h) ^ ((r >> 1) & 7)) << 4);

    store_stage(0, ax[0]);
    load_stage(2, ax[0]);

    // K-step g, U = g & 1 (compile time): stage g sits in LDS buffer U; stage g + 1 leaves register slot U ^ 1 for the other buffer
    // (its last readers, K-step g - 1, are behind this step's barrier) and the slot takes stage g + 3; the weight registers of K-step
    // g (slot U) are refilled with K-step g + 2 as they are used
    auto step = [&](int g, auto u_tag, auto st_tag, auto al_tag, auto bl_tag) {
        constexpr int U = decltype(u_tag)::value;
        constexpr bool ST = decltype(st_tag)::value, AL = decltype(al_tag)::value, BL = decltype(bl_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (ST) store_stage(U ^ 1, ax[U ^ 1]);
        if constexpr (AL) load_stage(g + 3, ax[U ^ 1]);
        const char* st = smem + U * 3 * ABUF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fh[TI], fm[TI], fl[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                fh[i] = *(const bf16x8*)(st + a_off[ks] + i * 4096);
                fm[i] = *(const bf16x8*)(st + ABUF + a_off[ks] + i * 4096);
                fl[i] = *(const bf16x8*)(st + 2 * ABUF + a_off[ks] + i * 4096);
            }
            const bf16x8 bh = bq[U][0][ks], bm = bq[U][1][ks], bl = bq[U][2][ks];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                acc0[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[i], bh, acc0[i], 0, 0, 0);
                acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[i], bh, acc1[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[i], bl, acc1[i], 0, 0, 0);
                acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm[i], bm, acc1[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm[i], bh, acc1[i], 0, 0, 0);
                acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[i], bm, acc1[i], 0, 0, 0);
            }
            if constexpr (BL) {
                bq[U][0][ks] = load_b(g + 2, 0, ks);
                bq[U][1][ks] = load_b(g + 2, 1, ks);
                bq[U][2][ks] = load_b(g + 2, 2, ks);
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    int g = 0;
    for (; g < nk - 4; g += 2) {
        step(g, std::integral_constant<int, 0>{}, T{}, T{}, T{});
        step(g + 1, std::integral_constant<int, 1>{}, T{}, T{}, T{});
    }
    step(g, std::integral_constant<int, 0>{}, T{}, T{}, T{});            // K-step nk - 4: requests stage nk - 1, the last
    step(g + 1, std::integral_constant<int, 1>{}, T{}, F{}, T{});        // nk - 3
    step(g + 2, std::integral_constant<int, 0>{}, T{}, F{}, F{});        // nk - 2: stores stage nk - 1
    step(g + 3, std::integral_constant<int, 1>{}, F{}, F{}, F{});        // nk - 1

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // every wave's LDS traffic is done before the buffers become the tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- epilogue: hi*hi + corrections + bias, f32 tile through LDS, 16-byte stores ----
    const int colw = wave * 32 + r;
    const float badd = (p.bias && n0 + colw < p.N) ? p.bias[n0 + colw] : 0.0f;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            *(float*)(smem + row * CROW + colw * 4) = (acc0[i][e] + acc1[i][e]) + badd;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    constexpr int ITS = BM * (BN / 4) / NT;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int qid = tid + it * NT;
        const int row = qid / (BN / 4), cc = qid % (BN / 4);
        const int grow = m0 + row, gcol = n0 + cc * 4;
        if (grow < p.M && gcol < p.N) *(u32x4*)(p.C + (long)grow * p.ldc + gcol) = *(const u32x4*)(smem + row * CROW + cc * 16);
    }
}

}  // namespace

// bytes of the split, fragment-ordered copy of W [N][K] (N rounded up to 128 columns); 0: shape not supported (K % 128, K >= 256)
extern "C" int64_t sat_gemm_f32x3_packed_bytes(int N, int K) {
    if (N <= 0 || K < 256 || (K & 127)) return 0;
    return (int64_t)sat_cdiv(N, 128) * 128 * K * 3 * 2;
}

extern "C" int sat_gemm_f32x3_pack(const float* W, int N, int K, void* packed, sat_stream_t stream) {
    if (!W || !packed || sat_gemm_f32x3_packed_bytes(N, K) == 0 || (((uintptr_t)W | (uintptr_t)packed) & 15)) return SAT_ERR_ARG;
    const long total = (long)sat_cdiv(N, 128) * 4 * (K >> 6) * 256;      // one thread per (32-column block, K-step, substep, lane)
    int grid = sat_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(x3_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, (u32x4*)packed, N, K, total);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// C[M][ldc] = A[M][lda] * W^T + bias with W given as its packed split copy; N % 4 == 0, ldc % 4 == 0, lda % 4 == 0, K % 128 == 0, K >= 256
extern "C" int sat_gemm_f32x3(const float* A, int64_t lda, const void* packed, const float* bias, float* C, int64_t ldc, int M, int N,
                              int K, sat_stream_t stream) {
    if (!A || !packed || !C || M <= 0 || N <= 0) return SAT_ERR_ARG;
    if (sat_gemm_f32x3_packed_bytes(N, K) == 0 || (N & 3) || (ldc & 3) || (lda & 3) || lda < K || ldc < N) return SAT_ERR_UNSUPPORTED;
    if ((((uintptr_t)A | (uintptr_t)packed | (uintptr_t)C) & 15) || (long)M * lda * 4 >= 0x7fffffe0L) return SAT_ERR_ARG;
    X3Args a = {};
    a.A = A; a.lda = lda; a.Bp = (const char*)packed; a.bias = bias; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K;
    a.a_bytes = (long)M * lda * 4;
    // 64-row tiles (two workgroups per CU) unless 128-row ones waste no more rows and still fill the chip's 256 CUs
    static const int bm_env = [] { const char* e = getenv("SAT_X3_BM"); return e ? atoi(e) : 0; }();
    const long t128 = (long)sat_cdiv(M, 128) * sat_cdiv(N, 128);
    const bool big = bm_env ? bm_env == 128 : (sat_cdiv(M, 128) * 128 == sat_cdiv(M, 64) * 64 && t128 >= 512);
    const dim3 block(256);
    if (big) {
        a.tiles_m = sat_cdiv(M, 128);
        hipLaunchKernelGGL(gemm_x3_kernel<128>, dim3(a.tiles_m * sat_cdiv(N, 128)), block, 0, (hipStream_t)stream, a);
    } else {
        a.tiles_m = sat_cdiv(M, 64);
        hipLaunchKernelGGL(gemm_x3_kernel<64>, dim3(a.tiles_m * sat_cdiv(N, 128)), block, 0, (hipStream_t)stream, a);
    }
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}
