// bf16 implicit-GEMM convolution, deep-prefetch version: the dominant kernel of the Show-and-Tell step
// (`self.resnet(images)`, models.py:27).
//
//   out[m, co] = sum_{kh,kw,ci} in[n, ho*s+kh-p, wo*s+kw-p, ci] * w[co, kh, kw, ci]      m = (n, ho, wo), NHWC
//
// ResNet bottleneck layers at batch 64 sit at the HBM/MFMA ridge (1x1 conv, C_in*C_out/(C_in+C_out) ~ 200
// FLOP/B) and a 128-wide tile consumes a K-step in ~0.1 us, so the kernel lives or dies by bytes in flight:
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 16 B/lane, no VGPR staging), into a ring of
//     S stages; S-1 K-steps (24-32 KB each) are in flight per workgroup behind a counted s_waitcnt vmcnt(N)
//     and ONE raw s_barrier per K-step;
//   * LDS rows are 128 B (one K-step) unpadded -- an LDS-DMA write is lane-linear -- and bank conflicts are
//     removed by an XOR swizzle of the 16-byte chunk index, chunk ^ ((row>>1)&7), applied to the per-lane
//     SOURCE address and to the fragment read (ds_read_b128 stays conflict free);
//   * zero padding / tile tails: the lane's source pointer is redirected to a 16-byte zero word; per-row tap
//     validity is a bit mask computed once, so the K loop's address work is ~10 VALU per LDS-DMA piece;
//   * 8 waves per workgroup (two per SIMD): one wave's address/LDS-read phase hides under its partner's MFMAs;
//   * epilogue: f32 accumulators -> per-tile BatchNorm column sums (fixed order) and a bf16 tile staged through
//     LDS so global stores are full 16-byte-per-lane rows;
//   * blockIdx -> tile map is XCD-aware (tiles sharing an activation panel share an L2).
#include "sat_internal.h"
#include <hip/hip_ext.h>

// diagnostics (sat_run_ops_timed): when armed, the NEXT conv launch of this thread records its own begin / end
// timestamps into these events (hipExtLaunchKernelGGL: the dispatch packet's timestamps, what rocprofv3 reports)
static thread_local hipEvent_t t_ev_start = nullptr, t_ev_stop = nullptr;
void sat_conv_arm_timer(hipEvent_t start, hipEvent_t stop) { t_ev_start = start; t_ev_stop = stop; }
// the armed timer, consumed (the fused conv3 launch of sat_conv3_fused.hip reports its own span the same way)
void sat_conv_take_timer(hipEvent_t* start, hipEvent_t* stop) {
    *start = t_ev_start; *stop = t_ev_stop;
    t_ev_start = t_ev_stop = nullptr;
}

namespace {

__device__ u32x4 g_zero16;   // zero-initialised device global: the source of every padded / out-of-range chunk

struct ConvArgs {
    const bf16_t* A;
    const bf16_t* B;
    const bf16_t* Bp;        // the weights once more, in MFMA fragment order (sat_conv_pack_weights; conv_pw_kernel / conv_aw_kernel), or NULL
    long in_bytes;           // bytes of one group's input tensor (conv_aw_kernel reads it through a bounds-checked buffer)
    bf16_t* C;
    float* stat_partial;
    int M, N, K;
    long ldb, ldc;
    int Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad, padw;
    long sN, sH, sW;
    int tiles_n;
    int linear;     // 1x1 / stride 1 / no padding on a dense NHWC tensor: output row m reads input pixel m
    // training-mode BatchNorm statistics, atomic form: fixed-point column sums (sum, sum of squares) added with 64-bit
    // INTEGER atomics into acc[2][N].  Integer addition is associative, so the totals are bitwise reproducible
    // whatever the arrival order; the consumer kernel (bn_act / maxpool / the next conv) turns them into scale/shift itself,
    // which removes the separate finalize launch.  Used when there are few M-tiles (<= ~400 adds per word).
    long long* acc;
    // Input-side fusion: the A operand is the RAW output of the previous conv and its BatchNorm + ReLU is applied to the
    // staged operand in LDS, so the normalised tensor never exists in HBM (1x1 convs: every landed stage of the ring kernel,
    // the register panel of conv_xp_kernel; 3x3: the LDS-resident patch of conv_pr_kernel).  The (scale, shift) table comes
    // precomputed (in_scale/in_shift) or is derived here from the previous conv's integer sums (in_acc).
    const float* in_scale;
    const float* in_shift;
    const long long* in_acc;      // [2][Cin], this step's parity
    long long* in_acc_clear;      // other parity (cleared by workgroup 0) or NULL
    const float* in_gamma;
    const float* in_beta;
    float* in_running_mean;
    float* in_running_var;
    double in_count;
    double in_inv;           // 1 / (2^22 * in_count) (conv_xp_kernel / conv_pr_kernel: the division happens on the host)
    float in_momentum, in_eps;
    int in_affine;
    // Output-side fusion (inference: BatchNorm is a fixed per-channel affine): out = [relu](acc*out_scale[n] +
    // out_shift[n] [+ residual[m][n]]) straight from the f32 accumulators -- no normalise / add launch at all.
    const float* out_scale;
    const float* out_shift;
    const bf16_t* residual;  // [M][ldc] like C, or NULL
    int out_relu;
    int ct;                  // conv_xp_kernel: column tiles per workgroup
    // GROUPED launch (sat_op.groups > 1, blockIdx.y = group): G independent batches with the SAME weights in one launch -- every
    // per-batch pointer moves by its group stride (elements of its own type), nothing else changes: each group is the very
    // instruction sequence of the ungrouped launch on its batch (same tiles, same summation order, own statistics)
    long gs_a, gs_c, gs_partial, gs_acc, gs_in_acc, gs_in_run;
    long gs_out_tab;         // SAT_CONV_GROUP_TABLE: out_scale / out_shift move by this many floats per group (the table SAT_OP_BN_FROM_GRAM writes), else 0
    // SAT_CONV_IN_RESIDUAL (conv_ay_kernel): the operand is relu(bn(A) + in_res), built on its way to LDS and written to Y as well
    // (the previous bottleneck's normalise + add + ReLU inside this bottleneck's conv1); both shaped like A, group stride gs_a
    const bf16_t* in_res;
    bf16_t* Y;
};
// fixed-point scale of the atomic statistics is SAT_STAT_SCALE (sat_internal.h)
constexpr double kStatScale = SAT_STAT_SCALE;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 0 || N == 2 || N == 3 || N == 4 || N == 5 || N == 6 || N == 7 || N == 8 || N == 9 || N == 10 || N == 12 ||
                      N == 15 || N == 16 || N == 17 || N == 18 || N == 19 || N == 20 || N == 21 || N == 22 || N == 24,
                  "add the vmcnt literal");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    if constexpr (N == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
    if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    if constexpr (N == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if constexpr (N == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

// BN: tile width (128/64); S: ring stages; NW: waves (4/8); UNIFORM: Cin % 64 == 0 (a K-step never straddles a tap);
// SPEC: wave specialisation -- waves 0..NW/2-1 own the accumulators (ds_read + MFMA only), waves NW/2..NW-1 only
// issue the LDS-DMA (address walk + global_load_lds).  Each SIMD then holds one consumer and one loader, whose
// instruction streams are complementary (matrix pipe vs memory issue) instead of two lock-stepped copies.
// PF: consumers prefetch the first MFMA fragments of K-step kt+1 during the last MFMAs of K-step kt, so no LDS read
// latency is exposed behind the per-K-step barrier; the loaders' counted wait then has to cover K-step kt+1 at
// barrier kt (one in-flight K-step fewer than the ring could hold, hence S >= 4).
// BM: tile rows (128 or 64).
// grouped launches: group g = blockIdx.y works on its own batch -- per-batch pointers shifted by the group strides of ConvArgs
__device__ __forceinline__ ConvArgs group_args(const ConvArgs& q) {
    ConvArgs p = q;
    const long g = blockIdx.y;
    if (g) {
        p.A += g * q.gs_a;
        p.C += g * q.gs_c;
        if (p.residual) p.residual += g * q.gs_c;
        if (p.in_res) { p.in_res += g * q.gs_a; p.Y += g * q.gs_a; }
        if (p.out_scale) { p.out_scale += g * q.gs_out_tab; p.out_shift += g * q.gs_out_tab; }
        if (p.stat_partial) p.stat_partial += g * q.gs_partial;
        if (p.acc) p.acc += g * q.gs_acc;
        if (p.in_acc) p.in_acc += g * q.gs_in_acc;
        if (p.in_acc_clear) p.in_acc_clear += g * q.gs_in_acc;
        if (p.in_running_mean) { p.in_running_mean += g * q.gs_in_run; p.in_running_var += g * q.gs_in_run; }
    }
    return p;
}

template <int BN, int S, int NW, bool UNIFORM, bool SPEC = false, bool PF = false, int BM = 128>
__global__ __launch_bounds__(NW * 64) void conv_glds_kernel(const ConvArgs p_) {
    const ConvArgs p = group_args(p_);
    constexpr int BK = 64, NT = NW * 64;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int LW = SPEC ? NW / 2 : NW;                // waves that issue the LDS-DMA
    constexpr int CW = SPEC ? NW / 2 : NW;                // waves that own accumulators
    // consumer wave grid: 4 waves: 2x2;  8 waves: 4(M)x2(N) for the 128x64 tile, 2(M)x4(N) otherwise
    constexpr int WGM = (CW == 4) ? 2 : ((BM == 128 && BN == 64) ? 4 : 2);
    constexpr int WGN = CW / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int NAI = BM / 8 / LW, NBI = BN / 8 / LW;   // LDS-DMA pieces (8 rows x 128 B) per loader wave per stage
    constexpr int LPW = NAI + NBI;
    constexpr int D = S - 1;                               // K-steps kept in flight
    constexpr int WAITN = LPW * (PF ? D - 2 : D - 1);      // loader pieces that may still be in flight at a barrier
    static_assert(!PF || S >= 4, "fragment prefetch needs one more landed stage");
    constexpr int CROW = BN * 2 + 16;                      // bf16 C-tile row stride in LDS (epilogue)
    static_assert(BM * CROW + 4 * WGM * BN * 4 <= S * STAGE, "epilogue tile + stat scratch must fit the ring");
    static_assert(TM >= 1 && TN >= 1 && NAI >= 1 && NBI >= 1, "bad tile/wave split");
    constexpr int TAB_BYTES = 2 * 512 * 4;                 // input-BN (scale, shift) table: up to 512 input channels
    __shared__ __attribute__((aligned(16))) char smem[S * STAGE + TAB_BYTES];   // ONE LDS object (ring + table)
    float* in_tab = (float*)(smem + S * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = SPEC ? (wave >= CW) : true;
    const bool is_consumer = SPEC ? (wave < CW) : true;
    const int lw = SPEC ? (is_loader ? wave - CW : 0) : wave;      // loader index
    const int cw = is_consumer ? wave : 0;                          // consumer index
    const int wm = cw / WGN, wn = cw % WGN;
    const int r = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_m = swz / p.tiles_n, tile_n = swz - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const bf16_t* zero = (const bf16_t*)&g_zero16;

    // ---- K-invariant per-lane state: lane covers row (l>>3) of each 8-row LDS-DMA piece, LDS chunk slot l&7;
    //      it fetches logical chunk (l&7) ^ ((row>>1)&7) so that the LDS image is XOR-swizzled ----
    const bf16_t* a_ptr[NAI];       // pixel (n, ho*s-p, wo*s-p), channel = this lane's chunk
    unsigned a_mask[NAI];           // UNIFORM: bit t = tap t in bounds for this row
    int a_hi0[NAI], a_wi0[NAI], a_c[NAI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int row = lw * (NAI * 8) + i * 8 + (lane >> 3);
        a_c[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        const int grow = m0 + row;
        a_mask[i] = 0u;
        if (grow < p.M && p.linear) {
            a_hi0[i] = 0; a_wi0[i] = 0;
            a_ptr[i] = p.A + ((long)grow * p.sW + a_c[i]);
            a_mask[i] = 1u;
        } else if (grow < p.M) {
            const int hw = p.Hout * p.Wout;
            const int n = grow / hw;
            const int rem = grow - n * hw;
            const int ho = rem / p.Wout;
            const int wo = rem - ho * p.Wout;
            a_hi0[i] = ho * p.stride - p.pad;
            a_wi0[i] = wo * p.stride - p.padw;
            a_ptr[i] = p.A + ((long)n * p.sN + (long)a_hi0[i] * p.sH + (long)a_wi0[i] * p.sW + a_c[i]);
            if constexpr (UNIFORM) {
                for (int kh = 0; kh < p.KH; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw) {
                        const bool in = ((unsigned)(a_hi0[i] + kh) < (unsigned)p.Hin) && ((unsigned)(a_wi0[i] + kw) < (unsigned)p.Win);
                        a_mask[i] |= (in ? 1u : 0u) << (kh * p.KW + kw);
                    }
            }
        } else {
            a_hi0[i] = -(1 << 28); a_wi0[i] = -(1 << 28); a_ptr[i] = zero;
        }
    }
    const bf16_t* b_ptr[NBI];
    int b_c[NBI];
    bool b_ok[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int row = lw * (NBI * 8) + i * 8 + (lane >> 3);
        b_c[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        const int gn = n0 + row;
        b_ok[i] = gn < p.N;
        b_ptr[i] = b_ok[i] ? p.B + ((long)gn * p.ldb + b_c[i]) : zero;
    }
    const int nk = (p.K + BK - 1) / BK;
    // linear (1x1) walk: per-lane pointers advance by one K-step; invalid rows stay on the zero word
    int a_step[NAI], b_step[NBI];
    const bf16_t* b_walk[NBI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) a_step[i] = (m0 + lw * (NAI * 8) + i * 8 + (lane >> 3) < p.M) ? BK : 0;
#pragma unroll
    for (int i = 0; i < NBI; ++i) { b_walk[i] = b_ptr[i]; b_step[i] = b_ok[i] ? BK : 0; }

    // scalar walk of the K axis for the UNIFORM path (no division in the loop)
    int is_kt = 0, is_tap = 0, is_cb = 0, is_kh = 0, is_kw = 0;

    auto issue = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
        const int kt = is_kt;
        const bool live = kt < nk;                       // beyond the K range: zero-source dummies keep counts uniform
        if constexpr (UNIFORM) {
            if (!live) {                                   // uniform: the ring's trailing dummy stages
#pragma unroll
                for (int i = 0; i < NAI; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t)zero, (lptr_t)(sA + (lw * (NAI * 8) + i * 8) * 128), 16, 0, 0);
#pragma unroll
                for (int i = 0; i < NBI; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t)zero, (lptr_t)(sB + (lw * (NBI * 8) + i * 8) * 128), 16, 0, 0);
                ++is_kt;
                return;
            }
            if (p.linear) {
                // 1x1 / stride 1 / no padding: no taps, no bounds -- the per-lane pointers just walk K.
                // (rows past M and columns past N hold the zero word with a zero stride.)
#pragma unroll
                for (int i = 0; i < NAI; ++i) {
                    __builtin_amdgcn_global_load_lds((gptr_t)a_ptr[i], (lptr_t)(sA + (lw * (NAI * 8) + i * 8) * 128), 16, 0, 0);
                    a_ptr[i] += a_step[i];
                }
#pragma unroll
                for (int i = 0; i < NBI; ++i) {
                    __builtin_amdgcn_global_load_lds((gptr_t)b_walk[i], (lptr_t)(sB + (lw * (NBI * 8) + i * 8) * 128), 16, 0, 0);
                    b_walk[i] += b_step[i];
                }
                ++is_kt;
                return;
            }
            const long tapoff = (long)is_kh * p.sH + (long)is_kw * p.sW + is_cb;
#pragma unroll
            for (int i = 0; i < NAI; ++i) {
                const bool ok = live && ((a_mask[i] >> is_tap) & 1u);
                const bf16_t* src = ok ? a_ptr[i] + tapoff : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + (lw * (NAI * 8) + i * 8) * 128), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NBI; ++i) {
                const bf16_t* src = (live && b_ok[i]) ? b_ptr[i] + (long)kt * BK : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sB + (lw * (NBI * 8) + i * 8) * 128), 16, 0, 0);
            }
            is_cb += BK;
            if (is_cb >= p.Cin) {
                is_cb = 0; ++is_tap; ++is_kw;
                if (is_kw == p.KW) { is_kw = 0; ++is_kh; }
            }
        } else {
            const int k0 = kt * BK;
#pragma unroll
            for (int i = 0; i < NAI; ++i) {
                const int kk = k0 + a_c[i];
                const int tap = kk / p.Cin;
                const int c = kk - tap * p.Cin;
                const int kh = tap / p.KW;
                const int kw = tap - kh * p.KW;
                const int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
                const bool ok = (kk < p.K) && ((unsigned)hi < (unsigned)p.Hin) && ((unsigned)wi < (unsigned)p.Win);
                // a_ptr already carries this lane's chunk offset a_c: add the tap offset and (c - a_c)
                const bf16_t* src = ok ? a_ptr[i] + ((long)kh * p.sH + (long)kw * p.sW + (c - a_c[i])) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + (lw * (NAI * 8) + i * 8) * 128), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NBI; ++i) {
                const bool ok = b_ok[i] && (k0 + b_c[i] < p.K);
                const bf16_t* src = ok ? b_ptr[i] + k0 : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sB + (lw * (NBI * 8) + i * 8) * 128), 16, 0, 0);
            }
        }
        ++is_kt;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // fragment read offsets, K-invariant: row*128 + ((chunk ^ ((row>>1)&7)) * 16), chunk = 2*ks + h
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

    if (p.in_affine) {
        // (scale, shift) of the input's BatchNorm for all Cin (= K) channels, into LDS
        const double inv = 1.0 / (kStatScale * p.in_count);     // same arithmetic as bn_table_from_acc
        for (int c = tid; c < p.Cin; c += NT) {
            float sc, sh;
            if (p.in_acc) {
                const long long s1 = p.in_acc[c], s2 = p.in_acc[p.Cin + c];
                const double mean = (double)s1 * inv;
                double var = (double)s2 * inv - mean * mean;
                if (var < 0.0) var = 0.0;
                const float invstd = 1.0f / sqrtf((float)var + p.in_eps);
                sc = p.in_gamma[c] * invstd;
                sh = p.in_beta[c] - (float)mean * sc;
                if (bid == 0) {
                    if (p.in_running_mean) {
                        const double unbiased = p.in_count > 1.0 ? var * p.in_count / (p.in_count - 1.0) : var;
                        // the batch statistic enters as an f32 value: a deferred update (sat_bn_running_apply) is then bit-identical
                        p.in_running_mean[c] = (float)((1.0 - p.in_momentum) * p.in_running_mean[c] + p.in_momentum * (double)(float)mean);
                        p.in_running_var[c] = (float)((1.0 - p.in_momentum) * p.in_running_var[c] + p.in_momentum * (double)(float)unbiased);
                    }
                    if (p.in_acc_clear) {
                        p.in_acc_clear[c] = 0;
                        p.in_acc_clear[p.Cin + c] = 0;
                    }
                }
            } else {
                sc = p.in_scale[c];
                sh = p.in_shift[c];
            }
            in_tab[c] = sc;
            in_tab[p.Cin + c] = sh;
        }
        __syncthreads();          // nothing is in flight yet: a plain barrier (with its LDS wait) is fine here
    }

    // relu(x*scale + shift) on the A half of a landed stage, in place (rows past M stay zero)
    auto affine_stage = [&](int buf, int kt) {
        char* sA = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < BM * 8 / NT; ++j) {
            const int q = tid + j * NT;
            const int row = q >> 3, pos = q & 7;
            if (m0 + row < p.M) {
                const int c0 = kt * BK + ((pos ^ ((row >> 1) & 7)) << 3);      // channel of this chunk's first element
                bf16x8 v = *(const bf16x8*)(sA + row * 128 + pos * 16);
                const f32x4 s0 = *(const f32x4*)(in_tab + c0), s1 = *(const f32x4*)(in_tab + c0 + 4);
                const f32x4 t0 = *(const f32x4*)(in_tab + p.Cin + c0), t1 = *(const f32x4*)(in_tab + p.Cin + c0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (bf16_t)fmaxf((float)v[e] * s0[e] + t0[e], 0.0f);
                    v[e + 4] = (bf16_t)fmaxf((float)v[e + 4] * s1[e] + t1[e], 0.0f);
                }
                *(bf16x8*)(sA + row * 128 + pos * 16) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes are done ...
        __builtin_amdgcn_s_barrier();                         // ... and everybody's (raw barrier: the ring stays in flight)
        asm volatile("" ::: "memory");
    };

    auto compute = [&](int buf) {
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
    };

    // PF: fragments of (stage, substep 0) are carried in registers across the barrier
    bf16x8 caf[TM], cbf[TN];
    auto load_frag = [&](const char* st, int ks, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *(const bf16x8*)(st + a_off[i][ks]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *(const bf16x8*)(st + b_off[j][ks]);
    };
    auto compute_pf = [&](int buf, bool first) {
        const char* st = smem + buf * STAGE;
        const char* nst = smem + (buf + 1 == S ? 0 : buf + 1) * STAGE;
        if (first) load_frag(st, 0, caf, cbf);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 naf[TM], nbf[TN];
            if (ks < 3) load_frag(st, ks + 1, naf, nbf);
            else load_frag(nst, 0, naf, nbf);          // K-step kt+1 has landed (WAITN); past the end: a dummy stage
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(caf[i], cbf[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) caf[i] = naf[i];
#pragma unroll
            for (int j = 0; j < TN; ++j) cbf[j] = nbf[j];
        }
    };

    // Ring protocol (one raw s_barrier per K-step, all waves):
    //   loader : wait until its own pieces of K-step kt have landed (all but the D-1 younger K-steps done) -> barrier
    //            -> issue K-step kt+D into the slot K-step kt-1 occupied
    //   consumer: barrier -> read + MFMA K-step kt
    // Passing barrier kt tells a consumer that every loader's wait covered K-step kt, and tells a loader that every
    // consumer has finished its reads of K-step kt-1 (reads retire before the MFMAs that precede the barrier).
    if constexpr (SPEC) {
        if (is_loader) {
#pragma unroll
            for (int s = 0; s < D; ++s) issue(s);
            int buf = 0;
            for (int kt = 0; kt < nk; ++kt) {
                wait_vmcnt<WAITN>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                int nbuf = buf + D;
                if (nbuf >= S) nbuf -= S;
                issue(nbuf);
                buf = (buf + 1 == S) ? 0 : buf + 1;
            }
        } else {
            int buf = 0;
            for (int kt = 0; kt < nk; ++kt) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");        // no LDS read may be hoisted above the barrier
                if constexpr (PF) compute_pf(buf, kt == 0);
                else compute(buf);
                buf = (buf + 1 == S) ? 0 : buf + 1;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < D; ++s) issue(s);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt<WAITN>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");            // no LDS read may be hoisted above the barrier
            int nbuf = buf + D;
            if (nbuf >= S) nbuf -= S;
            issue(nbuf);
            if constexpr (PF) {
                compute_pf(buf, kt == 0);
            } else {
                if (p.in_affine) affine_stage(buf, kt);
                compute(buf);
            }
            buf = (buf + 1 == S) ? 0 : buf + 1;
        }
    }
    // drain the dummy prefetches and let every wave finish its last reads before the ring is reused
    wait_vmcnt<0>();
    __syncthreads();

    // ---- epilogue 1: BatchNorm partial column sums from the f32 accumulators ----
    float* red = (float*)(smem + BM * CROW);     // [WGM][2][BN] floats, placed after the C tile
    if ((p.stat_partial || p.acc) && is_consumer) {
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
    }
    if (p.acc) {
        // this tile's column sums join the launch-wide fixed-point accumulators
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            const int col = n0 + c;
            if (col < p.N) {
                float s = 0.0f, q = 0.0f;
#pragma unroll
                for (int g = 0; g < WGM; ++g) {
                    s += red[(g * 2 + 0) * BN + c];
                    q += red[(g * 2 + 1) * BN + c];
                }
                atomicAdd((unsigned long long*)(p.acc + col), (unsigned long long)__double2ll_rn((double)s * kStatScale));
                atomicAdd((unsigned long long*)(p.acc + p.N + col), (unsigned long long)__double2ll_rn((double)q * kStatScale));
            }
        }
    }
    // ---- epilogue 2: bf16 C tile through LDS (C/D map: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)) ----
    if (is_consumer)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * WN + j * 32 + r;
        float osc = 1.0f, osh = 0.0f;
        if (p.out_scale && n0 + col < p.N) { osc = p.out_scale[n0 + col]; osh = p.out_shift[n0 + col]; }
        const bool relu_now = p.out_relu && !p.residual;      // with a residual the ReLU follows the add (store phase)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = acc[i][j][e];
                if (p.out_scale) v = v * osc + osh;
                if (relu_now) v = fmaxf(v, 0.0f);
                *(bf16_t*)(smem + row * CROW + col * 2) = (bf16_t)v;
            }
    }
    __syncthreads();
    if (p.stat_partial) {
        for (int c = tid; c < BN; c += NT) {
            const int col = n0 + c;
            if (col < p.N) {
                float s = 0.0f, q = 0.0f;
#pragma unroll
                for (int g = 0; g < WGM; ++g) {          // fixed order over the M-waves
                    s += red[(g * 2 + 0) * BN + c];
                    q += red[(g * 2 + 1) * BN + c];
                }
                p.stat_partial[((long)tile_m * 2 + 0) * p.N + col] = s;
                p.stat_partial[((long)tile_m * 2 + 1) * p.N + col] = q;
            }
        }
    }
    constexpr int CPR = BN / 8;                  // 16-byte chunks per C row
#pragma unroll
    for (int it = 0; it < BM * CPR / NT; ++it) {
        const int qid = tid + it * NT;
        const int row = qid / CPR, cc = qid - row * CPR;
        const int grow = m0 + row, gcol = n0 + cc * 8;
        if (grow < p.M && gcol < p.N) {           // N % 8 == 0: a chunk is all in or all out
            if (p.residual) {                     // out = [relu](affine(acc) + residual), 16 bytes of each per thread
                const bf16x8 c = *(const bf16x8*)(smem + row * CROW + cc * 16);
                const bf16x8 z = *(const bf16x8*)(p.residual + (long)grow * p.ldc + gcol);
                bf16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float v = (float)c[k] + (float)z[k];
                    if (p.out_relu) v = fmaxf(v, 0.0f);
                    o[k] = (bf16_t)v;
                }
                *(bf16x8*)(p.C + (long)grow * p.ldc + gcol) = o;
            } else {
                store16_wt(p.C + (long)grow * p.ldc + gcol, *(const u32x4*)(smem + row * CROW + cc * 16));
            }
        }
    }
}

template <int BN, int S, int NW, bool SPEC = false, bool PF = false, int BM = 128>
int launch_glds(ConvArgs& a, int groups, hipStream_t s) {
    const int tm = sat_cdiv(a.M, BM), tn = sat_cdiv(a.N, BN);
    a.tiles_n = tn;
    const bool uniform = (a.Cin % 64 == 0) && (a.KH * a.KW <= 32);
    const dim3 grid(tm * tn, groups), block(NW * 64);
    hipEvent_t e0 = t_ev_start, e1 = t_ev_stop;     // armed: timed diagnostic launch (same kernel, same grid, + the packet's timestamps)
    t_ev_start = t_ev_stop = nullptr;
    if (uniform) {
        if (e0) hipExtLaunchKernelGGL((conv_glds_kernel<BN, S, NW, true, SPEC, PF, BM>), grid, block, 0, s, e0, e1, 0, a);
        else hipLaunchKernelGGL((conv_glds_kernel<BN, S, NW, true, SPEC, PF, BM>), grid, block, 0, s, a);
    } else {
        if (e0) hipExtLaunchKernelGGL((conv_glds_kernel<BN, S, NW, false, SPEC, PF, BM>), grid, block, 0, s, e0, e1, 0, a);
        else hipLaunchKernelGGL((conv_glds_kernel<BN, S, NW, false, SPEC, PF, BM>), grid, block, 0, s, a);
    }
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// kernel variants: (tile width, ring stages, waves, wave specialisation, fragment prefetch, tile rows).
// LDS = S * (BM/8 + BN/8) KB (+ the table) decides workgroups per CU.
struct Variant { int bn, s, nw, spec, pf, bm, xp, pr, stem, pw, aw, ap, ay, rs; };  // xp: conv_xp_kernel with xp column tiles per workgroup; pr: conv_pr_kernel; stem: conv_stem_kernel; pw: conv_pw_kernel; aw: conv_aw_kernel
constexpr Variant kVariants[] = {
    {128, 4, 8, 0, 0, 128}, {128, 3, 8, 0, 0, 128}, {128, 2, 8, 0, 0, 128}, {64, 4, 8, 0, 0, 128}, {64, 3, 8, 0, 0, 128},
    {64, 2, 8, 0, 0, 128}, {128, 4, 4, 0, 0, 128}, {128, 2, 4, 0, 0, 128}, {64, 3, 4, 0, 0, 128}, {64, 2, 4, 0, 0, 128},
    {128, 4, 8, 1, 0, 128}, {128, 3, 8, 1, 0, 128}, {128, 2, 8, 1, 0, 128}, {64, 4, 8, 1, 0, 128}, {64, 3, 8, 1, 0, 128},   // 4 consumer + 4 loader waves
    {128, 4, 8, 1, 1, 128}, {128, 4, 8, 0, 1, 128}, {128, 4, 4, 0, 1, 128}, {64, 4, 8, 1, 1, 128}, {64, 4, 8, 0, 1, 128},
    {64, 4, 4, 0, 1, 128},                                                                                               // fragment prefetch
    {256, 2, 8, 0, 0, 128}, {256, 3, 8, 0, 0, 128}, {256, 3, 8, 0, 0, 64},                                               // wide tiles
    {128, 3, 8, 0, 0, 64}, {128, 2, 8, 0, 0, 64},                                                                        // 64-row tiles: 2 workgroups per CU on the N = 256 layers
    {128, 5, 4, 0, 0, 128, 1}, {128, 5, 4, 0, 0, 128, 2}, {128, 5, 4, 0, 0, 128, 4},                                     // register-resident A panel (expansion 1x1 convs, sat_conv_xp.inc)
    {128, 6, 8, 1, 0, 128, 0, 1},                                                                                        // LDS-resident input patch (3x3 / stride 1, sat_conv_pr.inc)
    {64, 1, 4, 0, 0, 128, 0, 0, 1},                                                                                      // persistent stem kernel: weights in registers, input row segments in LDS (sat_conv_stem.inc)
    {128, 1, 4, 0, 0, 128, 0, 0, 0, 1},                                                                                  // LDS-resident input patch + weights straight into registers from the fragment-ordered copy (3x3 / stride 1, sat_conv_pw.inc)
    {128, 3, 4, 0, 0, 128, 0, 0, 0, 0, 1},                                                                               // 1x1: activations through registers into LDS, weights straight into registers (sat_conv_aw.inc)
    {256, 3, 8, 0, 0, 128, 0, 0, 0, 0, 2},                                                                               // ... eight waves, 256-column tiles: the activations staged once per row tile
    {128, 2, 4, 0, 0, 128, 0, 0, 0, 0, 0, 1},                                                                            // expansion 1x1 (K = 256): weights resident in registers, the workgroup persistent over row tiles (sat_conv_ap.inc)
    {128, 2, 4, 0, 0, 128, 0, 0, 0, 0, 0, 0, 1},                                                                         // conv1 that also finishes the previous bottleneck: operand = relu(bn3(c3) + y), written out as it goes (sat_conv_ay.inc)
    {256, 2, 8, 0, 0, 128, 0, 0, 0, 0, 0, 0, 2},                                                                         // ... eight waves, 256-column tiles
    {64, 1, 4, 0, 0, 128, 0, 0, 0, 0, 0, 0, 0, 1},                                                                       // 3x3 / stride 1 over 32 channels on large maps (Inception stem): weights in registers, whole input rows in LDS (sat_conv_rs.inc)
    {64, 1, 4, 0, 0, 128, 0, 0, 0, 0, 0, 0, 0, 2},                                                                       // ... 64 -> 64 channels on 56 x 56 maps (ResNet layer 1): two output rows per step, statistics per workgroup
    {64, 1, 4, 0, 0, 128, 0, 0, 0, 0, 0, 0, 0, 3},                                                                       // ... 3 x 3 / stride 2 over the image's padded 8 channels -> 32 (the first conv of the Inception stem)
    {64, 1, 4, 0, 0, 128, 0, 0, 0, 0, 0, 0, 0, 4},                                                                       // ... ResNet's 7 x 7 stem in the stem kernel's layout
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
constexpr int kVariantPr = 29;

#include "sat_conv_xp.inc"
#include "sat_conv_pr.inc"
#include "sat_conv_stem.inc"
#include "sat_conv_pw.inc"
#include "sat_conv_aw.inc"
#include "sat_conv_ap.inc"
#include "sat_conv_ay.inc"
#include "sat_conv_rs.inc"

int launch_variant(int v, ConvArgs& a, int groups, hipStream_t s) {
    switch (v) {
        case 0: return launch_glds<128, 4, 8>(a, groups, s);
        case 1: return launch_glds<128, 3, 8>(a, groups, s);
        case 2: return launch_glds<128, 2, 8>(a, groups, s);
        case 3: return launch_glds<64, 4, 8>(a, groups, s);
        case 4: return launch_glds<64, 3, 8>(a, groups, s);
        case 5: return launch_glds<64, 2, 8>(a, groups, s);
        case 6: return launch_glds<128, 4, 4>(a, groups, s);
        case 7: return launch_glds<128, 2, 4>(a, groups, s);
        case 8: return launch_glds<64, 3, 4>(a, groups, s);
        case 9: return launch_glds<64, 2, 4>(a, groups, s);
        case 10: return launch_glds<128, 4, 8, true>(a, groups, s);
        case 11: return launch_glds<128, 3, 8, true>(a, groups, s);
        case 12: return launch_glds<128, 2, 8, true>(a, groups, s);
        case 13: return launch_glds<64, 4, 8, true>(a, groups, s);
        case 14: return launch_glds<64, 3, 8, true>(a, groups, s);
        case 15: return launch_glds<128, 4, 8, true, true>(a, groups, s);
        case 16: return launch_glds<128, 4, 8, false, true>(a, groups, s);
        case 17: return launch_glds<128, 4, 4, false, true>(a, groups, s);
        case 18: return launch_glds<64, 4, 8, true, true>(a, groups, s);
        case 19: return launch_glds<64, 4, 8, false, true>(a, groups, s);
        case 20: return launch_glds<64, 4, 4, false, true>(a, groups, s);
        case 21: return launch_glds<256, 2, 8, false, false, 128>(a, groups, s);
        case 22: return launch_glds<256, 3, 8, false, false, 128>(a, groups, s);
        case 23: return launch_glds<256, 3, 8, false, false, 64>(a, groups, s);
        case 24: return launch_glds<128, 3, 8, false, false, 64>(a, groups, s);
        case 25: return launch_glds<128, 2, 8, false, false, 64>(a, groups, s);
        case 26: return launch_xp(a, 1, groups, s);
        case 27: return launch_xp(a, 2, groups, s);
        case 28: return launch_xp(a, 4, groups, s);
        case 29: return launch_pr(a, groups, s);
        case 30: return launch_stem(a, groups, s);
        case 31: return launch_pw(a, groups, s);
        case 32: return launch_aw<4>(a, groups, s);
        case 33: return launch_aw<8>(a, groups, s);
        case 34: return launch_ap(a, groups, s);
        case 35: return launch_ay<4>(a, groups, s);
        case 36: return launch_ay<8>(a, groups, s);
        case 37: return launch_rs(a, groups, s);
        case 38: return launch_rs64(a, groups, s);
        case 39: return launch_rs8(a, groups, s);
        case 40: return launch_rs_stem(a, groups, s);
        default: return SAT_ERR_ARG;
    }
}

// What fixes the BITS of a variant's BatchNorm column sums (the conv output itself is bit-identical across ring variants and
// conv_xp_kernel: same MFMA, same K order): the kernel family and, for the ring kernel, the tile rows and the consumer wave
// grid (rows summed per lane, then the M-waves in order).  Two variants with the same signature leave the same statistics, so a
// grouped launch and an ungrouped one on variants of one signature give a batch the same bits.
// What fixes the bits of the conv OUTPUT: the order in which the K axis is walked.  0: tap major (ring kernel, conv_xp_kernel,
// conv_aw_kernel, stem kernel: bit-identical outputs), 1: channel-block major (conv_pr_kernel, conv_pw_kernel: bit-identical to each other)
int output_family(int v) { return (kVariants[v].pr || kVariants[v].pw) ? 1 : 0; }
constexpr int kFamilySig = 100000;          // signature constraints >= this one name an output family only (inference: no statistics)
bool signature_matches(int v, int want);

int stat_signature(int v) {
    const Variant& k = kVariants[v];
    if (k.rs) return k.rs == 2 ? 7001 : (k.rs == 4 ? 7002 : 7000);       // (a wave's pixels of a row, then the four waves in order / of a workgroup's whole run of steps)
    if (k.ap) return 6000;          // (a lane's 64 rows of a tile, the tiles of a worker in order, then the two halves)
    if (k.aw || k.ay) return 5000;          // (a lane's 64 rows, then the two halves: the same for the four- and the eight-wave form)
    if (k.pw) return 4000;
    if (k.stem) return 3000;
    if (k.pr) return 2000;
    if (k.xp) return 1000;
    const int cw = k.spec ? k.nw / 2 : k.nw;
    const int wgm = (cw == 4) ? 2 : ((k.bm == 128 && k.bn == 64) ? 4 : 2);
    return k.bm * 8 + wgm;
}

bool signature_matches(int v, int want) {
    return want >= kFamilySig ? output_family(v) == want - kFamilySig : stat_signature(v) == want;
}

ConvArgs make_args(const sat_op* op) {
    ConvArgs a = {};
    a.A = (const bf16_t*)op->in0; a.B = (const bf16_t*)op->w; a.C = (bf16_t*)op->out;
    a.Bp = (const bf16_t*)op->w_packed;
    a.in_bytes = (long)op->N * op->sN * 2;
    a.stat_partial = op->stat_partial;
    a.acc = (long long*)op->stat_acc;
    a.in_affine = 0;
    if (op->scale0 || op->stat_acc1) {          // BatchNorm + ReLU of the INPUT fused into the A staging
        a.in_affine = 1;
        a.in_scale = op->scale0; a.in_shift = op->shift0;
        a.in_acc = (const long long*)op->stat_acc1;
        a.in_gamma = op->gamma1; a.in_beta = op->beta1;
        a.in_running_mean = op->running_mean1; a.in_running_var = op->running_var1;
        a.in_count = (double)op->count; a.in_momentum = op->momentum; a.in_eps = op->eps;
    }
    a.out_scale = op->scale1; a.out_shift = op->shift1;        // inference epilogue: affine (+ residual) (+ ReLU)
    a.residual = (const bf16_t*)op->in1;
    if (op->flags & SAT_CONV_IN_RESIDUAL) {     // in1 belongs to the INPUT side: operand = relu(bn(in0) + in1), also written to out1
        a.residual = nullptr;
        a.in_res = (const bf16_t*)op->in1;
        a.Y = (bf16_t*)op->out1;
    }
    a.out_relu = op->flags & 1;
    a.M = op->N * op->Hout * op->Wout; a.N = op->Cout; a.K = op->KH * op->KW * op->Cin;
    a.ldb = a.K; a.ldc = (op->ldc >= op->Cout) ? op->ldc : op->Cout;
    a.Hin = op->Hin; a.Win = op->Win; a.Cin = op->Cin; a.Hout = op->Hout; a.Wout = op->Wout;
    a.KH = op->KH; a.KW = op->KW; a.stride = op->stride; a.pad = op->pad;
    a.padw = (op->flags & SAT_CONV_PADW) ? op->pad_w : op->pad;
    a.sN = op->sN; a.sH = op->sH; a.sW = op->sW;
    a.linear = (op->KH == 1 && op->KW == 1 && op->stride == 1 && op->pad == 0 && a.padw == 0 && op->Hout == op->Hin &&
                op->Wout == op->Win && op->sH == (long)op->Win * op->sW && op->sN == (long)op->Hin * op->sH) ? 1 : 0;
    // grouped launch: every per-batch buffer is `groups` consecutive copies of the ungrouped one (include/sat_hip.h, sat_op.groups)
    a.gs_a = (long)op->N * op->sN;
    a.gs_c = (long)a.M * a.ldc;
    a.gs_partial = (long)op->tiles_m * 2 * a.N;
    a.gs_acc = 4L * a.N;                        // [2 parities][2][N]
    a.gs_in_acc = 4L * a.Cin;
    a.gs_in_run = 2L * a.Cin;                   // [mean row][var row]
    a.gs_out_tab = (op->flags & SAT_CONV_GROUP_TABLE) ? 2L * a.N : 0L;      // [scale row][shift row] per group
    return a;
}

int op_groups(const sat_op* op) { return op->groups > 1 ? op->groups : 1; }

bool variant_ok(int v, const ConvArgs& a) {
    if (v < 0 || v >= kNumVariants) return false;
    const Variant& k = kVariants[v];
    if (k.ay) return ay_ok(a, k.ay == 2 ? 8 : 4);
    if (a.in_res) return false;
    if (k.rs) return k.rs == 2 ? rs64_ok(a) : (k.rs == 3 ? rs8_ok(a) : (k.rs == 4 ? rs_stem_ok(a) : rs_ok(a)));                                      // only conv_ay_kernel builds its operand from two tensors
    if (k.ap) return ap_ok(a);
    if (k.aw) return aw_ok(a, k.aw == 2 ? 8 : 4);
    if (k.pw) return pw_ok(a);
    if (k.stem) return stem_ok(a);
    if (k.pr) return pr_ok(a);
    if (k.xp) return xp_ok(a, k.xp);
    if (k.bn >= 128 && a.N <= 64) return false;
    if (k.bn == 256 && a.N <= 128) return false;
    if ((k.spec || k.pf) && a.in_affine) return false;               // the in-LDS input transform lives in the plain unified-wave loop
    if (a.in_affine && !a.linear) return false;                      // ... and is 1x1-only (no tap mask): 3x3 convs fuse it in conv_pr_kernel
    if (k.bm != 128 && a.stat_partial) return false;                 // the per-tile statistics slabs are 128-row tiles
    return true;
}

int heuristic_variant(const ConvArgs& a) {
    // 128x128 with a deep ring when it still leaves >= 2 tiles per CU and K is long enough to use the ring;
    // otherwise 128x64 with a shallower ring (more workgroups per CU to overlap prologue/epilogue phases)
    if (a.in_res) return ay_ok(a, 4) ? 35 : 36;           // the operand built from the raw conv3 tensor and the residual: conv_ay_kernel
    if (a.in_affine && !a.linear) return kVariantPr;      // 3x3 with a fused input BatchNorm: the LDS-resident patch (the builder fuses bn1 only where it can run)
    if (stem_ok(a)) return 30;                            // the op program's stem layout: the persistent stem kernel
    if (rs_ok(a)) return 37;                              // 3x3 over 32 channels on a large map (Inception stem): whole input rows in LDS
    if (rs_stem_ok(a)) return 40;                         // ... the stem layout with an inference epilogue (conv_stem_kernel has none)
    if (rs8_ok(a)) return 39;                             // ... the stem's first conv (stride 2 over the padded image)
    if (rs64_ok(a)) return 38;                            // 3x3 64 -> 64 on ~56-pixel rows (ResNet layer 1): the same, two output rows per step
    const long t128 = (long)sat_cdiv(a.M, 128) * sat_cdiv(a.N, 128);
    const int nk = sat_cdiv(a.K, 64);
    if (a.N > 64 && t128 >= 512) return nk <= 4 ? 2 : 0;
    return nk <= 4 ? 5 : 4;
}

// argument checks + parity / group bookkeeping shared by the launch and the tuner
int prepare_args(const sat_op* op, int parity, ConvArgs& a) {
    if (op->Cout % 8) return SAT_ERR_UNSUPPORTED;
    a = make_args(op);
    if (a.acc) a.acc += (long)parity * 2 * a.N;                          // [2 parities][2][N]
    if ((a.out_scale != nullptr) != (a.out_shift != nullptr)) return SAT_ERR_ARG;
    if (a.residual && (!a.out_scale || (const void*)a.residual == (const void*)a.C)) return SAT_ERR_ARG;
    if (a.out_scale && (a.stat_partial || a.acc)) return SAT_ERR_ARG;     // batch statistics and a fixed affine exclude each other
    if (op->flags & SAT_CONV_IN_RESIDUAL) {
        if (!a.in_res || !a.Y || !a.in_affine || (((uintptr_t)a.in_res | (uintptr_t)a.Y) & 15)) return SAT_ERR_ARG;
        if (!ay_ok(a, 4) && !ay_ok(a, 8)) return SAT_ERR_UNSUPPORTED;
    }
    if (a.in_affine) {
        if ((a.Cin > 512 && !a.in_res) || (a.Cin % 64) || (a.KH * a.KW > 32)) return SAT_ERR_UNSUPPORTED;
        if (a.in_acc) {
            if (!a.in_gamma || !a.in_beta || a.in_count < 1) return SAT_ERR_ARG;
            long long* base = (long long*)op->stat_acc1;       // [2 parities][2][Cin]
            a.in_acc = base + (long)parity * 2 * a.Cin;
            a.in_acc_clear = base + (long)(1 - parity) * 2 * a.Cin;
        } else if (!a.in_scale || !a.in_shift) {
            return SAT_ERR_ARG;
        }
        if (!a.linear && !pr_ok(a)) return SAT_ERR_UNSUPPORTED;
    }
    if (op_groups(op) > 1) {
        // grouped: batch statistics only (a fixed affine needs no groups: eval-mode batches simply concatenate), running
        // statistics in the per-group [mean row][var row] log layout
        // (train-mode conv3 behind SAT_OP_BN_FROM_GRAM: a per-group (scale, shift) table and a per-group residual, SAT_CONV_GROUP_TABLE)
        if (((a.out_scale || a.residual) && !(op->flags & SAT_CONV_GROUP_TABLE)) || (a.in_affine && !a.in_acc)) return SAT_ERR_UNSUPPORTED;
        if (a.in_running_mean && a.in_running_var != a.in_running_mean + a.Cin) return SAT_ERR_ARG;
        if (a.stat_partial && op->tiles_m < sat_cdiv(a.M, 128)) return SAT_ERR_ARG;
        if (op_groups(op) > 65535) return SAT_ERR_ARG;
    }
    return SAT_OK;
}

}  // namespace

// bf16 SAT_OP_CONV; arguments already validated by sat_conv_launch
int sat_conv_glds_launch(const sat_op* op, int parity, hipStream_t s) {
    ConvArgs a;
    SAT_TRY(prepare_args(op, parity, a));
    int v = (op->variant > 0 && op->variant <= kNumVariants) ? op->variant - 1 : heuristic_variant(a);
    if (!variant_ok(v, a)) v = heuristic_variant(a);      // e.g. the in-LDS transforms live in the plain unified-wave loop
    return launch_variant(v, a, op_groups(op), s);
}

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>
#include <tuple>
#include <stdio.h>

constexpr int kTuneTab = 2048;      // channels of the tuner's neutral (scale 1, shift 0) input-BatchNorm table

typedef std::tuple<int, int, int, int, int, int, int, int, int, int, int, int, int> TuneKey;
static std::map<TuneKey, std::vector<int>> g_tune_cache;      // variants (0-based) by ascending replay time
static std::mutex g_tune_mu;                       // the per-geometry result cache is shared by every caller thread

static TuneKey tune_key(const sat_op* op, int groups, int want_sig) {
    return TuneKey(op->N, op->Hin, op->Win, op->Cin, op->Hout, op->Wout, op->Cout, op->KH, op->KW, op->stride,
                   ((op->stat_partial || op->stat_acc) ? 1 : 0) + ((op->scale0 || op->stat_acc1) ? 2 : 0) +
                       (op->scale1 ? 4 : 0) + (op->in1 ? 8 : 0) + (op->w_packed ? 16 : 0), groups, want_sig);
}

// time every variant the kernel can run for `op` as a launch of `groups` groups (only variants of statistics signature
// `want_sig` when >= 0); returns the fastest in *best_v.  Tuning launches write the op's own output buffer and touch no
// statistics / running buffers.
static int tune_one(const sat_op* op, int groups, int want_sig, int reps, float* scratch, hipEvent_t e0, hipEvent_t e1, hipStream_t s,
                    bool verbose, std::vector<int>* ranked) {
    ConvArgs a = make_args(op);
    a.acc = nullptr;
    if (a.in_affine) {           // ... nor derive from / clear the live accumulators: the neutral table stands in
        a.in_acc = nullptr; a.in_acc_clear = nullptr; a.in_running_mean = nullptr; a.in_running_var = nullptr;
        a.in_scale = scratch; a.in_shift = scratch + kTuneTab;
        if ((a.Cin > 512 && !a.in_res) || a.Cin > kTuneTab || (a.Cin % 64) || (a.KH * a.KW > 32)) return SAT_ERR_UNSUPPORTED;
    }
    std::vector<std::pair<float, int>> timed;
    int rc = SAT_OK;
    for (int v = 0; v < kNumVariants && rc == SAT_OK; ++v) {
        if (!variant_ok(v, a)) continue;
        if (want_sig >= 0 && !signature_matches(v, want_sig)) continue;
        float tmin = 1e30f;
        for (int round = 0; round < 4 && rc == SAT_OK; ++round) {       // round 0 = warm-up, then best of 3
            if (hipEventRecord(e0, s) != hipSuccess) { rc = SAT_ERR_UNSUPPORTED; break; }
            for (int r = 0; r < reps && rc == SAT_OK; ++r) rc = launch_variant(v, a, groups, s);
            if (rc != SAT_OK) break;
            if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { rc = SAT_ERR_UNSUPPORTED; break; }
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = SAT_ERR_UNSUPPORTED; break; }
            if (round >= 1 && ms / reps < tmin) tmin = ms / reps;      // per launch
        }
        if (verbose) fprintf(stderr, "  tune M=%d N=%d K=%d G=%d v%d(%d,%d,%d,%s) %.2f us\n", a.M, a.N, a.K, groups, v, kVariants[v].bn,
                             kVariants[v].s, kVariants[v].nw, kVariants[v].spec ? (kVariants[v].pf ? "spec+pf" : "spec") : (kVariants[v].pf ? "pf" : "-"), tmin * 1e3f);
        timed.emplace_back(tmin, v);
    }
    if (rc != SAT_OK) return rc;
    std::sort(timed.begin(), timed.end());
    ranked->clear();
    for (auto& tv : timed) ranked->push_back(tv.second);
    if (verbose && !timed.empty())
        fprintf(stderr, "tune M=%d N=%d K=%d G=%d -> v%d %.2f us (%.0f TFLOP/s)\n", a.M, a.N, a.K, groups, timed[0].second, timed[0].first * 1e3f,
                2.0 * groups * a.M * a.N * a.K / (timed[0].first * 1e-3) / 1e12);
    return rc;
}

// the candidates of one op by ascending replay time (cached per geometry, group count and signature constraint); empty: the op
// is not a bf16 conv / has nothing to choose from (the launch falls back on the heuristic)
static int ranked_variants(sat_op* op, int reps, float* scratch, hipEvent_t e0, hipEvent_t e1, hipStream_t s, bool verbose,
                           std::vector<int>* ranked) {
    ranked->clear();
    if (op->kind != SAT_OP_CONV || op->dtype != SAT_BF16 || (op->Cout % 8)) return SAT_OK;
    const int groups = op_groups(op);
    // variant < 0 on entry: only variants of signature -variant (the caller keeps every program of one model on the signatures of
    // its first one, so that a batch gets the same bits whichever program runs it)
    const int want = op->variant < 0 ? -op->variant : -1;
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        auto it = g_tune_cache.find(tune_key(op, groups, want));
        if (it != g_tune_cache.end()) { *ranked = it->second; return SAT_OK; }
    }
    int rc = tune_one(op, groups, want, reps, scratch, e0, e1, s, verbose, ranked);
    if (rc == SAT_ERR_UNSUPPORTED) { ranked->clear(); return want >= 0 ? rc : SAT_OK; }
    if (rc != SAT_OK) return rc;
    if (want >= 0 && ranked->empty()) return SAT_ERR_UNSUPPORTED;       // no variant of that signature runs this op
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tune_cache[tune_key(op, groups, want)] = *ranked;
    return SAT_OK;
}

static int autotune_impl(sat_op* ops, int n_ops, int reps, float* scratch, int64_t scratch_bytes, sat_stream_t stream, int topk,
                         int32_t* cand) {
    if (!ops || n_ops < 0 || reps < 1 || topk < 1) return SAT_ERR_ARG;
    if (!scratch || scratch_bytes < (int64_t)(2 * kTuneTab * sizeof(float))) return SAT_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const bool verbose = getenv("SAT_TUNE_VERBOSE") != nullptr;
    // neutral input-BatchNorm table (scale 1, shift 0) in the CALLER's scratch: the library allocates nothing
    if (hipMemsetD32Async((hipDeviceptr_t)scratch, 0x3f800000, kTuneTab, s) != hipSuccess ||
        hipMemsetAsync(scratch + kTuneTab, 0, kTuneTab * sizeof(float), s) != hipSuccess)
        return SAT_ERR_UNSUPPORTED;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return SAT_ERR_UNSUPPORTED;
    int rc = SAT_OK;
    std::vector<int> ranked;
    for (int i = 0; i < n_ops && rc == SAT_OK; ++i) {
        rc = ranked_variants(ops + i, reps, scratch, e0, e1, s, verbose, &ranked);
        if (rc != SAT_OK) break;
        if (cand)
            for (int k = 0; k < topk; ++k) cand[(long)i * topk + k] = k < (int)ranked.size() ? ranked[k] + 1 : 0;
        if (ops[i].kind == SAT_OP_CONV && ops[i].dtype == SAT_BF16) ops[i].variant = ranked.empty() ? 0 : ranked[0] + 1;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

extern "C" int sat_conv_autotune(sat_op* ops, int n_ops, int reps, float* scratch, int64_t scratch_bytes,
                                 sat_stream_t stream) {
    return autotune_impl(ops, n_ops, reps, scratch, scratch_bytes, stream, 1, nullptr);
}

// the same, and the `topk` fastest variants of every op (1-based, fastest first, 0 = no more) in cand[n_ops][topk] [host]: a
// caller that can time its whole program (sat_run_ops_timed) picks among them IN the program -- a replayed launch finds its
// operand warm, a launch in the program does not, and the two rankings differ by a few microseconds either way
extern "C" int sat_conv_autotune_topk(sat_op* ops, int n_ops, int reps, float* scratch, int64_t scratch_bytes,
                                      sat_stream_t stream, int topk, int32_t* cand) {
    if (!cand) return SAT_ERR_ARG;
    return autotune_impl(ops, n_ops, reps, scratch, scratch_bytes, stream, topk, cand);
}

// the statistics signature of a variant number as stored in sat_op.variant (1-based; 0 / out of range: -1): callers that load a
// saved tuning table check that a constrained op's variant still has the signature its leader program chose
extern "C" int sat_conv_variant_signature(int variant) {
    return (variant >= 1 && variant <= kNumVariants) ? stat_signature(variant - 1) : -1;
}
// ... and what fixes the bits of its OUTPUT alone (inference programs, where there are no statistics): 100000 + the K-order family
extern "C" int sat_conv_variant_family(int variant) {
    return (variant >= 1 && variant <= kNumVariants) ? kFamilySig + output_family(variant - 1) : -1;
}

// number of kernel variants of this build (a saved tuning table names variants by number: it is only valid for the build it was made on)
extern "C" int sat_conv_num_variants(void) { return kNumVariants; }

// The variant (1-based) a bf16 SAT_OP_CONV runs when nothing was tuned for its geometry: the built-in heuristic, or -- when a
// statistics signature / output family is asked for (want_sig >= 0) and the heuristic's choice has another one -- the first variant
// of that signature the op can run.  Depends on the op's geometry only: every process, rank and box gets the same answer.  0: none.
extern "C" int sat_conv_default_variant(const sat_op* op, int want_sig) {
    if (!op || op->kind != SAT_OP_CONV || op->dtype != SAT_BF16) return 0;
    ConvArgs a;
    if (prepare_args(op, 0, a) != SAT_OK) return 0;
    const int hv = heuristic_variant(a);
    if (variant_ok(hv, a) && (want_sig < 0 || signature_matches(hv, want_sig))) return hv + 1;
    for (int v = 0; v < kNumVariants; ++v)
        if (variant_ok(v, a) && (want_sig < 0 || signature_matches(v, want_sig))) return v + 1;
    return 0;
}

// weights [Cout][KH*KW][Cin] bf16 (the kernels' layout) -> the MFMA fragment order conv_pw_kernel streams into registers; `packed`
// holds Cout * KH*KW * Cin elements.  Frozen stacks pack once per weight version (ConvStackProgram).
extern "C" int sat_conv_pack_weights(const void* w, void* packed, int Cout, int Cin, int taps, sat_stream_t stream) {
    if (!w || !packed || Cout < 32 || (Cout % 32) || Cin < 64 || (Cin % 64) || taps < 1) return SAT_ERR_ARG;
    const long n16 = (long)Cout * taps * Cin / 8;
    const int grid = (int)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)packed, Cout, Cin, taps);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}
