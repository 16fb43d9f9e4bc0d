// Expansion conv of a ResNet bottleneck with its train-mode BatchNorm, the residual add and the ReLU in ONE launch
// (`self.resnet(images)`, models.py:27; torchvision Bottleneck.forward: out = relu(bn3(conv3(a2)) + identity)).
//
//   y[m, co] = relu( bn3( sum_k relu(bn2(raw2[m, k])) * w[co, k] ) + x[m, co] )          bf16 tensors, f32 accumulate
//
// Train-mode BatchNorm needs the statistics of the WHOLE conv output before any element can be normalised -- a grid-wide
// dependency that the three-launch form (conv3 -> raw tensor -> normalise+add) pays with one write and one read of the
// block's largest tensor and a launch that streams 77 MB (layer 3, batch 64).  With three stacks in flight the step is bound by
// exactly that memory traffic (profiles/r03_twopass_ab.txt: the normalise+add launches are 0.95 ms of a 4.78 ms step).
// Here every workgroup keeps its f32 accumulators IN REGISTERS across a grid-wide barrier:
//   1. K phase: the workgroup owns 128 rows x 512 output columns (4 blocks of 128).  Its A panel (128 x K, K = 256) is loaded
//      once into LDS and bn2 + ReLU is applied there once; the weights stream through a 5-stage LDS-DMA ring (one
//      128-column x 64-k stage per barrier, counted vmcnt); 128 accumulator registers per thread.
//   2. statistics: column sums / sums of squares of the f32 accumulators -> 2^22 fixed-point 64-bit integer atomics (bitwise
//      reproducible whatever the arrival order), as the stand-alone conv does.
//   3. grid barrier: every wave drains its atomics, one lane per workgroup adds to an arrival counter and polls it (relaxed
//      agent-scope loads, s_sleep, BOUNDED: on a timeout the kernel raises a sticky error word and leaves).
//   4. epilogue: (scale, shift) from the integer sums (the arithmetic of bn_table_from_acc), then per 128-column block the
//      normalise + add + ReLU kernel's arithmetic on the bf16-rounded accumulators with the residual read straight from
//      memory: y is written ONCE; the raw conv tensor never exists.
// Residency: the barrier needs every workgroup resident at once -- grid <= CUs, one workgroup per CU (146 KB of LDS) -- and no
// OTHER spinning kernel may hold CUs it needs.  All such kernels of the process (this one in up to three look-ahead streams, the
// persistent LSTM recurrence) therefore run under one device-wide TOKEN: a one-wave acquire kernel ahead of the launch spins
// (holding no LDS, one wave slot) until the token is free; the last workgroup to leave releases it.  Ordinary kernels that
// occupy CUs when the launch starts finish on their own.  Every wait is bounded.
#include "sat_internal.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

namespace {

__device__ u32x4 g_zero16_f;
__device__ unsigned g_resident_token;      // 0 = free; the ONE co-residency token of this process on this device

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr unsigned kSpinDefault = 1u << 21;
constexpr int NB = 4;                      // 128-column blocks per workgroup
constexpr int KS = 4;                      // K-steps of 64 (K = 256)

struct FusedArgs {
    const bf16_t* A;          // raw2 [M][K]
    const bf16_t* W;          // [N][K]
    const bf16_t* R;          // residual x [M][N]
    bf16_t* Y;                // [M][N]
    int M, N;
    int tiles_g;              // column groups of 512
    // bn2 (input side), from the integer sums of conv2
    const long long* in_acc; long long* in_acc_clear; int in_shards;
    const float* in_gamma; const float* in_beta; float* in_rm; float* in_rv;
    // bn3 (output side)
    long long* out_acc; long long* out_acc_clear;
    const float* out_gamma; const float* out_beta; float* out_rm; float* out_rv;
    double count; float momentum, eps;
    unsigned* sync;           // [0] arrivals, [1] departures (both 0 between launches)
    unsigned* err;            // sticky error word (shared by the launches of one program)
    unsigned spin_limit;
    int hold_token;           // the launch runs under the residency token: the last workgroup to leave releases it
    unsigned long long* dbg;  // diagnostics (sat_conv3_fused_debug): [workgroup][8] s_memtime stamps of lane 0, or NULL
};
constexpr double kStatScale = SAT_STAT_SCALE;

__global__ __launch_bounds__(64) void token_acquire_kernel(unsigned* err, unsigned spin_limit) {
    if (threadIdx.x != 0) return;
    unsigned spins = 0;
    while (atomicCAS(&g_resident_token, 0u, 1u) != 0u) {
        if (++spins > spin_limit) {                 // a holder that never left: report, then proceed (every later wait is bounded too)
            if (err) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

__global__ __launch_bounds__(64) void token_release_kernel() {
    if (threadIdx.x == 0) __hip_atomic_store(&g_resident_token, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(512) void conv3_fused_kernel(const FusedArgs p) {
    constexpr int K = 64 * KS, NT = 512;
    constexpr int SLAB = 128 * 128;                    // one K-step of the A panel / one ring stage: 128 rows x 128 B
    constexpr int A_BYTES = KS * SLAB, SB = 5, RING = SB * SLAB;   // 4 weight stages (64 KB) in flight per CU
    constexpr int CROW = 128 * 2;                      // bf16 C-tile row stride (epilogue staging, first half of the A region)
    constexpr int RBLK = 128 * 256;                    // one 128 x 128 bf16 block of the residual
    static_assert(128 * CROW + RBLK <= A_BYTES && 2 * NB * 2 * 128 * 4 + 2 * 512 * 4 + RBLK <= RING, "epilogue buffers fit");
    __shared__ __attribute__((aligned(16))) char smem[A_BYTES + RING + 2 * K * 4];
    char* sA = smem;
    char* sB = smem + A_BYTES;
    float* in_tab = (float*)(smem + A_BYTES + RING);   // bn2 (scale, shift)
    float* red = (float*)sB;                           // after the K phase: [2 wm][NB][2][128] column sums ...
    float* otab = (float*)(sB + 2 * NB * 2 * 128 * 4); // ... and bn3 (scale, shift) of this workgroup's 512 columns
    // epilogue: the residual blocks are prefetched into LDS by LDS-DMA (two buffers) while the grid barrier is waited for
    char* rbuf0 = sA + 128 * CROW;
    char* rbuf1 = sB + 2 * NB * 2 * 128 * 4 + 2 * 512 * 4;
    __shared__ int s_flag;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;           // 2 (M) x 4 (N) waves: 64 x 32 of every 128 x 128 block
    const int r = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_m = swz / p.tiles_g, g = swz - tile_m * p.tiles_g;
    const int m0 = tile_m * 128, n0 = g * (128 * NB);
    const bf16_t* zero = (const bf16_t*)&g_zero16_f;
    auto stamp = [&](int k) {
        if (p.dbg && tid == 0) p.dbg[(long)bid * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ---- issue the A panel (4 slabs x 2 pieces per wave) and the first two weight stages (2 pieces per wave each) ----
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 16 + i * 8 + (lane >> 3);
            const int c = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
            const bf16_t* src = (m0 + row < p.M) ? p.A + ((long)(m0 + row) * K + s * 64 + c) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + s * SLAB + (wave * 16 + i * 8) * 128), 16, 0, 0);
        }
    const bf16_t* b_ptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 16 + i * 8 + (lane >> 3);
        const int c = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        b_ptr[i] = p.W + ((long)(n0 + row) * K + c);          // N % 512 == 0: every weight row exists
    }
    // stage t = nb * KS + kt: weight rows n0 + nb*128 .., k = kt*64 ..
    auto issue_b = [&](int t, int buf) {
        const bool live = t < NB * KS;
        const long off = live ? ((long)(t / KS) * 128 * K + (t % KS) * 64) : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(live ? b_ptr[i] + off : zero), (lptr_t)(sB + buf * SLAB + (wave * 16 + i * 8) * 128), 16, 0, 0);
    };
#pragma unroll
    for (int t0 = 0; t0 < SB - 1; ++t0) issue_b(t0, t0);

    // ---- bn2 (scale, shift) for the K = 256 input channels, from conv2's integer sums (bn_table_from_acc's arithmetic) ----
    {
        const double inv = 1.0 / (kStatScale * p.count);
        for (int c = tid; c < K; c += NT) {
            long long s1 = 0, s2 = 0;
            for (int sh = 0; sh < p.in_shards; ++sh) {
                s1 += p.in_acc[(long)sh * 2 * K + c];
                s2 += p.in_acc[(long)sh * 2 * K + K + c];
            }
            const double mean = (double)s1 * inv;
            double var = (double)s2 * inv - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = 1.0f / sqrtf((float)var + p.eps);
            const float sc = p.in_gamma[c] * invstd;
            in_tab[c] = sc;
            in_tab[K + c] = p.in_beta[c] - (float)mean * sc;
            if (bid == 0) {
                if (p.in_rm) {
                    const double unbiased = p.count > 1.0 ? var * p.count / (p.count - 1.0) : var;
                    p.in_rm[c] = (float)((1.0 - p.momentum) * p.in_rm[c] + p.momentum * (double)(float)mean);
                    p.in_rv[c] = (float)((1.0 - p.momentum) * p.in_rv[c] + p.momentum * (double)(float)unbiased);
                }
                if (p.in_acc_clear)
                    for (int sh = 0; sh < p.in_shards; ++sh) {
                        p.in_acc_clear[(long)sh * 2 * K + c] = 0;
                        p.in_acc_clear[(long)sh * 2 * K + K + c] = 0;
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // my A pieces have landed (the four weight stages may still fly)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my table entries are written
    __builtin_amdgcn_s_barrier();                         // everybody's
    asm volatile("" ::: "memory");
    stamp(1);
    // ---- relu(x * scale + shift) on the A panel, in place, once ----
#pragma unroll
    for (int j = 0; j < KS * 128 * 8 / NT; ++j) {
        const int q = tid + j * NT;
        const int s = q >> 10, row = (q >> 3) & 127, pos = q & 7;
        if (m0 + row < p.M) {
            const int c0 = s * 64 + ((pos ^ ((row >> 1) & 7)) << 3);
            bf16x8 v = *(const bf16x8*)(sA + s * SLAB + row * 128 + pos * 16);
            const f32x4 s0 = *(const f32x4*)(in_tab + c0), s1 = *(const f32x4*)(in_tab + c0 + 4);
            const f32x4 t0 = *(const f32x4*)(in_tab + K + c0), t1 = *(const f32x4*)(in_tab + K + c0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (bf16_t)fmaxf((float)v[e] * s0[e] + t0[e], 0.0f);
                v[e + 4] = (bf16_t)fmaxf((float)v[e + 4] * s1[e] + t1[e], 0.0f);
            }
            *(bf16x8*)(sA + s * SLAB + row * 128 + pos * 16) = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (the first ring barrier below orders these writes before any fragment read)

    f32x16 acc[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nb][i][e] = 0.0f;

    int a_off[2][4], b_off[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a_off[i][ks] = row * 128 + (((2 * ks + h) ^ ((row >> 1) & 7)) << 4);
    }
    {
        const int row = wn * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b_off[ks] = row * 128 + (((2 * ks + h) ^ ((row >> 1) & 7)) << 4);
    }

    // ---- K phase: 16 weight stages through the ring (stage t in slot t % 3; two stages in flight) ----
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int kt = 0; kt < KS; ++kt) {
            const int t = nb * KS + kt;
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // my pieces of stage t have landed (three younger stages in flight)
            __builtin_amdgcn_s_barrier();                         // everybody's; everybody has finished reading stage t-1
            asm volatile("" ::: "memory");
            issue_b(t + SB - 1, (t + SB - 1) % SB);               // into the slot stage t-1 occupied
            const char* stA = sA + kt * SLAB;
            const char* stB = sB + (t % SB) * SLAB;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 a0 = *(const bf16x8*)(stA + a_off[0][ks]), a1 = *(const bf16x8*)(stA + a_off[1][ks]);
                const bf16x8 b = *(const bf16x8*)(stB + b_off[ks]);
                acc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[nb][0], 0, 0, 0);
                acc[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[nb][1], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(2);

    // ---- statistics of this workgroup's 128 x 512 f32 accumulators -> integer atomics ----
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float s = 0.0f, q = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = acc[nb][i][e];
                s += v;
                q += v * v;
            }
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (h == 0) {
            red[((wm * NB + nb) * 2 + 0) * 128 + wn * 32 + r] = s;
            red[((wm * NB + nb) * 2 + 1) * 128 + wn * 32 + r] = q;
        }
    }
    __syncthreads();
    {
        const int nb = tid >> 7, c = tid & 127;          // 512 threads = 512 columns
        const int col = n0 + nb * 128 + c;
        const float s = red[((0 * NB + nb) * 2 + 0) * 128 + c] + red[((1 * NB + nb) * 2 + 0) * 128 + c];
        const float q = red[((0 * NB + nb) * 2 + 1) * 128 + c] + red[((1 * NB + nb) * 2 + 1) * 128 + c];
        atomicAdd((unsigned long long*)(p.out_acc + col), (unsigned long long)__double2ll_rn((double)s * kStatScale));
        atomicAdd((unsigned long long*)(p.out_acc + p.N + col), (unsigned long long)__double2ll_rn((double)q * kStatScale));
    }
    // ---- residual blocks 0 and 1 -> LDS by LDS-DMA (4 pieces of 4 rows x 256 B per wave and block): they land while the grid
    //      barrier is waited for.  A landed block is read back as [row][16 chunks of 16 B]: conflict free.
    auto issue_r = [&](int nb, char* dst) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int piece = wave * 4 + it;
            const int row = piece * 4 + (lane >> 4);
            const bf16_t* src = (m0 + row < p.M) ? p.R + ((long)(m0 + row) * p.N + n0 + nb * 128 + (lane & 15) * 8) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + piece * 1024), 16, 0, 0);
        }
    };
    issue_r(0, rbuf0);                                   // (neither buffer overlaps `red` / `otab`; the A panel is dead)
    issue_r(1, rbuf1);
    // ---- grid barrier: my atomics are acknowledged (all but the 8 younger LDS-DMA pieces), the workgroup's are (barrier),
    //      ONE lane arrives and polls ----
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stamp(3);
    if (tid == 0) {
        int ok = 1;
        __hip_atomic_fetch_add(p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) {
            if ((++spins & 63u) == 0 &&
                (spins > p.spin_limit || __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u)) {
                __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        s_flag = ok;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool ok = s_flag != 0;
    stamp(4);

    if (ok) {
        // ---- bn3 (scale, shift) of my 512 columns from the now complete sums; tile_m == 0 also updates the running statistics
        //      and clears the other parity (bn_table_from_acc's arithmetic; the sums are read past L1/L2: sc1 loads) ----
        {
            const int col = n0 + tid;
            const double inv = 1.0 / (kStatScale * p.count);
            const long long s1 = __hip_atomic_load(p.out_acc + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long s2 = __hip_atomic_load(p.out_acc + p.N + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double mean = (double)s1 * inv;
            double var = (double)s2 * inv - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = 1.0f / sqrtf((float)var + p.eps);
            const float sc = p.out_gamma[col] * invstd;
            otab[tid] = sc;
            otab[512 + tid] = p.out_beta[col] - (float)mean * sc;
            if (tile_m == 0) {
                if (p.out_rm) {
                    const double unbiased = p.count > 1.0 ? var * p.count / (p.count - 1.0) : var;
                    p.out_rm[col] = (float)((1.0 - p.momentum) * p.out_rm[col] + p.momentum * (double)(float)mean);
                    p.out_rv[col] = (float)((1.0 - p.momentum) * p.out_rv[col] + p.momentum * (double)(float)unbiased);
                }
                if (p.out_acc_clear) {
                    p.out_acc_clear[col] = 0;
                    p.out_acc_clear[p.N + col] = 0;
                }
            }
        }
        // ---- per 128-column block: bf16 C tile through LDS, then y = relu(c * scale + shift + x) with x from the prefetched LDS
        //      block, 16 bytes per thread; the block after next is fetched into the buffer just consumed.  Raw barriers: a
        //      __syncthreads() would drain the LDS-DMA in flight. ----
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            char* rb = (nb & 1) ? rbuf1 : rbuf0;
            const int col = wn * 32 + r;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    *(bf16_t*)(sA + row * CROW + col * 2) = (bf16_t)acc[nb][i][e];
                }
            // my pieces of residual block nb have landed.  A wave's memory counter is in order: behind block nb's 4 pieces come
            // at most the previous block's stores and the 4 pieces of block nb+1 -- waiting for all but the 4 youngest
            // operations therefore covers block nb whether or not this wave stored anything (rows past M store nothing)
            if (nb < NB - 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // C tile + table + everybody's residual pieces are in LDS
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < 128 * 16 / NT; ++it) {
                const int qid = tid + it * NT;
                const int row = qid >> 4, cc = qid & 15;
                const int grow = m0 + row, gcol = n0 + nb * 128 + cc * 8;
                const bf16x8 c = *(const bf16x8*)(sA + row * CROW + cc * 16);
                const bf16x8 z = *(const bf16x8*)(rb + row * 256 + cc * 16);
                const float* ts = otab + nb * 128 + cc * 8;
                const f32x4 s0 = *(const f32x4*)ts, s1 = *(const f32x4*)(ts + 4);
                const f32x4 t0 = *(const f32x4*)(ts + 512), t1 = *(const f32x4*)(ts + 512 + 4);
                bf16x8 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[k] = (bf16_t)fmaxf((float)c[k] * s0[k] + t0[k] + (float)z[k], 0.0f);
                    o[k + 4] = (bf16_t)fmaxf((float)c[k + 4] * s1[k] + t1[k] + (float)z[k + 4], 0.0f);
                }
                if (grow < p.M) store16_wt(p.Y + (long)grow * p.N + gcol, *(const u32x4*)&o);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // everybody is done with the C tile and with rb
            asm volatile("" ::: "memory");
            if (nb + 2 < NB) issue_r(nb + 2, rb);
        }
    }
    stamp(5);
    // ---- departure: the last workgroup to leave re-arms the counters and releases the residency token ----
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)nwg - 1u) {
            __hip_atomic_store(p.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.hold_token) __hip_atomic_store(&g_resident_token, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

unsigned spin_limit_env() {
    static unsigned v = 0;
    if (!v) {
        const char* e = getenv("SAT_FUSED_SPIN_LIMIT");
        v = (e && atol(e) > 0) ? (unsigned)atol(e) : kSpinDefault;
    }
    return v;
}

int cu_count() {
    static int n = -1;
    if (n < 0) {
        int dev = 0, v = 0;
        n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? v : 0;
    }
    return n;
}

}  // namespace

void sat_conv_take_timer(hipEvent_t* start, hipEvent_t* stop);        // sat_conv_glds.hip: the armed diagnostic timer, consumed

// The residency token for OTHER kernels that need all their workgroups resident (the persistent LSTM recurrence): acquire ahead of
// the launch, release behind it, both as one-wave kernels on the same stream.
int sat_resident_token_acquire(unsigned* err, hipStream_t s) {
    hipLaunchKernelGGL(token_acquire_kernel, dim3(1), dim3(64), 0, s, err, spin_limit_env() * 64u);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}
// the token only costs launches once a kernel that shares it with others has run in this process
static int g_token_in_use = 0;
int sat_resident_token_in_use() { return g_token_in_use; }
int sat_resident_token_release(hipStream_t s) {
    hipLaunchKernelGGL(token_release_kernel, dim3(1), dim3(64), 0, s);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

static unsigned long long* g_dbg_stamps = nullptr;
// diagnostics: device buffer of [grid][8] u64 that lane 0 of every workgroup of the NEXT fused launches fills with s_memtime
// stamps (0 start, 1 A panel landed + table, 2 K phase done, 3 statistics acknowledged, 4 grid barrier passed, 5 epilogue done)
unsigned long long* sat_dbg_stamps() { return g_dbg_stamps; }      // conv_xp_kernel's stamps go to the same buffer
extern "C" int sat_conv3_fused_debug(void* stamps) {
    g_dbg_stamps = (unsigned long long*)stamps;
    return SAT_OK;
}

// 1 when SAT_OP_CONV3_FUSED can run this geometry on this device: K = 256, N % 512 == 0, grid <= CUs
extern "C" int sat_conv3_fused_ok(int64_t M, int N, int K) {
    if (K != 64 * KS || N < 128 * NB || (N % (128 * NB))) return 0;
    const long grid = (long)sat_cdiv(M, 128) * (N / (128 * NB));
    return (grid >= 1 && grid <= cu_count()) ? 1 : 0;
}

// SAT_OP_CONV3_FUSED (bf16): in0 = raw conv2 output [M][K], w [N][K], in1 = residual [M][N], out [M][N];
// stat_acc1 (+ gamma1 / beta1 / running_*1) = conv2's integer sums (bn2, input side); stat_acc (+ gamma / beta / running_*) = this
// conv's sums (bn3, output side); scale_out = uint32 sync words [2] (zero), shift_out = uint32 sticky error word.
int sat_conv3_fused_launch(const sat_op* op, int parity, hipStream_t s) {
    if (op->dtype != SAT_BF16 || !op->in0 || !op->w || !op->in1 || !op->out || !op->stat_acc || !op->stat_acc1 || !op->gamma ||
        !op->beta || !op->gamma1 || !op->beta1 || !op->scale_out || !op->shift_out || op->count < 1)
        return SAT_ERR_ARG;
    const long M = (long)op->N * op->Hout * op->Wout;
    if (op->KH != 1 || op->KW != 1 || op->stride != 1 || op->pad != 0 || !sat_conv3_fused_ok(M, op->Cout, op->Cin)) return SAT_ERR_UNSUPPORTED;
    if (op->in1 == op->out || op->in0 == op->out) return SAT_ERR_ARG;
    const int in_shards = op->stat_shards1 > 1 ? op->stat_shards1 : 1;
    if ((in_shards & (in_shards - 1)) || in_shards > 8 || op->stat_shards > 1) return SAT_ERR_ARG;
    FusedArgs a = {};
    a.A = (const bf16_t*)op->in0; a.W = (const bf16_t*)op->w; a.R = (const bf16_t*)op->in1; a.Y = (bf16_t*)op->out;
    a.M = (int)M; a.N = op->Cout; a.tiles_g = op->Cout / (128 * NB);
    long long* ib = (long long*)op->stat_acc1;
    a.in_acc = ib + (long)parity * in_shards * 2 * op->Cin;
    a.in_acc_clear = ib + (long)(1 - parity) * in_shards * 2 * op->Cin;
    a.in_shards = in_shards;
    a.in_gamma = op->gamma1; a.in_beta = op->beta1; a.in_rm = op->running_mean1; a.in_rv = op->running_var1;
    long long* ob = (long long*)op->stat_acc;
    a.out_acc = ob + (long)parity * 2 * op->Cout;
    a.out_acc_clear = ob + (long)(1 - parity) * 2 * op->Cout;
    a.out_gamma = op->gamma; a.out_beta = op->beta; a.out_rm = op->running_mean; a.out_rv = op->running_var;
    a.count = (double)op->count; a.momentum = op->momentum; a.eps = op->eps;
    a.sync = (unsigned*)op->scale_out; a.err = (unsigned*)op->shift_out;
    a.spin_limit = spin_limit_env();
    a.hold_token = 1;
    a.dbg = g_dbg_stamps;
    g_token_in_use = 1;
    SAT_TRY(sat_resident_token_acquire(a.err, s));
    const dim3 grid(sat_cdiv(M, 128) * a.tiles_g), block(512);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    sat_conv_take_timer(&e0, &e1);
    if (e0) hipExtLaunchKernelGGL(conv3_fused_kernel, grid, block, 0, s, e0, e1, 0, a);
    else hipLaunchKernelGGL(conv3_fused_kernel, grid, block, 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}
