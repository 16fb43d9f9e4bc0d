// Persistent LSTM recurrence (models.py:52 `self.lstm(packed)`, the serial part): ALL time steps of one layer in ONE
// launch, W_hh resident in registers, the per-step hidden-state exchange through data-tagged granules.
//
// The recurrence h_t = f(x-gates_t + h_{t-1} W_hh^T) is independent per batch row, and its weight matrix is the
// same every step.  So the batch is cut into GROUPS of 8 rows and the 4H gate columns of one group are cut over
// MEMBER workgroups of 16 hidden units (64 gate columns) each:
//   * a member loads its W_hh slice [64 columns x H] ONCE into registers in MFMA B-operand layout (128 VGPRs at
//     H = 512) -- the per-step kernels re-read 4 MB of weights from L2 every step;
//   * per step it needs only its group's h_{t-1} (8 rows x H, 16 KB), not the whole batch: each member publishes its
//     8 x 16 slice as SELF-TAGGED 4-byte words -- |h| <= 1, so bit 30 of its f32 pattern (the top exponent bit) is always
//     0 and carries the hand-off phase instead: the data IS the flag (cdna_hip_programming.md Guideline 16, R2) at 4 bytes per
//     value instead of an 8-byte {tag, value} granule -- written with write-through (sc1) 16-byte stores, and sweeps the
//     group's [8 rows x H] words with sc1 16-byte loads until every word shows the phase, re-reading only the pieces that
//     did not (round 4: the exchange moves half the bytes and a quarter of the load instructions; the poll traffic was 9x the
//     kernel's other bytes, profiles/r03_pmc_traffic.json); two parity buffers, since a member can run at most one step ahead
//     of the slowest member of its group;
//   * groups never talk to each other: there is no grid-wide barrier, and a group whose rows have all ended
//     (packed sequences: batch_sizes[t] shrinks) simply leaves;
//   * gates = v_mfma_f32_16x16x4_f32 (exact f32: same K permutation on both operands), gate math fused, c_t in a
//     register for the whole sequence; tapes (activated gates, c, h, h_prev) are written for the backward.
// Every spin is bounded: on a timeout the kernel raises a sticky error word and every workgroup drains.
//
// Needs groups * members <= CUs (all workgroups co-resident); the host falls back to one launch per step otherwise.
#include "sat_internal.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

namespace {

constexpr int kMaxT = 64;
constexpr int kRows = 8;          // batch rows per group
constexpr int kUnits = 16;        // hidden units per member workgroup (4 waves x 4 units)
constexpr unsigned kSpinLimit = 1u << 22;   // default bound of one hand-off wait, in sweeps (SAT_LSTM_SPIN_LIMIT overrides)

struct PersistArgs {
    float* GA;            // [N][4H] in: x-gates (+ both biases); out: activated gates i,f,g,o
    const float* W;       // [4H][H]
    float* CS;            // [N][H]
    float* HS;            // [N][H]
    float* HP;            // [N][H]  (rows of step 0 pre-zeroed by the host)
    unsigned* xch;        // [2 parities][groups][8 rows][H] self-tagged words, zeroed per call
    unsigned* err;        // sticky timeout word (zeroed per call; the CALLER reads it back: sat_lstm_fwd_status_offset)
    unsigned spin_limit;  // sweeps a workgroup waits for its group before it gives up
    int dbg_stall;        // test build only (-DSAT_TESTHOOKS, SAT_LSTM_DEBUG_STALL=1): workgroup 0 never publishes -> its group times out; always 0 in the product library
    int xcd_map;          // 1: group = blockIdx.x % groups, member = blockIdx.x / groups -- with 8 groups the members of a group have equal blockIdx.x % 8,
                          // i.e. share an XCD under round-robin placement (speed only, never correctness; SAT_LSTM_XCD_MAP, measured: profiles/r05_lstm_ab.txt)
    int H, T, B, members;
    int prefix[kMaxT + 1];
};

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

// 16-byte write-through (sc1) buffer accesses: what `global_load/store_dwordx4 ... sc1` does, with the compiler keeping the
// wait counts (aux 16 = sc1: the access bypasses this CU's L1 and goes through to memory -- agent-scope visibility)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xch_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 load16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ void store16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, (int)byte_off, 0, 16);
}
// hidden-state word <-> self-tagged word: bit 30 of a finite h (|h| <= 1) is 0 and carries the phase; a NaN travels as the
// pattern of 1.5 (exponent 127 with a non-zero mantissa: no |h| <= 1 has it) so that a diverged run stays a diverged run
constexpr unsigned kPhaseBit = 0x40000000u, kNanCode = 0x3fc00000u;
__device__ __forceinline__ unsigned tag_h(float h, unsigned phase) {
    unsigned b = __float_as_uint(h);
    if ((b & 0x7f800000u) == 0x7f800000u) b = (b & 0x80000000u) | kNanCode;
    return (b & ~kPhaseBit) | (phase ? kPhaseBit : 0u);
}
__device__ __forceinline__ float untag_h(unsigned w) {
    w &= ~kPhaseBit;
    return ((w & 0x7fffffffu) == kNanCode) ? __uint_as_float(0x7fc00000u) : __uint_as_float(w);
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// KR = batch rows per group: 8 (half of every 16-row MFMA tile is padding, twice the groups), or 16 -- round 5: a full tile, half the
// workgroups; what lets H = 1024 (64 members per group: 4 groups x 64 = 256 workgroups at batch 64, W_hh slice = 256 registers per
// lane across VGPRs + AGPRs, one wave per SIMD) run persistently at all
template <int NKB, int KR>      // H = 16 * NKB, NKB even
__global__ __launch_bounds__(256) void lstm_persist_kernel(const PersistArgs p) {
    constexpr int kRows = KR;
    constexpr int H = 16 * NKB;
    constexpr int HROW = H + 4;                       // LDS row stride in floats: rows land 16 B apart in the bank row
    __shared__ __attribute__((aligned(16))) float h_lds[kRows * HROW];
    __shared__ __attribute__((aligned(16))) float c_lds[4][kRows][16];
    __shared__ int s_abort;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = gridDim.x / p.members;
    const int group = p.xcd_map ? (int)(blockIdx.x % ngroups) : (int)(blockIdx.x / p.members);
    const int member = p.xcd_map ? (int)(blockIdx.x / ngroups) : (int)(blockIdx.x - group * p.members);
    const int row0 = group * kRows;                   // first batch row of this group
    const int n16 = lane & 15, kg = lane >> 4;
    const int u0 = member * kUnits + wave * 4;        // this wave's 4 hidden units
    const long wrow = (long)(n16 >> 2) * H + u0 + (n16 & 3);     // weight row feeding MFMA column n16: gate n16>>2, unit n16&3

    // ---- W_hh slice -> registers, once: lane (n16, kg) holds W[wrow][16 kb + 4 kg .. +3] ----
    f32x4 wreg[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) wreg[kb] = *(const f32x4*)(p.W + wrow * H + kb * 16 + kg * 4);

    for (int i = tid; i < kRows * HROW; i += 256) h_lds[i] = 0.0f;      // h_{-1} = 0
    if (tid == 0) s_abort = 0;
    __syncthreads();

    // epilogue ownership: lanes 0 .. 4 KR - 1 of each wave own (row = lane>>2, unit = u0 + (lane&3)); c lives in a register
    const int erow = lane >> 2, eu = u0 + (lane & 3);
    float c_reg = 0.0f;
    constexpr unsigned kSlab = kRows * H * 4;                           // bytes per (parity, group): [8 rows][H] words
    constexpr int NCHUNK = kRows * H / 4;                                // 16-byte pieces of a slab
    constexpr int NIT = (NCHUNK + 255) / 256;                            // pieces per thread
    const int groups = gridDim.x / p.members;

    for (int t = 0; t < p.T; ++t) {
        const int bs = p.prefix[t + 1] - p.prefix[t];
        const int active = bs - row0;                                     // rows of this group still running
        if (active <= 0) break;                                           // batch_sizes never grow: nothing left for this group
        const int bs_next = (t + 1 < p.T) ? p.prefix[t + 2] - p.prefix[t + 1] : 0;

        // ---- h_{t-1} of my group: sweep the group's [8 x H] words of step t-1 until every one shows that step's phase ----
        if (t > 0) {
            const unsigned phase = (((unsigned)(t - 1) >> 1) & 1u) ^ 1u;    // use k = (t-1)/2 of this parity buffer: 1, 0, 1, ... (zeroed buffer = 0)
            const __amdgpu_buffer_rsrc_t src = xch_rsrc((const char*)p.xch + ((long)((t - 1) & 1) * groups + group) * kSlab, kSlab);
            const unsigned want = phase ? kPhaseBit : 0u;
            unsigned pending = (1u << NIT) - 1u;                          // bit k: piece tid + 256 k not yet seen complete
#pragma unroll
            for (int k = 0; k < NIT; ++k)
                if (tid + k * 256 >= NCHUNK) pending &= ~(1u << k);
            unsigned spins = 0;
            bool fail = false;
            for (;;) {
                u32x4 x[NIT];
#pragma unroll
                for (int k = 0; k < NIT; ++k)                             // all of a thread's pieces in flight together
                    if (pending & (1u << k)) x[k] = load16_sc1(src, (unsigned)(tid + k * 256) * 16u);
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    if (!(pending & (1u << k))) continue;
                    const u32x4 w = x[k];
                    if (((w[0] & w[1] & w[2] & w[3]) & kPhaseBit) == want && (((w[0] | w[1] | w[2] | w[3]) & kPhaseBit) == want)) {
                        const int i = tid + k * 256, r = i / (H / 4), c4 = i - r * (H / 4);
                        f32x4 v = {untag_h(w[0]), untag_h(w[1]), untag_h(w[2]), untag_h(w[3])};
                        *(f32x4*)(h_lds + r * HROW + c4 * 4) = v;
                        pending &= ~(1u << k);
                    }
                }
                if (__syncthreads_and(pending == 0u ? 1 : 0)) break;
                if ((++spins & 63u) == 0) {            // every 64 sweeps: bounded spin + another workgroup's verdict
                    if (tid == 0 && (spins > p.spin_limit ||
                                     __hip_atomic_load((gu32*)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))
                        s_abort = 1;
                    __syncthreads();
                    if (s_abort) { fail = true; break; }
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (fail) {
                if (tid == 0) __hip_atomic_store((gu32*)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }

        // ---- the epilogue's own operands, fetched under the MFMAs: x-gates of (row, unit) ----
        float xg[4] = {0.f, 0.f, 0.f, 0.f};
        const bool own = lane < 4 * kRows && erow < active;
        const long prow = (long)p.prefix[t] + row0 + erow;               // packed row of (t, batch row)
        if (own) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xg[g] = p.GA[prow * 4 * H + (long)g * H + eu];
        }

        // ---- gates += h_{t-1} W_hh^T : 16 rows (8 real) x 16 columns per wave, K = H ----
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const bool arow = n16 < kRows;                                    // rows 8..15 of the MFMA tile are padding
        const float* hsrc = h_lds + n16 * HROW + kg * 4;
#pragma unroll
        for (int kb = 0; kb < NKB; kb += 2) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            if (arow) {
                a0 = *(const f32x4*)(hsrc + kb * 16);
                a1 = *(const f32x4*)(hsrc + kb * 16 + 16);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], wreg[kb][e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], wreg[kb + 1][e], acc1, 0, 0, 0);
            }
        }
        // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg -> rows 0..KR-1 sit in lanes with 4 kg < KR
        if (kg * 4 < kRows) {
#pragma unroll
            for (int e = 0; e < 4; ++e) c_lds[wave][kg * 4 + e][n16] = acc0[e] + acc1[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // wave-local hand-off through LDS
        __builtin_amdgcn_wave_barrier();

        if (lane < 4 * kRows) {
            float g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) g4[g] = c_lds[wave][erow][g * 4 + (lane & 3)] + xg[g];
            const float gi = sigm(g4[0]), gf = sigm(g4[1]), gg = tanhf(g4[2]), go = sigm(g4[3]);
            const float c_new = gf * c_reg + gi * gg;
            const float h_new = go * tanhf(c_new);
            c_reg = c_new;
            if (own) {
                float* ga = p.GA + prow * 4 * H;
                ga[eu] = gi; ga[H + eu] = gf; ga[2 * H + eu] = gg; ga[3 * H + eu] = go;
                p.CS[prow * H + eu] = c_new;
                p.HS[prow * H + eu] = h_new;
                if (row0 + erow < bs_next) p.HP[((long)p.prefix[t + 1] + row0 + erow) * H + eu] = h_new;
            }
            // publish h_t for the group: the wave's 4 units of a row go out as ONE 16-byte write-through store (lane & 3 == 0
            // collects its three neighbours' words), phase bit = the parity buffer's use count, parity t & 1
            {
                const unsigned phase = (((unsigned)t >> 1) & 1u) ^ 1u;
                const unsigned w0 = tag_h(h_new, phase);
                u32x4 w;
                w[0] = w0;
                w[1] = (unsigned)__shfl_down((int)w0, 1, 64);
                w[2] = (unsigned)__shfl_down((int)w0, 2, 64);
                w[3] = (unsigned)__shfl_down((int)w0, 3, 64);
                if ((lane & 3) == 0 && t + 1 < p.T && !(p.dbg_stall && group == 0 && member == 0)) {
                    const __amdgpu_buffer_rsrc_t dst = xch_rsrc((const char*)p.xch + ((long)(t & 1) * groups + group) * kSlab, kSlab);
                    store16_sc1(dst, (unsigned)(erow * H + u0) * 4u, w);
                }
            }
        }
        __syncthreads();         // c_lds / h_lds are rewritten by the next step
    }
}


// ---- persistent BACKWARD recurrence (train.py:144 through models.py:52) -------------------------------------------------------
// dh_t = dHS_t + DG_{t+1} W_hh, gate backward, dc carried -- all time steps in ONE launch, same decomposition as the forward:
// groups of 8 batch rows, member workgroups of 16 hidden units.  The contraction runs over K = 4H gate columns, and a member
// PRODUCES exactly 64 of them (its 16 units x 4 gates).  So the product is split over K by member: each member multiplies ITS
// OWN fresh d(pre-activation) tile [8 x 64] with its W_hh slice [64 x H] (register resident: 128 VGPRs at H = 512) into a partial
// dh for ALL H units and hands every other member the 8 x 16 block that member owns -- the exchange carries 8 x 16 granules per
// (source, destination) pair and a member sweeps members x 128 granules per step, the forward's volume, summing the partials in
// member order as they arrive in its registers (no LDS staging; even / odd sources in two thread halves, combined in a fixed
// order).  dc stays in a register for the whole sequence.  Going backwards in time a group JOINS when its rows start
// (batch_sizes shrink with t); it never leaves.
struct PersistBwdArgs {
    const float* dHS;     // [N][H]
    const float* GA;      // [N][4H] activated gates i,f,g,o
    const float* CS;      // [N][H]
    const float* W;       // [4H][H]
    float* DG;            // [N][4H] out
    unsigned long long* xch;   // [2 parities][groups][dest member][src member][8 rows][16 units] granules (NOT zeroed: epoch tags)
    unsigned* err;
    unsigned spin_limit;
    unsigned epoch;       // call counter: tags are epoch * 128 + step + 1, so granules of an earlier call never match (no 17 MB memset)
    int dbg_stall;
    int xcd_map;
    int H, T, B, members;
    int prefix[kMaxT + 1];
};

template <int NKB>      // H = 16 * NKB (NKB = members)
__global__ __launch_bounds__(256) void lstm_persist_bwd_kernel(const PersistBwdArgs p) {
    constexpr int H = 16 * NKB, K4 = 4 * H;
    constexpr int TPW = (NKB + 3) / 4;                 // destination members (16-column tiles) per wave
    __shared__ __attribute__((aligned(16))) float d_lds[kRows][64 + 4];      // my DG_t tile: [row][gate * 16 + unit]
    __shared__ float half_lds[128];
    __shared__ int s_abort;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = gridDim.x / p.members;
    const int group = p.xcd_map ? (int)(blockIdx.x % ngroups) : (int)(blockIdx.x / p.members);
    const int member = p.xcd_map ? (int)(blockIdx.x / ngroups) : (int)(blockIdx.x - group * p.members);
    const int row0 = group * kRows;
    const int n16 = lane & 15, kg = lane >> 4;
    const int u0 = member * kUnits;

    // ---- W_hh slice -> registers: rows = my 64 gate columns (local k = gate * 16 + unit), columns = all H hidden units; wave w
    //      owns destination members w * TPW ..; lane (n16, kg) holds W[gate*H + u0 + unit][dest * 16 + n16] for local
    //      k = 16 kb + 4 kg + e ----
    f32x4 wreg[4][TPW];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            const int dest = wave * TPW + tl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kl = kb * 16 + kg * 4 + e;                     // gate = kl >> 4, unit = kl & 15
                wreg[kb][tl][e] = dest < NKB ? p.W[(long)((kl >> 4) * H + u0 + (kl & 15)) * H + dest * 16 + n16] : 0.0f;
            }
        }
    if (tid == 0) s_abort = 0;
    __syncthreads();

    // epilogue ownership: threads 0..127 own (row = tid >> 4, unit = u0 + (tid & 15)); dc lives in a register
    const int erow = (tid & 127) >> 4, eu = u0 + (tid & 15);
    float dc_reg = 0.0f;
    gu64* xch = (gu64*)p.xch;
    const long pair = (long)kRows * 16;                                     // granules per (dest, src) pair
    const long slab = (long)p.members * p.members * pair;                   // granules per (parity, group)
    const int groups = gridDim.x / p.members;

    int Tg = 0;                                                             // the steps in which this group has rows: t in [0, Tg)
    for (int t = 0; t < p.T; ++t)
        if (p.prefix[t + 1] - p.prefix[t] - row0 > 0) Tg = t + 1;

    for (int t = Tg - 1; t >= 0; --t) {
        const int active = p.prefix[t + 1] - p.prefix[t] - row0;           // > 0 for every t < Tg
        const int next_active = (t + 1 < Tg) ? p.prefix[t + 2] - p.prefix[t + 1] - row0 : 0;

        // ---- the epilogue's own operands (issued before the sweep: they do not depend on it) ----
        const bool own = tid < 128 && erow < active;
        const long prow = (long)p.prefix[t] + row0 + erow;
        float dhs = 0.f, gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cs = 0.f, cprev = 0.f;
        if (own) {
            dhs = p.dHS[prow * H + eu];
            const float* ga = p.GA + prow * K4;
            gi = ga[eu]; gf = ga[H + eu]; gg = ga[2 * H + eu]; go = ga[3 * H + eu];
            cs = p.CS[prow * H + eu];
            if (t > 0) cprev = p.CS[((long)p.prefix[t - 1] + row0 + erow) * H + eu];
        }

        // ---- the recurrent term: partial dh blocks of step t+1 from every member of my group (tag = t + 2), summed per
        //      (row, unit) over the source members in a fixed order: thread half q sums sources q, q+2, q+4, ... ----
        float rec = 0.0f;
        if (t + 1 < Tg) {
            const gu64* src = xch + (((long)((t + 1) & 1) * groups + group) * p.members + member) * p.members * pair;
            const unsigned want = p.epoch * 128u + (unsigned)(t + 2);
            const int q = tid >> 7, cell = tid & 127;
            unsigned spins = 0;
            bool fail = false;
            static_assert(NKB <= 32, "one 16-granule batch per thread half");
            // granule k of this thread: source member q + 2k.  A sweep re-reads only the granules that have not shown the tag yet
            // (their values stay in registers), so a hand-off that arrives piecemeal costs its bytes once, not once per sweep
            unsigned long long x[16];
            unsigned pending = 0u;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (q + 2 * k < NKB) pending |= 1u << k;
            for (;;) {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (pending & (1u << k))
                        x[k] = __hip_atomic_load(src + (long)(q + 2 * k) * pair + cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if ((pending & (1u << k)) && (unsigned)(x[k] >> 32) == want) pending &= ~(1u << k);
                if (__syncthreads_and(pending == 0u ? 1 : 0)) {
                    float sum = 0.0f;
#pragma unroll
                    for (int k = 0; k < 16; ++k)                      // fixed order over the source members: bitwise reproducible
                        if (q + 2 * k < NKB) sum += __uint_as_float((unsigned)x[k]);
                    rec = sum;
                    break;
                }
                if ((++spins & 63u) == 0) {
                    if (tid == 0 && (spins > p.spin_limit ||
                                     __hip_atomic_load((gu32*)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))
                        s_abort = 1;
                    __syncthreads();
                    if (s_abort) { fail = true; break; }
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (fail) {
                if (tid == 0) __hip_atomic_store((gu32*)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            if (tid >= 128) half_lds[tid & 127] = rec;
        }
        __syncthreads();

        if (tid < 128) {
            float dh = dhs;
            if (erow < next_active) dh += rec + half_lds[tid];        // even sources + odd sources: fixed order
            const float dcn = erow < next_active ? dc_reg : 0.0f;
            const float tc = tanhf(cs);
            const float d_o = dh * tc;
            const float dc = dh * go * (1.0f - tc * tc) + dcn;
            float d4[4];
            d4[0] = dc * gg * gi * (1.0f - gi);
            d4[1] = dc * cprev * gf * (1.0f - gf);
            d4[2] = dc * gi * (1.0f - gg * gg);
            d4[3] = d_o * go * (1.0f - go);
            dc_reg = dc * gf;
            if (!own) { d4[0] = d4[1] = d4[2] = d4[3] = 0.0f; dc_reg = 0.0f; }
            if (own) {
                float* dg = p.DG + prow * K4;
                dg[eu] = d4[0]; dg[H + eu] = d4[1]; dg[2 * H + eu] = d4[2]; dg[3 * H + eu] = d4[3];
            }
#pragma unroll
            for (int gte = 0; gte < 4; ++gte) d_lds[erow][gte * 16 + (tid & 15)] = d4[gte];
        }
        if (t == 0) break;                          // nothing consumes a hand-off of step 0
        __syncthreads();

        // ---- my share of dh_{t-1}: P[8 x H] = DG_t[8 x 64 (mine)] W_hh[64 x H]; tile tl of wave w belongs to member w*TPW + tl ----
        f32x4 acc[TPW];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) acc[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (n16 < kRows) a = *(const f32x4*)(&d_lds[n16][kb * 16 + kg * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl)
                    acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wreg[kb][tl][e], acc[tl], 0, 0, 0);
        }
        // C/D map: col = lane & 15 (unit of the destination member), row = (lane >> 4) * 4 + reg: rows 0..7 sit in lanes kg < 2
        if (kg < 2 && !(p.dbg_stall && group == 0 && member == 0)) {
            gu64* dst0 = xch + ((long)(t & 1) * groups + group) * slab;
            const unsigned long long tag = (unsigned long long)(p.epoch * 128u + (unsigned)(t + 1)) << 32;
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) {
                const int dest = wave * TPW + tl;
                if (dest < NKB) {
                    gu64* dst = dst0 + ((long)dest * p.members + member) * pair + n16;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        __hip_atomic_store(dst + (kg * 4 + e) * 16, tag | (unsigned long long)__float_as_uint(acc[tl][e]),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();         // d_lds / half_lds are rewritten by the next step
    }
}

}  // namespace

bool sat_lstm_persist_has(int H);

// workspace: the exchange (2 parities) + the error word, in one block that is zeroed per call.  Sized for 16-row groups (>= what
// 8-row groups need): the rows per group are chosen at launch (sat_lstm_persist_rows)
extern "C" int64_t sat_lstm_fwd_ws_bytes(int B, int H) {
    if (H < 16 || (H % 16)) return 0;
    const int64_t rows = (B + 15) / 16 * 16;
    return 2 * rows * H * 4 + 64;
}

// Byte offset of the recurrence's STATUS WORD (u32) inside the sat_lstm_fwd workspace: 0 after a clean run, non-zero when a
// workgroup of the persistent launch gave up waiting for its group (the tapes / HS of that call are then INVALID).  The word is
// zeroed by every call; the caller copies it out behind the call and raises -- it must never be ignored.
extern "C" int64_t sat_lstm_fwd_status_offset(int B, int H) {
    const int64_t n = sat_lstm_fwd_ws_bytes(B, H);
    return n > 0 ? n - 64 : -1;
}

// process-wide switch of the persistent recurrence (returns the previous setting): after a timeout the host side turns it off,
// so later calls take the one-launch-per-step path, which needs no co-residency
static int g_persist_enabled = 1;
extern "C" int sat_lstm_persist_enable(int on) {
    const int prev = g_persist_enabled;
    g_persist_enabled = on ? 1 : 0;
    return prev;
}

// rows per group of the persistent FORWARD recurrence for this layer, 0 = it cannot run (all workgroups must be co-resident: one per
// CU): 8 where ceil(B / 8) * H / 16 workgroups fit, else 16 (H = 1024 at batch 64).  SAT_LSTM_ROWS=16 prefers 16 wherever it fits
// (half the workgroups: profiles/r05_lstm_ab.txt).
bool sat_lstm_persist_has16(int H);
int sat_lstm_persist_rows(int B, int H, int T, int n_cu) {
    if (!g_persist_enabled || T > kMaxT || T < 1 || H < 16 || (H % 16)) return 0;
    static const int prefer16 = getenv("SAT_LSTM_ROWS") ? atoi(getenv("SAT_LSTM_ROWS")) == 16 : 0;
    const int members = H / kUnits;
    const bool fit8 = sat_lstm_persist_has(H) && ((B + 7) / 8) * members <= n_cu;
    const bool fit16 = sat_lstm_persist_has16(H) && ((B + 15) / 16) * members <= n_cu;
    if (fit16 && (prefer16 || !fit8)) return 16;
    return fit8 ? 8 : 0;
}

// can the persistent BACKWARD kernel run this layer?  (8-row groups only; all workgroups co-resident)
bool sat_lstm_persist_ok(int B, int H, int T, int n_cu) {
    if (!g_persist_enabled) return false;
    if (!sat_lstm_persist_has(H) || T > kMaxT || T < 1) return false;
    const int groups = (B + kRows - 1) / kRows, members = H / kUnits;
    return groups * members <= n_cu;
}

int sat_lstm_persist_launch(float* GA, const float* W, float* CS, float* HS, float* HP, const int32_t* batch_sizes, int T,
                            int H, int rows, void* workspace, int64_t ws_bytes, hipStream_t s) {
    const int B = batch_sizes[0];
    if (rows != 8 && rows != 16) return SAT_ERR_ARG;
    const int groups = (B + rows - 1) / rows, members = H / kUnits;
    const int64_t need = sat_lstm_fwd_ws_bytes(B, H);
    if (!workspace || ws_bytes < need) return SAT_ERR_WORKSPACE;
    PersistArgs a = {};
    a.GA = GA; a.W = W; a.CS = CS; a.HS = HS; a.HP = HP;
    a.xch = (unsigned*)workspace;
    a.err = (unsigned*)((char*)workspace + (need - 64));
    a.H = H; a.T = T; a.B = B; a.members = members;
    const char* sl = getenv("SAT_LSTM_SPIN_LIMIT");
    a.spin_limit = (sl && atol(sl) > 0) ? (unsigned)atol(sl) : kSpinLimit;
#ifdef SAT_TESTHOOKS        // fault injection exists only in the test build of the library (tests/_build/libsat_hip_testhooks.so)
    const char* ds = getenv("SAT_LSTM_DEBUG_STALL");
    a.dbg_stall = (ds && ds[0] == '1') ? 1 : 0;
#endif
    static const int xcd_map = getenv("SAT_LSTM_XCD_MAP") ? atoi(getenv("SAT_LSTM_XCD_MAP")) & 1 : 0;      // (bit 0: forward, bit 1: backward)
    a.xcd_map = xcd_map;
    a.prefix[0] = 0;
    for (int t = 0; t < T; ++t) a.prefix[t + 1] = a.prefix[t] + batch_sizes[t];
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)need, s);
    if (e != hipSuccess) return (int)e;
    // (all workgroups must be resident together; the only other kernels of this library that spin are this one's own instances)
    const dim3 grid(groups * members), block(256);
    if (rows == 16) {
        switch (H / 16) {
#define SAT_PERSIST_CASE(n) case n: hipLaunchKernelGGL((lstm_persist_kernel<n, 16>), grid, block, 0, s, a); break;
            SAT_PERSIST_CASE(16) SAT_PERSIST_CASE(32) SAT_PERSIST_CASE(64)
#undef SAT_PERSIST_CASE
            default: return SAT_ERR_UNSUPPORTED;
        }
    } else {
        switch (H / 16) {
#define SAT_PERSIST_CASE(n) case n: hipLaunchKernelGGL((lstm_persist_kernel<n, 8>), grid, block, 0, s, a); break;
            SAT_PERSIST_CASE(2) SAT_PERSIST_CASE(4) SAT_PERSIST_CASE(6) SAT_PERSIST_CASE(8) SAT_PERSIST_CASE(16) SAT_PERSIST_CASE(32)
#undef SAT_PERSIST_CASE
            default: return SAT_ERR_UNSUPPORTED;
        }
    }
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// granule exchange of the persistent BACKWARD recurrence (+ 64 bytes for its status word)
int64_t sat_lstm_persist_bwd_ws_bytes(int B, int H) {
    if (H < 16 || (H % 16)) return 0;
    const int64_t groups = (B + kRows - 1) / kRows, members = H / kUnits;
    return 2 * groups * members * members * kRows * 16 * 8 + 64;
}

static std::mutex g_xch_mu;
static std::unordered_map<void*, unsigned> g_xch_seen;   // exchange buffer -> generation it was last cleared in
static unsigned g_xch_epoch = 0, g_xch_generation = 1;

// the owner of a full backward workspace frees it (or hands the memory to somebody else): the next buffer at that address is
// cleared again before its first use
extern "C" int sat_lstm_ws_release(void* workspace) {
    std::lock_guard<std::mutex> lk(g_xch_mu);
    g_xch_seen.erase(workspace);
    return SAT_OK;
}

int sat_lstm_persist_bwd_launch(const float* dHS, const float* GA, const float* CS, const float* W, float* DG,
                                const int32_t* batch_sizes, int T, int H, void* xch, unsigned* err, hipStream_t s) {
    const int B = batch_sizes[0];
    const int groups = (B + kRows - 1) / kRows, members = H / kUnits;
    PersistBwdArgs a = {};
    a.dHS = dHS; a.GA = GA; a.CS = CS; a.W = W; a.DG = DG;
    a.xch = (unsigned long long*)xch; a.err = err;
    a.H = H; a.T = T; a.B = B; a.members = members;
    const char* sl = getenv("SAT_LSTM_SPIN_LIMIT");
    a.spin_limit = (sl && atol(sl) > 0) ? (unsigned)atol(sl) : kSpinLimit;
#ifdef SAT_TESTHOOKS
    const char* ds = getenv("SAT_LSTM_DEBUG_STALL");
    a.dbg_stall = (ds && ds[0] == '2') ? 1 : 0;
#endif
    // members of a group on equal blockIdx.x % 8 (one XCD under round-robin placement; speed only): measured at cfg 2 (round 5,
    // profiles/r05_lstm_ab.txt, tools/micro_lstm.py, two processes each): the BACKWARD's K-split exchange of 8-byte granules gains
    // 11 % (241.5 -> 214 us per call), the FORWARD's 4-byte self-tagged words lose 25 % (149.7 -> 187 us) -- so only here by default
    static const int xcd_map = getenv("SAT_LSTM_XCD_MAP") ? (atoi(getenv("SAT_LSTM_XCD_MAP")) >> 1) & 1 : 1;
    a.xcd_map = xcd_map && ((groups & (groups - 1)) == 0) && groups >= 8;
    a.prefix[0] = 0;
    for (int t = 0; t < T; ++t) a.prefix[t + 1] = a.prefix[t] + batch_sizes[t];
    // Tags are epoch * 128 + step + 1 with a process-wide call counter, so the exchange is never cleared per call.  The invariant
    // "no foreign bit pattern in the exchange region" lives HERE, in one place (round 5): the library clears a buffer the first
    // time it sees its address (and again when the 24-bit counter wraps: tags of 2^24 calls ago could match), remembers it, and
    // forgets it when the owner says the memory is gone (sat_lstm_ws_release) -- callers need not zero anything.
    bool clear = false;
    {
        std::lock_guard<std::mutex> lk(g_xch_mu);
        if (++g_xch_epoch >= (1u << 24)) { g_xch_epoch = 1; ++g_xch_generation; }
        a.epoch = g_xch_epoch;
        if (g_xch_seen.size() > 4096) g_xch_seen.clear();      // bounded: a forgotten buffer only costs it one clear
        auto it = g_xch_seen.find(xch);
        if (it == g_xch_seen.end()) { g_xch_seen.emplace(xch, g_xch_generation); clear = true; }
        else if (it->second != g_xch_generation) { it->second = g_xch_generation; clear = true; }
    }
    hipError_t e = hipMemsetAsync(err, 0, 64, s);
    if (e != hipSuccess) return (int)e;
    if (clear) {
        e = hipMemsetAsync(xch, 0, (size_t)(sat_lstm_persist_bwd_ws_bytes(B, H) - 64), s);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid(groups * members), block(256);
    switch (H / 16) {
#define SAT_PERSIST_CASE(n) case n: hipLaunchKernelGGL((lstm_persist_bwd_kernel<n>), grid, block, 0, s, a); break;
        SAT_PERSIST_CASE(2) SAT_PERSIST_CASE(4) SAT_PERSIST_CASE(6) SAT_PERSIST_CASE(8) SAT_PERSIST_CASE(16) SAT_PERSIST_CASE(32)
#undef SAT_PERSIST_CASE
        default: return SAT_ERR_UNSUPPORTED;
    }
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

bool sat_lstm_persist_has(int H) {
    switch (H / 16) { case 2: case 4: case 6: case 8: case 16: case 32: return (H % 16) == 0; default: return false; }
}
bool sat_lstm_persist_has16(int H) {
    switch (H / 16) { case 16: case 32: case 64: return (H % 16) == 0; default: return false; }
}
