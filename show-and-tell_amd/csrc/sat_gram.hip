// Train-mode bn3 WITHOUT a pass over conv3's output (`self.resnet(images)`, models.py:27, with nothing calling .eval(): BatchNorm2d
// normalises with batch statistics).
//
// The last BatchNorm of a bottleneck needs the batch mean / variance of c3 = a2 W3^T before conv3's epilogue can apply
// bn3 + residual add + ReLU -- and until round 5 that cost a launch of its own over the 4x-wide tensor (50 normalise+add launches,
// 42 % of the encoder's memory traffic, at the HBM roof).  The statistics are a linear / quadratic form of conv3's INPUT:
//     mean_c = w_c . mu            mu = sum_m a2[m] / M
//     var_c  = w_c^T (G / M - mu mu^T) w_c          G = a2^T a2   (P x P, P = planes = Cout / 4)
// so they can be had from a2 alone: a K = M GEMM of the [M][P] tensor with itself (a quarter of conv3's flops, one read of the
// quarter-wide c2), a centring in f64, and a small quadratic form.  conv3 then runs with the inference epilogue it already has
// (per-channel affine + residual + ReLU): the raw c3 tensor and the normalise+add launch never exist.
// Numerics (tools/gram_numerics.py, tests/test_gpu_gram.py): the centring happens in f64 on the COVARIANCE OF THE INPUTS, so the
// variance comes out to ~3e-7 relative -- the sums of squares of the outputs (the other route) are worse by E[c^2] / var.
//
//   SAT_OP_GRAM          gram_kernel: per (row slab, 128 x 128 tile pair ti <= tj, group) the partial G tile and column sums in
//                        f32 (MFMA accumulate): a2 = relu(bn2(c2)) is formed on the way to LDS exactly as conv3 forms its operand
//                        (same table arithmetic, same packed fma / convert / ReLU: bit for bit), pixels are the K axis, so BOTH
//                        operands are k-strided in the [pixel][channel] image: fragments come from ds_read_b64_tr_b16.
//   SAT_OP_GRAM_COV      slabs summed in slab order in f64 (fixed order: bitwise reproducible), cov = G / M - mu mu^T, rounded to
//                        f32 and split exactly into three bf16 terms (hi + mid + lo = the f32 value) -> [3 P][P] bf16, mu f64.
//   SAT_OP_GEMM_BF16_NT  T[3 P][N] f32 = covsplit [3 P][P] x W3 [N][P]^T on the bf16 matrix pipe (the conv's own weight matrix).
//   SAT_OP_BN_FROM_GRAM  var_c = sum_i W[c][i] (T0 + T1 + T2)[i][c], mean_c = sum_i W[c][i] mu[i] in f64 -> (scale, shift) table
//                        [2][N] + running statistics (or the deferred log of a look-ahead instance).
// Grouped programs (sat_op.groups): every buffer is G consecutive copies; a group's slabs depend on (M, P) only, so a batch gets the
// same bits from the grouped and the ungrouped program.
#include "sat_internal.h"
#include <type_traits>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

constexpr int kTile = 128;                  // channels per tile side
constexpr int kStage = 64;                  // pixels per LDS stage
constexpr double kStatScale = SAT_STAT_SCALE;

struct GramArgs {
    const bf16_t* A;          // [G][M][P] raw conv2 output
    long gs_a;
    long in_bytes;            // bytes of one group's tensor
    float* out;               // [G][n_slabs][slab_stride]: pairs x 128 x 128 partial tiles, then P column sums
    long gs_out, slab_stride;
    int M, P, R, n_slabs, nb, pairs;
    // BatchNorm + ReLU of the operand (bn2): precomputed table or derived from conv2's integer sums, as in the conv kernels
    const float* in_scale;
    const float* in_shift;
    const long long* in_acc;
    long gs_in_acc;
    const float* in_gamma;
    const float* in_beta;
    double in_inv;
    float in_eps;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {     // [rows][128 x bf16] image, 256-byte rows (cdna_hip_programming.md T10 (b))
    return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

__global__ __launch_bounds__(256, 2) void gram_kernel(const GramArgs p_) {
    GramArgs p = p_;
    {
        const long g = blockIdx.y;
        p.A += g * p_.gs_a;
        p.out += g * p_.gs_out;
        if (p.in_acc) p.in_acc += g * p_.gs_in_acc;
    }
    constexpr int BUF = kStage * 256;                            // one stage of one 128-channel block
    __shared__ __attribute__((aligned(16))) char smem[4 * BUF + 2 * 2 * kTile * 4];
    float* in_tab = (float*)(smem + 4 * BUF);                    // [2 blocks][scale 128 | shift 128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slab = blockIdx.x / p.pairs, pair = blockIdx.x - slab * p.pairs;
    int ti = 0, tj = 0;
    {
        int k = pair;
        for (ti = 0; ti < p.nb; ++ti) {
            const int n = p.nb - ti;
            if (k < n) { tj = ti + k; break; }
            k -= n;
        }
    }
    const bool diag = ti == tj;
    const int row_begin = slab * p.R;
    const int row_end = min(row_begin + p.R, p.M);
    const int nst = (row_end - row_begin + kStage - 1) / kStage;

    // ---- (scale, shift) of the operand's BatchNorm for the 128 channels of block ti (and tj): the conv kernels' arithmetic ----
    for (int c = tid; c < (diag ? kTile : 2 * kTile); c += 256) {
        const int blk = c >> 7, ch = (blk ? tj : ti) * kTile + (c & 127);
        float sc, sh;
        if (p.in_acc) {
            const long long s1 = p.in_acc[ch], s2 = p.in_acc[p.P + ch];
            const double mean = (double)s1 * p.in_inv;
            double var = (double)s2 * p.in_inv - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = 1.0f / sqrtf((float)var + p.in_eps);
            sc = p.in_gamma[ch] * invstd;
            sh = p.in_beta[ch] - (float)mean * sc;
        } else {
            sc = p.in_scale[ch];
            sh = p.in_shift[ch];
        }
        in_tab[blk * 256 + (c & 127)] = sc;
        in_tab[blk * 256 + 128 + (c & 127)] = sh;
    }
    __syncthreads();

    // ---- this thread's share of a stage: 16-byte chunk lc (channels lc*8 .. +7 of the block) of rows r0 + 16 j ----
    const int lc = tid & 15, r0 = tid >> 4;
    const int sw = ((r0 & 3) << 2) | ((r0 >> 2) & 3);
    const int st_off = 256 * r0 + 16 * (lc ^ sw);                // + j * 4096
    f32x4 sA0 = *(const f32x4*)(in_tab + lc * 8), sA1 = *(const f32x4*)(in_tab + lc * 8 + 4);
    f32x4 tA0 = *(const f32x4*)(in_tab + 128 + lc * 8), tA1 = *(const f32x4*)(in_tab + 128 + lc * 8 + 4);
    f32x4 sB0 = sA0, sB1 = sA1, tB0 = tA0, tB1 = tA1;
    if (!diag) {
        sB0 = *(const f32x4*)(in_tab + 256 + lc * 8); sB1 = *(const f32x4*)(in_tab + 256 + lc * 8 + 4);
        tB0 = *(const f32x4*)(in_tab + 384 + lc * 8); tB1 = *(const f32x4*)(in_tab + 384 + lc * 8 + 4);
    }
    const __amdgpu_buffer_rsrc_t asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)p.in_bytes, 0x00020000);
    const int colA = (ti * kTile + lc * 8) * 2, colB = (tj * kTile + lc * 8) * 2;
    const int rowbytes = p.P * 2;

    auto load_stage = [&](int s, u32x4 (&da)[4], u32x4 (&db)[4]) -> unsigned {
        unsigned ok = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = row_begin + s * kStage + r0 + 16 * j;
            const bool in = row < row_end;
            const int va = in ? row * rowbytes + colA : 0x7ffffff0;   // past the buffer: reads zeros
            const int vb = in ? row * rowbytes + colB : 0x7ffffff0;
            da[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, va, 0, 0));
            if (!diag) db[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, vb, 0, 0));
            if (in) ok |= 1u << j;
        }
        return ok;
    };
    auto xform = [&](u32x4 w, const f32x4& s0, const f32x4& s1, const f32x4& t0, const f32x4& t1, bool live) {
        // relu(x * scale + shift) per channel: the conv kernels' operand transform, bit for bit (packed fma, ONE packed convert
        // = round to nearest even, ReLU on the bf16 pair as int16); rows outside the slab are exact zeros
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x2 sc2, sh2, f;
            sc2[0] = q < 2 ? s0[2 * q] : s1[2 * q - 4]; sc2[1] = q < 2 ? s0[2 * q + 1] : s1[2 * q - 3];
            sh2[0] = q < 2 ? t0[2 * q] : t1[2 * q - 4]; sh2[1] = q < 2 ? t0[2 * q + 1] : t1[2 * q - 3];
            f[0] = __uint_as_float(w[q] << 16);
            f[1] = __uint_as_float(w[q] & 0xffff0000u);
            f = __builtin_elementwise_fma(f, sc2, sh2);
            const s16x2 pk = __builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2));
            const s16x2 zero2 = {0, 0};
            w[q] = live ? __builtin_bit_cast(unsigned int, __builtin_elementwise_max(pk, zero2)) : 0u;
        }
        return w;
    };
    auto store_stage = [&](int buf, u32x4 (&da)[4], u32x4 (&db)[4], unsigned ok) {
        char* a0 = smem + buf * BUF + st_off;
        char* b0 = smem + (2 + buf) * BUF + st_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *(u32x4*)(a0 + j * 4096) = xform(da[j], sA0, sA1, tA0, tA1, (ok >> j) & 1u);
            if (!diag) *(u32x4*)(b0 + j * 4096) = xform(db[j], sB0, sB1, tB0, tB1, (ok >> j) & 1u);
        }
    };

    // ---- fragment addresses (ds_read_b64_tr_b16): lane l of a 16-lane group supplies row q = (l&15)>>2, columns 4p..4p+3 (p = l&3)
    //      of a 4-pixel x 16-channel block and receives channel l&15 of the four pixels.  32x32x16 operand: lanes 0-15 / 16-31 =
    //      channels 0-15 / 16-31 of the 32-block, lane half h = k 8h..8h+7: read e (0/1) = pixels 8h + 4e .. +3 ----
    const int g16 = lane >> 4, sub = g16 & 1, h = g16 >> 1, q = (lane & 15) >> 2, pp = lane & 3;
    int fa[4][2], fb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 8 * h + 4 * e + q;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) fa[cb][e] = lds_off(row, cb * 4 + 2 * sub + (pp >> 1)) + 8 * (pp & 1);
        fb[e] = lds_off(row, wave * 4 + 2 * sub + (pp >> 1)) + 8 * (pp & 1);
    }
    auto frag = [&](const char* base, int off0, int off1) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + off0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + off1));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x16 acc[4], acc_s;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_s[e] = 0.0f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

    // two stages in flight in registers (a stage = 32 KB per workgroup; with one in flight the loop is bound by the load latency):
    // stage s sits in set s & 1 and is written to LDS buffer s & 1 one barrier before it is read
    u32x4 ra[2][4], rb[2][4];
    unsigned ok[2];
    ok[0] = load_stage(0, ra[0], rb[0]);
    ok[1] = nst > 1 ? load_stage(1, ra[1], rb[1]) : 0u;
    auto stage = [&](int s, auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        store_stage(U, ra[U], rb[U], ok[U]);
        if (s + 2 < nst) ok[U] = load_stage(s + 2, ra[U], rb[U]);    // in flight under this stage's and the next one's MFMAs
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's LDS writes ...
        __builtin_amdgcn_s_barrier();                             // ... and everybody's; everybody is done reading the other buffer
        asm volatile("" ::: "memory");
        const char* sa = smem + U * BUF;
        const char* sb = diag ? sa : smem + (2 + U) * BUF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 b = frag(sb + ks * 4096, fb[0], fb[1]);
            bf16x8 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag(sa + ks * 4096, fa[i][0], fa[i][1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc[i], 0, 0, 0);
            if (diag) acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, b, acc_s, 0, 0, 0);       // column sums: every row of D
        }
    };
    for (int s = 0; s < nst; s += 2) {
        stage(s, std::integral_constant<int, 0>{});
        if (s + 1 < nst) stage(s + 1, std::integral_constant<int, 1>{});
    }
    // ---- the partial tile: element (row = 32 i + (e&3) + 8 (e>>2) + 4 h2, col = 32 wave + r) ----
    const int r = lane & 31, h2 = lane >> 5;
    float* tile = p.out + (long)slab * p.slab_stride + (long)pair * (kTile * kTile);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) tile[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * h2) * kTile + 32 * wave + r] = acc[i][e];
    if (diag && h2 == 0) p.out[(long)slab * p.slab_stride + (long)p.pairs * (kTile * kTile) + tj * kTile + 32 * wave + r] = acc_s[0];
}

// ---- slabs -> covariance, split into three bf16 terms; mu ----
struct CovArgs {
    const float* slabs;       // [G][n_slabs][slab_stride]
    long gs_slabs, slab_stride;
    bf16_t* cov3;             // [G][3][P][P]
    double* mu;               // [G][P]
    int M, P, n_slabs, nb, pairs;
};

__device__ __forceinline__ int pair_index(int bi, int bj, int nb) {     // bi <= bj, row-major over the upper triangle
    return bi * nb - bi * (bi - 1) / 2 + (bj - bi);
}

__global__ __launch_bounds__(256) void gram_cov_kernel(const CovArgs p) {
    extern __shared__ double mu_s[];                              // [P], then [256] scratch
    double* part_s = mu_s + p.P;
    const int g = blockIdx.y;
    const float* slabs = p.slabs + (long)g * p.gs_slabs;
    const long sums_off = (long)p.pairs * (kTile * kTile);
    const double invM = 1.0 / (double)p.M;
    // mu: thread (channel c, part q) sums the slabs k = q, q + parts, ... in that order, the parts are then joined in order: a fixed
    // order whatever the timing.  Eight loads in flight per thread.
    for (int c0 = 0; c0 < p.P; c0 += 256) {
        const int width = min(256, p.P - c0), parts = 256 / width;
        const int c = c0 + threadIdx.x % width, q = threadIdx.x / width;
        double sacc = 0.0;
        if (q < parts) {
            for (int k = q; k < p.n_slabs; k += 8 * parts) {
                // (every load unconditional -- a slab index past the end re-reads the last slab and is dropped by a select on the
                // VALUE: a select on the load would make the compiler branch around each one and wait for it alone)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = slabs[(long)min(k + u * parts, p.n_slabs - 1) * p.slab_stride + sums_off + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) sacc += (k + u * parts < p.n_slabs) ? (double)v[u] : 0.0;
            }
        }
        part_s[threadIdx.x] = sacc;
        __syncthreads();
        if (q == 0) {
            double t = 0.0;
            for (int k = 0; k < parts; ++k) t += part_s[k * width + threadIdx.x];
            mu_s[c] = t * invM;
            if (blockIdx.x == 0) p.mu[(long)g * p.P + c] = t * invM;
        }
        __syncthreads();
    }
    const long n4 = (long)p.P * p.P / 4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx / (p.P / 4)), j4 = (int)(idx - (long)i * (p.P / 4)) * 4;
        const int bi = i >> 7, bj = j4 >> 7;
        double gs[4] = {0.0, 0.0, 0.0, 0.0};
        if (bi <= bj) {
            const long off = (long)pair_index(bi, bj, p.nb) * (kTile * kTile) + (long)(i & 127) * kTile + (j4 & 127);
            for (int k = 0; k < p.n_slabs; k += 8) {              // slab order, eight 16-byte loads in flight
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(slabs + (long)min(k + u, p.n_slabs - 1) * p.slab_stride + off);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double live = (k + u < p.n_slabs) ? 1.0 : 0.0;
                    gs[0] += live * (double)v[u][0]; gs[1] += live * (double)v[u][1]; gs[2] += live * (double)v[u][2]; gs[3] += live * (double)v[u][3];
                }
            }
        } else {                                                  // below the diagonal: the transposed tile (G is symmetric)
            const long off = (long)pair_index(bj, bi, p.nb) * (kTile * kTile) + (long)(j4 & 127) * kTile + (i & 127);
            for (int k = 0; k < p.n_slabs; k += 4) {
                float v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* t = slabs + (long)min(k + u, p.n_slabs - 1) * p.slab_stride + off;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = t[e * kTile];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double live = (k + u < p.n_slabs) ? 1.0 : 0.0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gs[e] += live * (double)v[u][e];
                }
            }
        }
        bf16x4 hi, mid, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float c = (float)(gs[e] * invM - mu_s[i] * mu_s[j4 + e]);
            const bf16_t a = (bf16_t)c;                           // hi + mid + lo == c exactly (3 x 8 significant bits)
            const float r1 = c - (float)a;
            const bf16_t b = (bf16_t)r1;
            const float r2 = r1 - (float)b;
            hi[e] = a; mid[e] = b; lo[e] = (bf16_t)r2;
        }
        bf16_t* dst = p.cov3 + (long)g * 3 * p.P * p.P + (long)i * p.P + j4;
        *(bf16x4*)dst = hi;
        *(bf16x4*)(dst + (long)p.P * p.P) = mid;
        *(bf16x4*)(dst + 2L * p.P * p.P) = lo;
    }
}

// ---- quadratic form + table ----
struct FinArgs {
    const float* T;           // [G][3][P][N]
    const bf16_t* W;          // [N][P]
    const double* mu;         // [G][P]
    const float* gamma;
    const float* beta;
    float* running_mean;      // [G][2][N] log layout when grouped (mean row, var row), or the model's buffers
    float* running_var;
    float* table;             // [G][2][N]: scale row, shift row
    long gs_run;
    int M, P, N;
    float momentum, eps;
};

__global__ __launch_bounds__(256) void bn_from_gram_kernel(const FinArgs p) {
    // thread (channel cl of 16, range q of 16): i in [q P/16, (q+1) P/16) -- all of its loads are independent and issued together
    // (W: 16-byte pieces of its own weight row; T: for a fixed i the 16 channels of the workgroup are 64 contiguous bytes)
    __shared__ double red[2][16][16];
    const int g = blockIdx.y, c0 = blockIdx.x * 16;
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4, c = c0 + cl;
    const int len = p.P / 16;                                     // 8, 16, 24 or 32
    const int i0 = q * len;
    const float* T0 = p.T + (long)g * 3 * p.P * p.N + c;
    const double* mu = p.mu + (long)g * p.P;
    const bf16_t* wrow = p.W + (long)c * p.P + i0;
    double v = 0.0, m = 0.0;
    for (int ib = 0; ib < len; ib += 8) {
        const bf16x8 w8 = *(const bf16x8*)(wrow + ib);
        float t[3][8];
        double mv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long i = i0 + ib + e;
            t[0][e] = T0[i * p.N];
            t[1][e] = T0[(p.P + i) * p.N];
            t[2][e] = T0[(2L * p.P + i) * p.N];
            mv[e] = mu[i];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double w = (double)(float)w8[e];
            v += w * (((double)t[0][e] + (double)t[1][e]) + (double)t[2][e]);
            m += w * mv[e];
        }
    }
    red[0][q][cl] = v;
    red[1][q][cl] = m;
    __syncthreads();
    if (q == 0) {
        double var = 0.0, mean = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {                            // fixed order over the ranges
            var += red[0][k][cl];
            mean += red[1][k][cl];
        }
        if (var < 0.0) var = 0.0;
        const float invstd = 1.0f / sqrtf((float)var + p.eps);
        const float sc = p.gamma[c] * invstd;
        float* tab = p.table + (long)g * 2 * p.N;
        tab[c] = sc;
        tab[p.N + c] = p.beta[c] - (float)mean * sc;
        if (p.running_mean) {
            float* rm = p.running_mean + (long)g * p.gs_run;
            float* rv = p.running_var + (long)g * p.gs_run;
            const double cnt = (double)p.M;
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            // the batch statistic enters as an f32 value: a deferred update (sat_bn_running_apply) is then bit-identical
            rm[c] = (float)((1.0 - p.momentum) * rm[c] + p.momentum * (double)(float)mean);
            rv[c] = (float)((1.0 - p.momentum) * rv[c] + p.momentum * (double)(float)unbiased);
        }
    }
}

int op_groups(const sat_op* op) { return op->groups > 1 ? op->groups : 1; }

}  // namespace

// rows per slab of the Gram launch: a function of (M, P) ONLY (never of the group count), so that a batch's slabs -- the order its
// statistics are summed in -- are the same in the grouped and the ungrouped program.  ~128 workgroups per group.
extern "C" int sat_gram_rows_per_slab(int64_t M, int P) {
    if (M < 1 || P < kTile || (P % kTile)) return 0;
    const int nb = P / kTile, pairs = nb * (nb + 1) / 2;
    int n_slabs = 128 / pairs;
    if (n_slabs < 1) n_slabs = 1;
    long R = (M + (long)n_slabs * kStage - 1) / ((long)n_slabs * kStage) * kStage;
    if (R < kStage) R = kStage;
    return (int)R;
}

extern "C" int sat_gram_slabs(int64_t M, int P) {
    const int R = sat_gram_rows_per_slab(M, P);
    return R > 0 ? (int)((M + R - 1) / R) : 0;
}

// floats of ONE group's slab buffer (SAT_OP_GRAM `out`)
extern "C" int64_t sat_gram_slab_floats(int64_t M, int P) {
    const int n = sat_gram_slabs(M, P);
    if (n <= 0) return 0;
    const int nb = P / kTile, pairs = nb * (nb + 1) / 2;
    return (int64_t)n * ((int64_t)pairs * kTile * kTile + P);
}

// SAT_OP_GRAM: in0 = raw conv2 output [G][M = N*Hout*Wout][P = Cout] bf16; its BatchNorm + ReLU (bn2) as a conv's fused input
// BatchNorm is given: scale0 / shift0, or stat_acc1 / gamma1 / beta1 / count / eps (the parity half is read, nothing is cleared or
// updated here -- conv3 does that); out = slabs f32 [G][sat_gram_slab_floats]
int sat_gram_launch(const sat_op* op, int parity, hipStream_t s) {
    if (!op->in0 || !op->out || op->dtype != SAT_BF16) return SAT_ERR_ARG;
    const long M = (long)op->N * op->Hout * op->Wout;
    const int P = op->Cout;
    if (P < kTile || (P % kTile) || P > 512 || M < 1 || M * P * 2 >= 0x7ffffff0L) return SAT_ERR_UNSUPPORTED;
    GramArgs a = {};
    a.A = (const bf16_t*)op->in0;
    a.gs_a = M * P;
    a.in_bytes = M * P * 2;
    a.out = (float*)op->out;
    a.M = (int)M; a.P = P;
    a.R = sat_gram_rows_per_slab(M, P);
    a.n_slabs = sat_gram_slabs(M, P);
    a.nb = P / kTile; a.pairs = a.nb * (a.nb + 1) / 2;
    a.slab_stride = (long)a.pairs * kTile * kTile + P;
    a.gs_out = (long)a.n_slabs * a.slab_stride;
    if (op->stat_acc1) {
        if (!op->gamma1 || !op->beta1 || op->count < 1) return SAT_ERR_ARG;
        a.in_acc = (const long long*)op->stat_acc1 + (long)parity * 2 * P;
        a.gs_in_acc = 4L * P;
        a.in_gamma = op->gamma1; a.in_beta = op->beta1;
        a.in_inv = 1.0 / (kStatScale * (double)op->count);
        a.in_eps = op->eps;
    } else if (op->scale0 && op->shift0) {
        if (op_groups(op) > 1) return SAT_ERR_UNSUPPORTED;
        a.in_scale = op->scale0; a.in_shift = op->shift0;
    } else {
        return SAT_ERR_ARG;
    }
    hipLaunchKernelGGL(gram_kernel, dim3(a.n_slabs * a.pairs, op_groups(op)), dim3(256), 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// SAT_OP_GRAM_COV: in0 = the slabs of SAT_OP_GRAM (same N/Hout/Wout/Cout); out = cov3 bf16 [G][3][P][P]; scale_out = mu, f64 [G][P]
int sat_gram_cov_launch(const sat_op* op, hipStream_t s) {
    if (!op->in0 || !op->out || !op->scale_out) return SAT_ERR_ARG;
    const long M = (long)op->N * op->Hout * op->Wout;
    const int P = op->Cout;
    if (P < kTile || (P % kTile) || P > 512) return SAT_ERR_UNSUPPORTED;
    CovArgs a = {};
    a.slabs = (const float*)op->in0;
    a.cov3 = (bf16_t*)op->out;
    a.mu = (double*)op->scale_out;
    a.M = (int)M; a.P = P;
    a.n_slabs = sat_gram_slabs(M, P);
    a.nb = P / kTile; a.pairs = a.nb * (a.nb + 1) / 2;
    a.slab_stride = (long)a.pairs * kTile * kTile + P;
    a.gs_slabs = (long)a.n_slabs * a.slab_stride;
    const int grid = (int)(((long)P * P / 4 + 255) / 256);
    hipLaunchKernelGGL(gram_cov_kernel, dim3(grid < 256 ? grid : 256, op_groups(op)), dim3(256), (size_t)(P + 256) * sizeof(double), s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}

// SAT_OP_GEMM_BF16_NT: out f32 [M][Cout] = in0 bf16 [M = N*Hout*Wout][K = Cin] x w bf16 [Cout][K]^T (sat_gemm_bf16_nt)
int sat_gemm_bf16_op_launch(const sat_op* op, hipStream_t s) {
    if (!op->in0 || !op->w || !op->out) return SAT_ERR_ARG;
    const long M = (long)op->N * op->Hout * op->Wout;
    return sat_gemm_bf16_nt(op->in0, op->Cin, op->w, op->Cin, (float*)op->out, op->Cout, nullptr, (int)M, op->Cout, op->Cin, 1, 0,
                            (sat_stream_t)s);
}

// SAT_OP_BN_FROM_GRAM: in0 = T f32 [G][3 P][Cout] (SAT_OP_GEMM_BF16_NT of cov3 with conv3's weights), w = those weights bf16 [Cout][P],
// in1 = mu f64 [G][P], gamma / beta / running_mean / running_var / momentum / eps / count (= M) of bn3, Cin = P;
// scale_out = table f32 [G][2][Cout] (scale row, shift row): conv3's scale1 = table, shift1 = table + Cout
int sat_bn_from_gram_launch(const sat_op* op, hipStream_t s) {
    if (!op->in0 || !op->in1 || !op->w || !op->gamma || !op->beta || !op->scale_out || op->count < 1) return SAT_ERR_ARG;
    const int P = op->Cin, N = op->Cout;
    if (P < kTile || (P % kTile) || P > 512 || (N % 64)) return SAT_ERR_UNSUPPORTED;
    if ((op->running_mean != nullptr) != (op->running_var != nullptr)) return SAT_ERR_ARG;
    const int groups = op_groups(op);
    if (groups > 1 && op->running_mean && op->running_var != op->running_mean + N) return SAT_ERR_ARG;     // the deferred log layout
    FinArgs a = {};
    a.T = (const float*)op->in0; a.W = (const bf16_t*)op->w; a.mu = (const double*)op->in1;
    a.gamma = op->gamma; a.beta = op->beta;
    a.running_mean = op->running_mean; a.running_var = op->running_var;
    a.gs_run = 2L * N;
    a.table = op->scale_out;
    a.M = (int)op->count; a.P = P; a.N = N;
    a.momentum = op->momentum; a.eps = op->eps;
    hipLaunchKernelGGL(bn_from_gram_kernel, dim3(N / 16, groups), dim3(256), 0, s, a);
    SAT_LAUNCH_CHECK();
    return SAT_OK;
}
