/*
 * sat_hip.h -- C ABI of libsat_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the Show-and-Tell
 * training hot path of incredible-vision/show-and-tell.
 *
 * The reference has NO native code and NO FFI: its hot path is a chain of torch operator
 * call sites.  Each entry point below replaces the call site cited next to it, so a maintainer can bind
 * it from Python with ctypes.  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked [host];
 *   - the caller owns every buffer (inputs, outputs, tapes, workspaces): nothing here allocates or frees;
 *   - every launch goes to `stream` (a hipStream_t); no host synchronisation, safe to capture in a hipGraph;
 *   - return 0 on success, SAT_ERR_* (>= 1001) for argument errors, otherwise a hipError_t value;
 *   - packed sequences are TIME-MAJOR: rows of step t are [prefix[t], prefix[t]+batch_sizes[t]),
 *     batch_sizes non-increasing (pack_padded_sequence with lengths sorted descending, models.py:51).
 */
#ifndef SAT_HIP_H
#define SAT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAT_ABI_VERSION 18

#define SAT_OK 0
#define SAT_ERR_ARG 1001
#define SAT_ERR_WORKSPACE 1002
#define SAT_ERR_UNSUPPORTED 1003

#define SAT_F32 0
#define SAT_BF16 1

typedef void* sat_stream_t; /* hipStream_t */

int sat_version(void);
const char* sat_error_string(int code);

/* ------------------------------------------------------------------------------------------------
 * Generic dense f32 GEMM on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32): C[M,N] = op(A) op(B) + bias + bias2
 *   amode 0: A[m*lda + k]          amode 2: A[k*lda + m]
 *   bmode 0: B[n*ldb + k]  ("NT")  bmode 1: B[k*ldb + n]
 * Replaces the cuBLAS/MKL GEMMs under nn.Linear / nn.LSTM (models.py:52-53) and their backward.
 * Requires K%4==0, lda/ldb%4==0 (+ M%4 for amode 2, N%4 for bmode 1), 16-byte aligned bases.
 */
int sat_gemm_f32(int amode, int bmode, const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc, const float* bias, const float* bias2,
                 int M, int N, int K, sat_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder op program (models.py:25-29 `self.resnet(images)`): the host describes the frozen conv stack
 * as an array of sat_op and one call launches all of it.  Activations are NHWC, dtype f32 or bf16.
 */
enum {
    SAT_OP_IMAGE_PREP = 1, /* NCHW f32 image -> zero-padded NHWC4 (in0 -> out); H/W = Hin/Win, pad = border; Cout = 8 (bf16):
                            * NHWC8 instead, one 16-byte chunk per pixel (3x3 stems: VGG conv1_1, model2.py:15) */
    SAT_OP_CONV = 2,       /* implicit-GEMM conv: in0 (NHWC) * w [Cout][KH*KW*Cin] -> out [M][Cout] (+ stat_partial) */
    SAT_OP_BN_FINALIZE = 3,/* partials -> scale/shift (+ running stats update); with stat_acc set ("acc mode"):
                            * partials -> the fixed-point integer accumulators stat_acc (parity half), consumers derive */
    SAT_OP_BN_RELU = 4,    /* out = relu(in0*scale0 + shift0) */
    SAT_OP_BN_ADD_RELU = 5,/* out = relu(in0*scale0+shift0 + (in1*scale1+shift1 | in1)) */
    SAT_OP_BN_RELU_MAXPOOL = 6, /* out = maxpool3x3/2(relu(in0*scale0+shift0)) */
    SAT_OP_AVGPOOL = 7,    /* out f32 [N][C] = mean over Hin*Win of in0 */
    SAT_OP_MAXPOOL3S2 = 10,/* out = maxpool 3x3 / stride 2 / no padding of in0 (NHWC, Cout channels; ldc = output row pitch) */
    SAT_OP_AVGPOOL3 = 11,  /* out = avgpool 3x3 / stride 1 / pad 1, count_include_pad (divide by 9) of in0 (NHWC, Cout channels) */
    SAT_OP_MAXPOOL2 = 9,   /* out = maxpool 2x2 / stride 2 of in0 (NHWC; Hin, Win even; Cout channels): VGG16, model2.py:15-16 */
    SAT_OP_BN_EVAL_BATCH = 8,/* eval mode: in0 = DEVICE array of `count` sat_bn_eval_item; every item's (scale, shift)
                              * from its running statistics in ONE launch (replaces one BN_FINALIZE per layer); eps */
    /* Train-mode bn3 of a bottleneck WITHOUT a pass over conv3's output (models.py:27, BatchNorm2d with batch statistics): mean and
     * variance of c3 = a2 W3^T are a linear / quadratic form of conv3's INPUT a2 = relu(bn2(c2)): mean_c = w_c . mu,
     * var_c = w_c^T (G / M - mu mu^T) w_c with G = a2^T a2 (P x P, P = planes).  Four ops in front of conv3, which then runs with
     * the inference epilogue (scale1 / shift1 = the table, in1 = the residual, flags bit 0 = ReLU; SAT_CONV_GROUP_TABLE when grouped):
     * the raw conv3 tensor and the SAT_OP_BN_ADD_RELU launch (42 % of the stack's memory traffic in round 4) never exist.
     * bf16 only; P in {128, 256, 384, 512}.  All honour `groups`. */
    SAT_OP_GRAM = 12,        /* in0 = raw conv2 output [G][M = N*Hout*Wout][P = Cout]; its BatchNorm + ReLU as on a conv's fused input
                              * (stat_acc1 / gamma1 / beta1 / count / eps, or scale0 / shift0): read only, nothing cleared or updated;
                              * out = f32 slabs [G][sat_gram_slab_floats(M, P)]: per row slab the partial 128 x 128 tiles of G (upper
                              * triangle of tile pairs) and the P column sums, f32 MFMA accumulate */
    SAT_OP_GRAM_COV = 13,    /* in0 = those slabs (same N / Hout / Wout / Cout); slabs summed in slab order in f64, cov = G / M - mu mu^T
                              * rounded to f32 and split exactly into three bf16 terms: out = bf16 [G][3][P][P]; scale_out = mu as
                              * DOUBLE [G][P] */
    SAT_OP_GEMM_BF16_NT = 14,/* out f32 [M = N*Hout*Wout][Cout] = in0 bf16 [M][K = Cin] x w bf16 [Cout][K]^T (sat_gemm_bf16_nt); here:
                              * T [G * 3 P][Cout] = cov3 x conv3's weight matrix */
    SAT_OP_BN_FROM_GRAM = 15 /* in0 = T f32 [G][3 P][Cout], w = conv3's weights bf16 [Cout][P = Cin], in1 = mu f64 [G][P]; gamma / beta /
                              * running_mean / running_var / momentum / eps / count (= M) of bn3: var_c = sum_i w[c][i] (T0+T1+T2)[i][c],
                              * mean_c = w_c . mu in f64 -> scale_out = f32 table [G][2][Cout] (scale row, shift row); running statistics
                              * updated in place (grouped: the deferred log [G][2][Cout]) */
};

/* one BatchNorm of an eval-mode stack (all pointers device memory, C floats each) */
typedef struct sat_bn_eval_item {
    const float* gamma;
    const float* beta;
    const float* running_mean;
    const float* running_var;
    float* scale_out;
    float* shift_out;
    int32_t C;
    int32_t reserved;
} sat_bn_eval_item;

/* one BatchNorm of a train-mode stack whose running-statistics update was deferred (sat_bn_running_apply) */
typedef struct sat_bn_running_item {
    float* running_mean;          /* the model's buffers (models.py:14 resnet152 BatchNorm2d), C floats */
    float* running_var;
    const float* batch_mean;      /* this batch's mean / unbiased variance as the program left them (f32) */
    const float* batch_var;
    int32_t C;
    int32_t reserved;
} sat_bn_running_item;

typedef struct sat_op {
    int32_t kind;
    int32_t dtype;            /* SAT_F32 / SAT_BF16 : element type of in0/in1/out/w */
    const void* in0;
    const void* in1;
    void* out;
    const void* w;
    const float* scale0;
    const float* shift0;
    const float* scale1;      /* NULL => in1 used as is */
    const float* shift1;
    float* stat_partial;      /* [tiles_m][2][Cout] f32 : per-tile column sum / sum of squares */
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* scale_out;
    float* shift_out;
    int32_t N, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad;
    int32_t training;         /* BN_FINALIZE: 1 = batch statistics (+running update), 0 = running statistics */
    int32_t tiles_m;          /* number of partial rows in stat_partial (= ceil(M/128)) */
    int64_t sN, sH, sW;       /* element strides of in0 for SAT_OP_CONV (lets the stem read a padded NHWC4 image) */
    int64_t count;            /* BN_FINALIZE: elements per channel (N*Hout*Wout) */
    float momentum, eps;
    int32_t variant;          /* SAT_OP_CONV: 0 = built-in heuristic, >0 = kernel variant chosen by sat_conv_autotune; < 0 on entry to
                               * sat_conv_autotune: choose among the variants of statistics signature -variant only */
    int32_t flags;            /* SAT_OP_CONV, bf16, inference: scale1/shift1 [Cout] set => the epilogue writes
                               * out = acc*scale1[n] + shift1[n] (+ in1[m][n], a residual shaped like out) and, with
                               * bit 0 of flags, ReLU -- BatchNorm (eval) + add + ReLU of a bottleneck without a
                               * separate launch (models.py:27 under eval.py:65 `model.eval()`) */
    /* Integer-atomic BatchNorm statistics (bf16, training, few M-tiles): stat_acc = int64 [2 step parities][2][C]
     * fixed-point (2^22) column sums, zero before first use.  On a SAT_OP_CONV the kernel ADDS this tile's sums into
     * parity p's half; on the consuming SAT_OP_BN_RELU / SAT_OP_BN_ADD_RELU (same pointer, plus gamma, beta, running_mean, running_var,
     * count, momentum, eps) every workgroup derives scale/shift from them, workgroup 0 updates the running statistics
     * and clears parity 1-p's half: no SAT_OP_BN_FINALIZE launch.  p is sat_run_ops_parity's argument and must
     * alternate between consecutive runs of the program.  The *1 fields describe in1's BatchNorm (BN_ADD_RELU). */
    void* stat_acc;
    void* stat_acc1;
    const float* gamma1;
    const float* beta1;
    float* running_mean1;
    float* running_var1;
    const void* w_packed;     /* SAT_OP_CONV, bf16, Cin % 64 == 0, Cout % 128 == 0, 3x3 / stride 1 / pad 1 or 1x1 / pad 0: the weights once more in
                               * MFMA fragment order (sat_conv_pack_weights) -- lets the tuner pick conv_pw_kernel (3x3) / conv_aw_kernel
                               * (1x1), which stream them straight into registers; NULL = not provided */
    int32_t reserved1[2];     /* (were stat_shards / stat_shards1: sharded accumulators, removed in ABI 13 -- a measured wash) */
    /* SAT_OP_CONV extras for Inception-style stacks (BASELINE configs[3]): flags bit 1 (SAT_CONV_PADW) = the padding differs per
     * axis: `pad` is the vertical one, pad_w the horizontal one (1x7 / 7x1 / 1x3 / 3x1 kernels); ldc = row pitch of `out` in
     * elements (0 = Cout): a conv / activation may write its channels into a slice of a wider (concatenated) NHWC tensor.
     * SAT_OP_BN_RELU honours ldc the same way for its output. */
    int32_t pad_w;
    /* GROUPED program (bf16, training): groups = G > 1 runs G independent BATCHES in every launch of the op (grid.y = group) --
     * the frozen conv stack of G look-ahead batches as ONE program, half (1/G) the launch boundaries and twice (G x) the
     * workgroups per launch.  N stays the batch of ONE group; every per-batch buffer is G consecutive copies of the ungrouped
     * one: in0 / in1 / out [G][N][H][W][C] (out: G x N*Hout*Wout rows of pitch ldc), stat_partial [G][tiles_m][2][Cout],
     * stat_acc / stat_acc1 [G][2 parities][2][C], running_mean / running_var of a DEFERRED program (sat_bn_running_apply)
     * [G][2][C] (mean row, then variance row: running_var == running_mean + C).  Weights, gamma / beta are shared.  Each group
     * keeps its own batch statistics and is, instruction for instruction, the ungrouped launch on its batch.  Honoured by
     * SAT_OP_CONV, SAT_OP_BN_FINALIZE (acc mode), SAT_OP_BN_RELU, SAT_OP_BN_ADD_RELU, SAT_OP_BN_RELU_MAXPOOL; SAT_OP_IMAGE_PREP
     * and SAT_OP_AVGPOOL are per image: give them N = G * batch.  0 / 1 = ungrouped. */
    int32_t groups;
    int64_t ldc;
    /* SAT_OP_CONV with flags bit 3 (SAT_CONV_IN_RESIDUAL; bf16, 1x1 / stride 1 on a dense NHWC tensor, w_packed, Cin a multiple of 128
     * in [256, 2048], Cout of 128): this conv ALSO finishes the bottleneck in front of it.  Its operand is
     *     y = relu(bn(in0) + in1)        (bn = the input BatchNorm of scale0 / shift0 or stat_acc1 / gamma1 / ...; in1 shaped like in0)
     * built on its way to LDS -- bit for bit what SAT_OP_BN_ADD_RELU writes -- and y is written to out1 (shaped like in0, per group like
     * in0) as well: the normalise + add + ReLU launch between conv3 of one bottleneck and conv1 of the next disappears
     * (models.py:27, train-mode BatchNorm).  out1 may alias in0 when Cout == 128 (or == 256 and the eight-wave variant runs it: the
     * library checks); it must not alias in1. */
    void* out1;
} sat_op;
#define SAT_CONV_PADW 2
#define SAT_CONV_IN_RESIDUAL 8
#define SAT_CONV_GROUP_TABLE 4   /* SAT_OP_CONV flags: scale1 / shift1 are PER GROUP, [G][2][Cout] apart (the table SAT_OP_BN_FROM_GRAM writes), and
                                  * in1 (the residual) is per group like `out`: lets a grouped train-mode program run conv3 with the
                                  * inference epilogue */
/* slab geometry of SAT_OP_GRAM: rows per slab and slab count are functions of (M, P) only -- never of `groups` -- so a batch's
 * statistics are summed in the same order in the grouped and the ungrouped program; floats of ONE group's slab buffer */
int sat_gram_rows_per_slab(int64_t M, int P);
int sat_gram_slabs(int64_t M, int P);
int64_t sat_gram_slab_floats(int64_t M, int P);
int sat_run_ops(const sat_op* ops /*[host]*/, int n_ops, sat_stream_t stream);
/* same, with the step parity (0/1) that selects the half of every stat_acc buffer in use */
/* Deferred running statistics.  A program built with its sat_op running_mean / running_var pointers aimed at PRIVATE, zeroed
 * buffers and momentum 1 leaves each layer's batch (mean, unbiased variance) there and touches no model state, so two batches'
 * frozen conv stacks may be in flight at once (TrainStep.prefetch_encoder).  This applies one finished batch to the model:
 * running = (1-momentum)*running + momentum*batch for every item, one launch, bit-identical to the in-kernel update.  The
 * caller orders the calls like the batches (nn.BatchNorm2d momentum update, train.py:128 `encoder.train()`). */
int sat_bn_running_apply(const sat_bn_running_item* items /*[device]*/, int n_items, float momentum, sat_stream_t stream);
/* counters[i] += value for n int64 counters [device]: `num_batches_tracked += 1` of every BatchNorm of a stack in one launch
 * (nn.BatchNorm2d / BatchNorm1d forward in training mode, models.py:27-28 under train.py:128 `encoder.train()`). */
int sat_counter_add(int64_t* counters /*[device]*/, int n, int64_t value, sat_stream_t stream);
int sat_run_ops_parity(const sat_op* ops /*[host]*/, int n_ops, int parity, sat_stream_t stream);
/* The same program as ONE hipGraph launch (the reference's `self.resnet(images)` issues ~1500 eager kernels per
 * forward, models.py:27; here the host enqueues one graph).  sat_graph_create records ops[0..n) for the given step
 * parity on an internal capture stream (nothing executes) and instantiates it; all pointers and shapes in ops[] are
 * frozen into the graph.  sat_graph_launch replays it on `stream`; sat_graph_destroy frees it.  No host sync. */
typedef struct sat_graph sat_graph;
int sat_graph_create(const sat_op* ops /*[host]*/, int n_ops, int parity, sat_graph** graph_out);
int sat_graph_launch(sat_graph* graph, sat_stream_t stream);
int sat_graph_destroy(sat_graph* graph);
/* conv -> train-mode BatchNorm -> ReLU of one bottleneck stage (models.py:27) as one call over three op records:
 * conv (statistics in its epilogue), finalize (scale/shift + running statistics), bnrelu (in0 must be conv->out) */
int sat_conv_bn_relu_fwd(const sat_op* conv, const sat_op* finalize, const sat_op* bnrelu, sat_stream_t stream);
/* rows of SAT_OP_CONV partials the conv kernel writes for M output pixels */
int sat_conv_tiles_m(int64_t M);
/* Build-time tuner (NOT for the hot path: it times launches with HIP events and synchronises): for every bf16
 * SAT_OP_CONV in ops[] run each kernel variant `reps` times on the op's own buffers and record the fastest in
 * ops[i].variant.  Results are cached per conv geometry inside the library (mutex-protected).  `scratch`: >= 16384
 * bytes of device memory owned by the caller (a neutral BatchNorm table for the input-fused convs lives there while
 * the tuner runs): like every other entry point, this one allocates no device memory. */
int sat_conv_autotune(sat_op* ops /*[host]*/, int n_ops, int reps, float* scratch, int64_t scratch_bytes,
                      sat_stream_t stream);
/* The same, and in cand[n_ops][topk] [host] the `topk` fastest variants of every op (1-based, fastest first, 0 = no more; all 0
 * for ops the tuner does not handle).  A replayed launch finds its operand warm, a launch inside the program does not, and the two
 * rankings differ by a few microseconds either way: a caller that can time its whole program (sat_run_ops_timed) makes the final
 * choice among the candidates IN the program (resnet.ConvStackProgram._autotune). */
int sat_conv_autotune_topk(sat_op* ops /*[host]*/, int n_ops, int reps, float* scratch, int64_t scratch_bytes,
                           sat_stream_t stream, int topk, int32_t* cand /*[host]*/);
/* What fixes the BITS of the BatchNorm column sums a kernel variant (sat_op.variant, 1-based) leaves: two variants with the
 * same signature give bit-identical statistics (the conv output is bit-identical across the ring variants anyway).  A caller
 * that runs one model through several programs (ungrouped, grouped look-ahead copies) lets its FIRST program tune freely, reads
 * the signatures of the variants it got, and hands them to the tuner for every other program (sat_op.variant = -signature on
 * entry to sat_conv_autotune): every batch then gets, bit for bit, the same statistics whichever program runs it.  -1: no such
 * variant. */
int sat_conv_variant_signature(int variant);
/* The same for the conv OUTPUT alone (inference programs: no statistics): 100000 + the order in which the variant walks the K axis
 * (0: tap major -- ring, expansion and stem kernels; 1: channel-block major -- the two LDS-patch 3x3 kernels).  Outputs are
 * bit-identical within a family; accepted by sat_conv_autotune as a constraint like a statistics signature. */
int sat_conv_variant_family(int variant);
/* Kernel selection WITHOUT timing (the default since ABI 16: the arithmetic of a run must not depend on a stopwatch).  A tuned
 * choice is a variant NUMBER, valid for the build it was measured on: sat_conv_num_variants() stamps a saved table
 * (show-and-tell_amd/tune/gfx950.json).  For a geometry no table names, sat_conv_default_variant gives the variant (1-based) the op
 * runs: the built-in heuristic, or -- want_sig >= 0 (a statistics signature / output family as above) and the heuristic's choice
 * has another one -- the first variant of that signature the op can run; 0 if there is none.  A function of the op's geometry
 * only: every process, rank and box gets the same answer (config.py:15 `random_seed`: same seed, same bits). */
int sat_conv_num_variants(void);
int sat_conv_default_variant(const sat_op* op /*[host]*/, int want_sig);
/* bf16 weights [Cout][taps][Cin] (the kernels' layout; Cout % 32 == 0, Cin % 64 == 0) -> `packed` (same element count) in the
 * MFMA fragment order conv_pw_kernel streams into registers: [Cout/32][Cin/64][taps][4][64 lanes][8].  Once per weight version
 * of a frozen stack (`self.resnet(images)`, models.py:14-15,27). */
int sat_conv_pack_weights(const void* w, void* packed, int Cout, int Cin, int taps, sat_stream_t stream);
/* Diagnostics, NOT the hot path (creates events, synchronises `stream`): run ops[0..n) once, in order, and return in
 * op_us[i] [host] the duration in microseconds of every bf16 SAT_OP_CONV launch taken from its own dispatch
 * timestamps (what rocprofv3 --kernel-trace reports for that launch); 0 for the other ops.  bench.py uses it for the
 * roofline figure of the dominant kernel. */
int sat_run_ops_timed(const sat_op* ops /*[host]*/, int n_ops, int parity, sat_stream_t stream, float* op_us /*[host]*/);

/* ------------------------------------------------------------------------------------------------
 * Encoder head: resnet.fc (Linear 2048->E) + BatchNorm1d(E, momentum=0.01)  (models.py:16-17,27-28)
 */
int sat_fc_bn1d_fwd(const float* pooled /*[B,F]*/, const float* w_fc /*[E,F]*/, const float* b_fc /*[E]*/,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, int training, int B, int F, int E,
                    float* feats /*[B,E]*/, float* xhat /*[B,E]*/, float* rstd /*[E]*/,
                    float* workspace, int64_t ws_bytes, sat_stream_t stream);
int64_t sat_fc_bn1d_ws_bytes(int B, int F, int E);
int sat_fc_bn1d_bwd(const float* dy /*[B,E]*/, const float* pooled, const float* xhat, const float* rstd,
                    const float* gamma, int B, int F, int E,
                    float* dw_fc /*[E,F]*/, float* db_fc, float* dgamma, float* dbeta,
                    float* workspace /* >= B*E floats */, int64_t ws_bytes, sat_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Decoder (models.py:47-54).
 */
/* embed + cat(features) + pack (models.py:49-51): X[N,E], row (t,b): t==0 ? features[b] : embed[captions[b][t-1]] */
int sat_embed_concat_fwd(const float* features /*[B,E]*/, const float* embed /*[V,E]*/,
                         const int64_t* captions /*[B][cap_stride]*/, int64_t cap_stride,
                         const int32_t* prefix /*[T+1] device*/, int T, int N /*=prefix[T]*/, int B, int E, int V,
                         float* X /*[N,E]*/, sat_stream_t stream);
/* backward of the above: d_embed (dense, zeroed here, deterministic order) and d_features */
int sat_embed_concat_bwd(const float* dX /*[N,E]*/, const int64_t* captions, int64_t cap_stride,
                         const int32_t* prefix /*device*/, int T, int N, int B, int E, int V,
                         float* d_embed /*[V,E]*/, float* d_features /*[B,E]*/, sat_stream_t stream);

/* Range check of an id matrix ids[rows][row_stride] (first `cols` columns): status[0] |= 1 when any id is outside
 * [lo, hi).  nn.Embedding (models.py:49) and nn.CrossEntropyLoss (train.py:143) raise on such ids; the kernels below
 * clamp them for memory safety only, the host wrappers launch this check on the caption matrix and raise when the
 * status word (device int32, zero-initialised by the caller) comes back set. */
int sat_validate_ids(const int64_t* ids, int64_t row_stride, int rows, int cols, int64_t lo, int64_t hi,
                     int32_t* status, sat_stream_t stream);

/* targets = pack_padded_sequence(captions[:,1:], lengths-1).data (train.py:134-135); prefix/T/N describe lengths-1 */
int sat_pack_targets(const int64_t* captions /*[B][cap_stride]*/, int64_t cap_stride, const int32_t* prefix /*device*/,
                     int T, int N, int64_t* targets /*[N]*/, sat_stream_t stream);

/* one nn.LSTM layer over a packed batch (models.py:52), zero initial state, gate order i,f,g,o.
 * Tapes for the backward: GA[N,4H] post-activation gates, CS[N,H] cell states, HS[N,H] outputs,
 * HP[N,H] previous hidden state per row.  c_state/h_state: [B,H] scratch. */
int sat_lstm_fwd(const float* X /*[N,In]*/, const float* w_ih /*[4H,In]*/, const float* w_hh /*[4H,H]*/,
                 const float* b_ih, const float* b_hh, const int32_t* batch_sizes /*[T] host*/, int T,
                 int In, int H, float* GA, float* CS, float* HS, float* HP, float* c_state,
                 void* workspace, int64_t ws_bytes, sat_stream_t stream);
/* workspace of the persistent recurrence (hidden-state exchange: [2 parities][groups of 8 rows][8][H] self-tagged 4-byte words --
 * bit 30 of |h| <= 1 carries the hand-off phase -- + an error word; the call zeroes it itself).  With it (and H in
 * {32,64,96,128,256,512}, T <= 64, ceil(B/8) * H/16 <= CUs) all T steps run in ONE launch with W_hh register-resident;
 * workspace NULL / too small, or a shape outside that envelope, falls back to one launch per step (same results). */
int64_t sat_lstm_fwd_ws_bytes(int B, int H);
/* Byte offset of the recurrence's STATUS WORD (uint32) in that workspace, or -1 when the shape has no persistent form.
 * sat_lstm_fwd zeroes it; the persistent launch sets it non-zero when a workgroup gave up waiting for its group (every wait
 * is bounded: co-tenants that keep its workgroups from being resident together).  The tapes and HS of such a call are
 * INVALID: the caller must copy the word out behind the call (hipMemcpyAsync to pinned memory) and treat non-zero as an
 * error -- `show-and-tell_amd.watch.ResidencyWatch` raises RuntimeError and switches the process to per-step launches. */
int64_t sat_lstm_fwd_status_offset(int B, int H);
/* process-wide switch of the persistent recurrence (default on); returns the previous setting.  Off: one launch per step. */
int sat_lstm_persist_enable(int on);
int64_t sat_lstm_bwd_ws_bytes(int B, int H);          /* minimum */
/* size that also lets the batched dW_ih / dX GEMMs run split-K when they would leave most CUs idle (N = packed rows), and the
 * recurrence as ONE persistent launch.  Layout (ABI 16): the granule-exchange region of the persistent recurrence and its status
 * word come FIRST, at offsets that depend on (B, H) only; the N-dependent slabs follow.  So ONE buffer sized with
 * sat_lstm_bwd_ws_bytes_max for the longest batch serves every call (any N <= n_max).  The exchange region is never cleared per
 * call -- granules carry a per-call epoch tag -- and the invariant "no foreign bit pattern in it" is the LIBRARY's: it clears the
 * region the first time it sees the buffer's address (and when its 24-bit epoch counter wraps) and remembers the address; an owner
 * that frees the buffer (or reuses the memory for something else) says so with sat_lstm_ws_release.  The caller zeroes nothing. */
int64_t sat_lstm_bwd_ws_bytes_full(int N, int B, int In, int H);
int64_t sat_lstm_bwd_ws_bytes_max(int n_max, int B, int In, int H);
int sat_lstm_ws_release(void* workspace);
/* With the FULL workspace (and the forward's envelope: H in {32,...,512}, T <= 64, ceil(B/8) * H/16 <= CUs) the backward
 * recurrence runs as ONE persistent launch too (SAT_LSTM_PERSIST_BWD=0: one launch per step).  Its STATUS WORD (uint32) sits at
 * this byte offset of the workspace (a function of B and H only): zeroed by every such call, non-zero when a bounded wait of the
 * persistent launch ran out -- DG and every gradient of that call are then INVALID; read it back behind the call as for
 * sat_lstm_fwd_status_offset. */
int64_t sat_lstm_bwd_status_offset(int N, int B, int In, int H);
int sat_lstm_bwd(const float* dHS /*[N,H]*/, const float* X, const float* w_ih, const float* w_hh,
                 const float* GA, const float* CS, const float* HP,
                 const int32_t* batch_sizes /*[T] host*/, int T, int In, int H,
                 float* DG /*[N,4H] out*/, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                 float* dX /*[N,In] or NULL*/, float* workspace, int64_t ws_bytes, sat_stream_t stream);

/* bf16 THROUGHPUT mode of an LSTM layer's batched GEMMs (BASELINE configs[1] names bf16): sat_lstm_fwd / sat_lstm_bwd with the
 * x-gates product and dW_ih / dW_hh / dX on v_mfma_f32_32x32x16_bf16 (f32 accumulate and outputs) from bf16 copies of the f32
 * operands made in `mixed_ws` (sat_lstm_mixed_ws_bytes, 256-byte aligned); the recurrence, the gate arithmetic, the bias
 * gradients and the master weights stay f32.  A too small mixed_ws silently keeps the exact-f32 GEMMs. */
int64_t sat_lstm_mixed_ws_bytes(int N /*packed rows*/, int In, int H);
int sat_lstm_fwd_bf16(const float* X, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                      const int32_t* batch_sizes /*[T] host*/, int T, int In, int H, float* GA, float* CS, float* HS, float* HP,
                      float* c_state, void* workspace, int64_t ws_bytes, void* mixed_ws, int64_t mixed_bytes, sat_stream_t stream);
int sat_lstm_bwd_bf16(const float* dHS, const float* X, const float* w_ih, const float* w_hh, const float* GA, const float* CS,
                      const float* HP, const int32_t* batch_sizes /*[T] host*/, int T, int In, int H, float* DG, float* dw_ih,
                      float* dw_hh, float* db_ih, float* db_hh, float* dX, float* workspace, int64_t ws_bytes, void* mixed_ws,
                      int64_t mixed_bytes, sat_stream_t stream);

/* vocab projection (models.py:53) */
/* logits rows have stride ldl >= V floats.  For the backward (sat_vocab_ce_bwd) ldl must be a multiple of 4 and the
 * pad columns [V, ldl) must hold zeros (allocate the buffer zero-filled: no kernel here writes the pad). */
int sat_vocab_logits_fwd(const float* Hs /*[N,H]*/, const float* w /*[V,H]*/, const float* b /*[V]*/,
                         int N, int H, int V, float* logits /*[N,ldl]*/, int64_t ldl, sat_stream_t stream);
/* row-wise softmax cross entropy (train.py:53,143): row_loss[n] = lse - logit[target]; loss_out[0] =
 * inv_denom * sum(row_loss) (fixed-order reduction).  write_grad: logits are overwritten IN PLACE with
 * d(loss)/d(logits) = (softmax - onehot) * inv_denom. */
int sat_ce_rows(float* logits /*[N,ldl]*/, int64_t ldl, const int64_t* targets /*[N]*/, int N, int V, float inv_denom,
                int write_grad, float* row_loss /*[N]*/, float* loss_out /*[1]*/, sat_stream_t stream);
/* ---- bf16 THROUGHPUT mode of the vocab projection and its backward (models.py:53, train.py:143-144; BASELINE configs[1]
 * names bf16): the three GEMMs run on v_mfma_f32_32x32x16_bf16 with f32 accumulate from bf16 COPIES of Hs / W / d(logits)
 * made per step inside `workspace`; logits, CE arithmetic, bias gradient, all outputs and the master weights stay f32.
 * d(loss)/d(logits) exists only as bf16, in the workspace, between the two calls.  Needs H % 64 == 0, V % 4 == 0,
 * V <= 12288 (SAT_ERR_UNSUPPORTED otherwise: use sat_vocab_logits_fwd / sat_ce_rows / sat_vocab_ce_bwd).
 * sat_vocab_ce_fwd_bf16: logits[N,ldl] = Hs W^T + b; row_loss[n] = lse - logit[target]; loss_out[0] = inv_denom * sum.
 * sat_vocab_ce_bwd_bf16: dw[V,H] = G^T Hs, db[V] = colsum(G), dHs[N,H] = G W with G = (softmax - onehot) * inv_denom. */
int64_t sat_vocab_bf16_ws_bytes(int N, int H, int V);     /* 0: shape unsupported */
int sat_vocab_ce_fwd_bf16(const float* Hs /*[N,H]*/, const float* w /*[V,H]*/, const float* b /*[V]*/, const int64_t* targets /*[N]*/,
                          int N, int H, int V, float inv_denom, float* logits /*[N,ldl]*/, int64_t ldl, float* row_loss /*[N]*/,
                          float* loss_out /*[1] or NULL*/, void* workspace /*256-byte aligned*/, int64_t ws_bytes, sat_stream_t stream);
int sat_vocab_ce_bwd_bf16(int N, int H, int V, float* dw /*[V,H]*/, float* db /*[V]*/, float* dHs /*[N,H]*/, void* workspace,
                          int64_t ws_bytes, sat_stream_t stream);
/* the kernel under them: C[M,N] f32 = A[M,K] B[N,K]^T (+ bias[N]); A, B bf16 with K contiguous, K % 64 == 0 (zero padded),
 * N % 4 == 0, lda / ldb % 8 == 0.  ksplit > 1: K slice z writes C + z * slab_stride, no bias (sum with sat_sum_slabs_f32). */
int sat_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                     int M, int N, int K, int ksplit, int64_t slab_stride, sat_stream_t stream);
/* out bf16 [C][ldo] = in^T for in f32 [R][ldi]; columns [R, ldo) zero (ldo % 4 == 0) */
int sat_transpose_f32_bf16(const float* in, int64_t ldi, int R, int C, void* out, int64_t ldo, sat_stream_t stream);
/* Vocab projection + cross entropy (models.py:53 + train.py:53,143) as ONE call, exact f32: logits = Hs w^T + b, then
 * row_loss[n] = logsumexp(logits[n]) - logits[n][targets[n]], loss_out[0] = inv_denom * sum(row_loss), and the logits are
 * overwritten IN PLACE with d(loss)/d(logits) = (softmax - onehot) * inv_denom, ready for sat_vocab_ce_bwd (= sat_vocab_logits_fwd
 * followed by sat_ce_rows with write_grad).  ldl >= V; pad columns [V, ldl) are left alone (zero-fill once when ldl > V). */
int sat_vocab_ce_fwd(const float* Hs /*[N,H]*/, const float* w /*[V,H]*/, const float* b /*[V]*/, const int64_t* targets /*[N]*/,
                     int N, int H, int V, float inv_denom, float* logits /*[N,ldl]*/, int64_t ldl,
                     float* row_loss /*[N]*/, float* loss_out /*[1]*/, sat_stream_t stream);
/* backward of the projection given dlogits: dW[V,H], db[V], dHs[N,H] */
int sat_vocab_ce_bwd(const float* dlogits /*[N,ldl]*/, int64_t ldl, const float* Hs, const float* w, int N, int H, int V,
                     float* dw, float* db, float* dHs, float* workspace, int64_t ws_bytes, sat_stream_t stream);
int64_t sat_vocab_ce_bwd_ws_bytes(int N, int H, int V);
/* f32 GEMM on the bf16 matrix pipe by a three-way bf16 split of both operands (hi + mid + lo, six products, hi*hi and the corrections
 * in separate f32 accumulators): the accuracy of an f32 GEMM -- not its bit pattern -- at 6 / 16 of the f32 pipe's cost.  For weights
 * that stay fixed over many calls (the vocab projection of the beam decode loop, models.py:53 / :63): sat_gemm_f32x3_pack splits W
 * [N][K] f32 once into its fragment-ordered copy (sat_gemm_f32x3_packed_bytes(N, K) bytes; 0 = shape not supported: K % 128 == 0,
 * K >= 256), sat_gemm_f32x3 computes C[M][ldc] = A[M][lda] * W^T + bias (N % 4 == 0, lda % 4 == 0, ldc % 4 == 0). */
int64_t sat_gemm_f32x3_packed_bytes(int N, int K);
int sat_gemm_f32x3_pack(const float* W, int N, int K, void* packed, sat_stream_t stream);
int sat_gemm_f32x3(const float* A, int64_t lda, const void* packed, const float* bias, float* C, int64_t ldc, int M, int N, int K,
                   sat_stream_t stream);

/* split-K variant of sat_gemm_f32: K-steps dealt to ksplit slices, slice z writes C + z*slab_stride (bias in slice 0);
 * sat_sum_slabs_f32 adds the slices in fixed order. */
int sat_gemm_f32_splitk(int amode, int bmode, const float* A, int64_t lda, const float* B, int64_t ldb,
                        float* C, int64_t ldc, const float* bias, const float* bias2, int M, int N, int K,
                        int ksplit, int64_t slab_stride, sat_stream_t stream);
int sat_sum_slabs_f32(const float* in, int nslab, int64_t slab_stride, int64_t n, float* out, sat_stream_t stream);
/* out[M,N] = A[M,K] * op(W) + bias for FEW rows (a decode / recurrence step: M <= a few hundred): K is split over the waves of
 * a workgroup and over grid slices so that 64 rows still fill the chip (the per-step GEMMs of model2.py:54-62 and their
 * backward).  w_kmajor 0: W[n*ldw + k]; 1: W[k*ldw + n].  K % 4 == 0, lda % 4 == 0.  Exact f32 (v_mfma_f32_16x16x4_f32). */
int sat_skinny_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, int w_kmajor, int M, int N, int K,
                        const float* bias, float* out, int64_t ldo, float* workspace, int64_t ws_bytes, sat_stream_t stream);
int64_t sat_skinny_gemm_ws_bytes(int M, int N, int K);
/* ... with an optional second product into the same output: out = A W + A2 W2 + bias (both W in the layout w_kmajor names);
 * workspace as for (M, N, K). */
int sat_skinny_gemm2_f32(const float* A, int64_t lda, const float* W, int64_t ldw, int K, const float* A2, int64_t lda2,
                         const float* W2, int64_t ldw2, int K2, int w_kmajor, int M, int N, const float* bias, float* out,
                         int64_t ldo, float* workspace, int64_t ws_bytes, sat_stream_t stream);
/* greedy argmax of one decode step (models.py:61-63): ids[b*ids_stride] = first argmax_v (h[b] . w[v] + b[v]) */
int sat_vocab_argmax(const float* h /*[B,H]*/, const float* w, const float* b, int B, int H, int V,
                     int64_t* ids, int64_t ids_stride, float* workspace, int64_t ws_bytes, sat_stream_t stream);
int64_t sat_vocab_argmax_ws_bytes(int B, int V);
/* one greedy decode step of one LSTM layer, state in place: x[B,In] -> h_out[B,H]; c[B,H] updated */
int sat_lstm_step(const float* x, const float* h_in, float* c, const float* w_ih, const float* w_hh,
                  const float* b_ih, const float* b_hh, int B, int In, int H, float* h_out, sat_stream_t stream);
/* rows of an embedding table: out[b] = embed[ids[b*ids_stride]] */
int sat_embed_rows(const float* embed, const int64_t* ids, int64_t ids_stride, int B, int E, int V,
                   float* out, sat_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Show-Attend-Tell decoder (model2.py:38-111: the model train.py:37 constructs).  f32, rows = batch rows of one step.
 * sat_attention_fwd: `attention_layer` (model2.py:73-78) for `rows` batch rows:
 *   h_att = tanh(ctx_enc[b] + proj[b][None,:]); alpha[b] = softmax_p(h_att . w_att); context[b] = mean_p(alpha[b,p] feats[b,p,:])
 *   ctx_enc, feats: [rows][P][C] (context_encode / features, model2.py:45-46); proj = weight_hh(hidden) [rows][ld_proj];
 *   alpha [rows][P] (nullable); context [rows][ld_ctx] (ld_ctx lets it land inside the LSTMCell input row).
 * sat_attention_bwd: its backward given d_context (h_att recomputed): d_ctx_enc[rows][P][C] += ..., d_proj [rows][C],
 *   d_watt_part [rows][C] (sum over rows = d weight_att; kept per row for a fixed-order reduction); d_feats (fine-tuning only):
 *   the gradient the weighted mean sends back into the features, accumulated.
 *   Both run as two launches (a per-position row-dot over (rows x position chunks) workgroups, then a per-channel pass over
 *   (rows x C/64) workgroups) so that a 64-row step fills the chip; the workspace carries the P-long vector between them.
 * sat_lstmcell_fwd: one nn.LSTMCell step (model2.py:58), c in place, optional tapes (activated gates, new c).
 * sat_rows_copy: out[r] = in[idx ? idx[r*idx_stride] : r] (embedding rows into a strided destination, state slices).
 * sat_rows_sum: out[c] (+)= sum_r in[r][c] in fixed order.
 */
int sat_attention_fwd(const float* ctx_enc, const float* feats, const float* proj, int64_t ld_proj, const float* w_att,
                      int rows, int P, int C, float* alpha, float* context, int64_t ld_ctx, float* workspace, int64_t ws_bytes,
                      sat_stream_t stream);
int64_t sat_attention_ws_bytes(int rows, int P);     /* raw scores (forward) / d_alpha (backward) between the two launches */
int sat_attention_bwd(const float* ctx_enc, const float* feats, const float* proj, int64_t ld_proj, const float* w_att,
                      const float* alpha, const float* d_ctx, int64_t ld_dctx,
                      const float* d_ctx2 /* nullable: d_context = d_ctx + d_ctx2 (its two consumers, model2.py:58 and :82) */,
                      int64_t ld_dctx2, int rows, int P, int C,
                      float* d_ctx_enc, float* d_proj, float* d_watt_part, float* d_feats /*[rows][P][C] += , or NULL*/,
                      float* workspace, int64_t ws_bytes, sat_stream_t stream);
int sat_lstmcell_fwd(const float* x /*[B,In]*/, const float* h_in /*[B,H]*/, float* c /*[B,H] in place*/, const float* w_ih,
                     const float* w_hh, const float* b_ih, const float* b_hh, int B, int In, int H, float* h_out,
                     float* gates /*[B,4H] or NULL*/, float* c_tape /*[B,H] or NULL*/, sat_stream_t stream);
int sat_rows_copy(const float* in, int64_t ldi, const int64_t* idx, int64_t idx_stride, int64_t nrows_in, int rows, int cols,
                  float* out, int64_t ldo, sat_stream_t stream);
int sat_rows_sum(const float* in, int64_t ld, int rows, int cols, float* out, int accumulate, sat_stream_t stream);
/* out[r] = a[r] + b[r] over `cols` columns of strided rows */
int sat_rows_add(const float* a, int64_t lda, const float* b, int64_t ldb, int rows, int cols, float* out, int64_t ldo,
                 sat_stream_t stream);
/* packed token ids out[row(t,b)] = captions[b][t + col0] (prefix/T/N as for sat_pack_targets: the embedding rows a step uses) */
int sat_pack_tokens(const int64_t* captions, int64_t cap_stride, const int32_t* prefix /*device*/, int T, int N, int col0,
                    int64_t* out, sat_stream_t stream);
/* dense embedding gradient (nn.Embedding, sparse=False; model2.py:28): table[V,E] = 0, then table[ids[n]] += rows[n] with a
 * fixed summation order (first occurrence sums its duplicates in row order) */
int sat_scatter_rows_add(const float* rows /*[N,E]*/, const int64_t* ids /*[N]*/, int N, int E, int V, float* table,
                         sat_stream_t stream);
/* pointwise backward of one nn.LSTMCell step: dh = dh_out (+ dh_carry for rows < n_carry), dc from dc_state (rows < n_carry);
 * DG[n,4H] = d(pre-activation gates); dc_state = d c_prev.  gates/c: the step's tapes; c_prev NULL = zeros. */
int sat_lstmcell_bwd_point(const float* dh_out /*[n,H]*/, const float* dh_carry /*[>=n_carry,H] or NULL*/, int n_carry,
                           const float* gates /*[n,4H]*/, const float* c /*[n,H]*/, const float* c_prev /*[n,H] or NULL*/,
                           float* dc_state /*[n,H]*/, float* DG /*[n,4H]*/, int n, int H, sat_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Conv-stack backward for fine-tuning (model2.py:87-89 `finetune(allow=True)`), f32 NHWC.  A 3x3/s1/p1 conv layer
 * y = relu(conv(x, W) + b) is differentiated with the kernels the forward already has:
 *   dZp = sat_pad_nhwc_f32(dY, mask = y)          zero-bordered d(pre-activation)               [N][H+2][W+2][Cout]
 *   db  = sat_colsum_f32(dZp)                     (the border rows are zeros)
 *   dW[:, kh, kw, :] = dZp^T . shift(Xp, kh, kw)  nine sat_gemm_f32_splitk (amode 2, bmode 1) over the flat padded pixel index:
 *                                                 with BOTH operands in the zero-bordered layout a tap is a constant flat offset
 *   dX  = SAT_OP_CONV(dZp as a pre-padded image, W flipped and transposed)                       (the forward conv kernel)
 * sat_maxpool2_bwd_f32: gradient of SAT_OP_MAXPOOL2 (first maximum in scan order, torch's tie rule).
 * sat_bcast_add_f32: out[b][p][:] += scale * v[b][:] (backward of the mean over positions, model2.py:68).
 */
int sat_pad_nhwc_f32(const float* in, const float* relu_mask_of /*or NULL*/, int N, int H, int W, int C, int pad, float* out,
                     sat_stream_t stream);
int sat_maxpool2_bwd_f32(const float* x, const float* dy, int N, int Hin, int Win, int C, float* dx, sat_stream_t stream);
int sat_bcast_add_f32(const float* v, int B, int P, int C, float scale, float* out, sat_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * On-device collate (data_loader.py:48-62 `collate_fn`): the batch arrives in dataset order -- images
 * [B][3*H*W] f32 and ragged captions (flat ids + offsets[B+1]) already in HBM; `order[r]` [device int32] is the sample that
 * lands in row r (decreasing caption length, ties in dataset order: the host derives it from the host-side lengths, which
 * it needs anyway for the packed-sequence bookkeeping).
 * sat_collate_captions: out[r][t] = t < len(order[r]) ? flat[offsets[order[r]] + t] : 0   (the zero-padded LongTensor)
 * sat_gather_rows_f32:  out row r = in row order[r]                                          (torch.stack in sorted order)
 */
int sat_collate_captions(const int64_t* flat, const int64_t* offsets /*[B+1]*/, const int32_t* order /*[B]*/, int B, int Tmax,
                         int64_t* out /*[B][Tmax]*/, sat_stream_t stream);
int sat_gather_rows_f32(const float* in, const int32_t* order, int rows, int64_t cols, float* out, sat_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Beam decode (the reference has only a stub, model2.py:113-114, next to the greedy loop
 * models.py:56-67).  Rows are (image b, hypothesis k) = b*K + k, K <= 8.
 * sat_beam_step: candidates (k, v) score scores_in[b,k] + log_softmax(logits[b*K+k])[v]; the best K of the K*V per
 *   image (ties: lower k*V+v) give parent[b,r] (k), token[b,r] (v), scores_out[b,r], r best-first.  A hypothesis with
 *   scores_in = -inf is dead.  last_tokens/end_id (NULL / <0 to disable): a hypothesis whose last token is end_id
 *   only continues with end_id, at unchanged score.
 * sat_beam_gather_rows: dst[b*K+k] = src[b*K+parent[b,k]] (LSTM h/c re-ordering), width floats per row, dst != src.
 * sat_beam_backtrack: parents/tokens [T][B*K] back-pointers -> ids [B*K][T].
 */
int sat_beam_step(const float* logits /*[B*K, ldl]*/, int64_t ldl, const float* scores_in /*[B*K]*/,
                  const int64_t* last_tokens /*[B*K] or NULL*/, int64_t end_id, int B, int K, int V,
                  int32_t* parent /*[B*K]*/, int64_t* token /*[B*K]*/, float* scores_out /*[B*K]*/,
                  void* workspace, int64_t ws_bytes, sat_stream_t stream);
int64_t sat_beam_step_ws_bytes(int B, int K);
int sat_beam_gather_rows(const float* src, const int32_t* parent, int B, int K, int width, float* dst,
                         sat_stream_t stream);
int sat_beam_backtrack(const int32_t* parents, const int64_t* tokens, int T, int B, int K, int64_t* ids,
                       sat_stream_t stream);
/* The whole decode loop of one batch as ONE call -- the 20 steps of `DecoderRNN.sample` (models.py:56-67; eval.py:99) enqueued from
 * C instead of step by step from the host language (round 4: seven launches per step through the FFI made the beam loop host-bound,
 * 2.44 ms per 20 steps for ~1.3 ms of GPU work).  Same kernels, same order as the step entry points: bit-identical results.
 * lstm_w: HOST array of 4 * num_layers device pointers (w_ih, w_hh, b_ih, b_hh per layer).
 * sat_beam_decode: beam width K <= 8 (the reference's sample_beam is a stub, model2.py:113-114); end_id < 0: none.  ids [B][K][steps]
 *   best first, scores [B][K] (nullable).  workspace: sat_beam_decode_ws_bytes, 256-byte aligned.
 * sat_greedy_decode: h / c [num_layers][B][H] = the initial LSTM state in, the final one out (models.py:61 hands `states` to the
 *   LSTM); h_tmp same size, x_tmp [B][E] scratch; ids [B][ids_stride], column i = step i; workspace: sat_vocab_argmax_ws_bytes. */
int64_t sat_beam_decode_ws_bytes(int B, int K, int E, int H, int V, int num_layers, int steps);
int sat_beam_decode(const float* features /*[B,E]*/, const float* embed /*[V,E]*/, const float* const* lstm_w /*[host]*/, int num_layers,
                    const float* lin_w /*[V,H]*/, const float* lin_b, int B, int K, int E, int H, int V, int steps, int64_t end_id,
                    int64_t* ids, float* scores, void* workspace, int64_t ws_bytes, sat_stream_t stream);
int sat_greedy_decode(const float* features, const float* embed, const float* const* lstm_w /*[host]*/, int num_layers,
                      const float* lin_w, const float* lin_b, int B, int E, int H, int V, int steps, float* h, float* c, float* h_tmp,
                      float* x_tmp, int64_t* ids, int64_t ids_stride, float* workspace, int64_t ws_bytes, sat_stream_t stream);
/* The truncation rule of the reference's id -> word loop (`evaluation`, eval.py:103-109: `if word == '<end>': break`):
 * kept[b] = number of ids of row b in front of the first end_id (T when the row has none).  ids: [B] rows of T int64 with
 * `stride` elements between rows (model.sample's [B,20], or one hypothesis plane of the beam ids). */
int sat_kept_tokens(const int64_t* ids, int64_t stride, int B, int T, int64_t end_id, int32_t* kept /*[B]*/,
                    sat_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * clip_gradient (train.py:88-91) + optim.Adam step (train.py:56,146) over one flat buffer.
 * clip <= 0 disables the clamp.  step is the 1-based Adam step count.
 */
int sat_clamp_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                        float beta2, float eps, float clip, int step, sat_stream_t stream);
/* The same update behind a device-side guard: `skip_if_nonzero` (device f32 word, may be NULL) is read by the kernel, and a
 * non-zero value drops the WHOLE update -- p, g, m, v stay bit for bit what they were.  The fused trainer aims it at the step's
 * fault slot (below), so a step whose persistent LSTM recurrence gave up never reaches the parameters. */
int sat_clamp_adam_step_guarded(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                float beta2, float eps, float clip, int step, const float* skip_if_nonzero,
                                sat_stream_t stream);
/* One step's fault flag: *slot = *sticky = (*sticky != 0 || any of the n_words (<= 8) u32 device status words != 0) ? 1.0f : 0.0f
 * (`status_words`: HOST array of device pointers -- sat_lstm_fwd_status_offset / sat_lstm_bwd_status_offset words of the
 * workspaces this step's calls ran with; `sticky`: device f32 the caller keeps across steps and clears once it has seen the
 * fault, may be NULL).  The trainer puts `slot` into the flat gradient buffer's trailing floats, so that in data-parallel
 * training the flag rides the last bucket's all-reduce(sum) and EVERY rank skips the update of a step one rank lost. */
int sat_step_fault_flag(const void* const* status_words, int n_words, float* sticky, float* slot, sat_stream_t stream);

/* column sums: out[c] = sum_r x[r*ld + c]  (bias gradients) */
int sat_colsum_f32(const float* x, int64_t ld, int rows, int cols, float* out, sat_stream_t stream);
/* f32 -> bf16 / bf16 -> f32 casts (weight shadow copies) */
int sat_cast_f32_bf16(const float* in, void* out, int64_t n, sat_stream_t stream);
int sat_cast_bf16_f32(const void* in, float* out, int64_t n, sat_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
