// Host-side orchestration of the hot path: the encoder op program (models.py:25-29), one LSTM layer
// forward / backward over a packed batch (models.py:52, train.py:144), the vocab projection + CE
// (models.py:53, train.py:143) and the encoder head.  Launch-only code: no allocation, no host sync,
// so a whole training step can be captured into one hipGraph by the caller.
#include "sat_internal.h"
#include <stdio.h>
#include <stdlib.h>

extern "C" int sat_version(void) { return SAT_ABI_VERSION; }

extern "C" const char* sat_error_string(int code) {
    switch (code) {
        case SAT_OK: return "ok";
        case SAT_ERR_ARG: return "sat: bad argument (null pointer, shape or alignment)";
        case SAT_ERR_WORKSPACE: return "sat: workspace too small";
        case SAT_ERR_UNSUPPORTED: return "sat: unsupported configuration";
        default: return hipGetErrorString((hipError_t)code);
    }
}

extern "C" int sat_run_ops(const sat_op* ops, int n_ops, sat_stream_t stream) {
    return sat_run_ops_parity(ops, n_ops, 0, stream);
}

extern "C" int sat_run_ops_parity(const sat_op* ops, int n_ops, int parity, sat_stream_t stream) {
    if (!ops || n_ops < 0 || (parity != 0 && parity != 1)) return SAT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n_ops; ++i) {
        const sat_op* op = ops + i;
        int rc;
        switch (op->kind) {
            case SAT_OP_IMAGE_PREP: rc = sat_image_prep_launch(op, s); break;
            case SAT_OP_CONV: rc = sat_conv_launch(op, parity, s); break;
            case SAT_OP_BN_FINALIZE: rc = sat_bn_finalize_launch(op, parity, s); break;
            case SAT_OP_BN_RELU: rc = sat_bn_act_launch(op, false, parity, s); break;
            case SAT_OP_BN_ADD_RELU: rc = sat_bn_act_launch(op, true, parity, s); break;
            case SAT_OP_BN_RELU_MAXPOOL: rc = sat_bn_relu_maxpool_launch(op, parity, s); break;
            case SAT_OP_AVGPOOL: rc = sat_avgpool_launch(op, s); break;
            case SAT_OP_BN_EVAL_BATCH: rc = sat_bn_eval_batch_launch(op, s); break;
            case SAT_OP_MAXPOOL2: rc = sat_maxpool2_launch(op, s); break;
            case SAT_OP_MAXPOOL3S2: rc = sat_pool3_launch(op, false, s); break;
            case SAT_OP_AVGPOOL3: rc = sat_pool3_launch(op, true, s); break;
            case SAT_OP_GRAM: rc = sat_gram_launch(op, parity, s); break;
            case SAT_OP_GRAM_COV: rc = sat_gram_cov_launch(op, s); break;
            case SAT_OP_GEMM_BF16_NT: rc = sat_gemm_bf16_op_launch(op, s); break;
            case SAT_OP_BN_FROM_GRAM: rc = sat_bn_from_gram_launch(op, s); break;
            default: rc = SAT_ERR_UNSUPPORTED;
        }
        if (rc != SAT_OK) return rc;
    }
    return SAT_OK;
}

struct sat_graph {
    hipGraphExec_t exec;
};

extern "C" int sat_graph_create(const sat_op* ops, int n_ops, int parity, sat_graph** graph_out) {
    if (!ops || n_ops <= 0 || !graph_out || (parity != 0 && parity != 1)) return SAT_ERR_ARG;
    *graph_out = nullptr;
    hipStream_t cs = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) return (int)e;
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(cs);
        return (int)e;
    }
    const int rc = sat_run_ops_parity(ops, n_ops, parity, (sat_stream_t)cs);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(cs, &g);           // always end the capture, also after a failed launch
    (void)hipStreamDestroy(cs);
    if (rc != SAT_OK || e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return rc != SAT_OK ? rc : (e != hipSuccess ? (int)e : SAT_ERR_UNSUPPORTED);
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return (int)e;
    sat_graph* out = new sat_graph;
    out->exec = exec;
    *graph_out = out;
    return SAT_OK;
}

extern "C" int sat_graph_launch(sat_graph* graph, sat_stream_t stream) {
    if (!graph || !graph->exec) return SAT_ERR_ARG;
    return (int)hipGraphLaunch(graph->exec, (hipStream_t)stream);
}

extern "C" int sat_graph_destroy(sat_graph* graph) {
    if (!graph) return SAT_ERR_ARG;
    hipError_t e = graph->exec ? hipGraphExecDestroy(graph->exec) : hipSuccess;
    delete graph;
    return (int)e;
}

// conv -> BatchNorm(batch statistics) -> ReLU of one bottleneck stage as a single call over three op records
// (SURVEY 8b names this entry point): statistics leave the conv epilogue, finalize makes (scale, shift), bnrelu applies.
extern "C" int sat_conv_bn_relu_fwd(const sat_op* conv, const sat_op* finalize, const sat_op* bnrelu,
                                    sat_stream_t stream) {
    if (!conv || !finalize || !bnrelu) return SAT_ERR_ARG;
    if (conv->kind != SAT_OP_CONV || finalize->kind != SAT_OP_BN_FINALIZE || bnrelu->kind != SAT_OP_BN_RELU)
        return SAT_ERR_ARG;
    if (conv->out != bnrelu->in0 || conv->Cout != finalize->Cout || conv->Cout != bnrelu->Cout) return SAT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    SAT_TRY(sat_conv_launch(conv, 0, s));
    SAT_TRY(sat_bn_finalize_launch(finalize, 0, s));
    return sat_bn_act_launch(bnrelu, false, 0, s);
}

void sat_conv_arm_timer(hipEvent_t start, hipEvent_t stop);      // sat_conv_glds.hip

// Diagnostics, NOT the hot path (creates events and synchronises the stream): run the program once in order and
// report every bf16 SAT_OP_CONV launch's own duration (the dispatch packet's begin/end timestamps, i.e. what
// rocprofv3 --kernel-trace reports for that launch) in op_us[i]; 0 for every other op.
extern "C" int sat_run_ops_timed(const sat_op* ops, int n_ops, int parity, sat_stream_t stream, float* op_us) {
    if (!ops || n_ops < 0 || !op_us || (parity != 0 && parity != 1)) return SAT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t* ev = new hipEvent_t[2 * (size_t)(n_ops > 0 ? n_ops : 1)];
    int n_ev = 0, rc = SAT_OK;
    for (int i = 0; i < n_ops && rc == SAT_OK; ++i) {
        const sat_op* op = ops + i;
        op_us[i] = 0.0f;
        const bool timed = op->kind == SAT_OP_CONV && op->dtype == SAT_BF16 && (op->Cout % 8) == 0;
        if (timed) {
            if (hipEventCreate(&ev[n_ev]) != hipSuccess) { rc = SAT_ERR_UNSUPPORTED; break; }
            if (hipEventCreate(&ev[n_ev + 1]) != hipSuccess) { (void)hipEventDestroy(ev[n_ev]); rc = SAT_ERR_UNSUPPORTED; break; }
            sat_conv_arm_timer(ev[n_ev], ev[n_ev + 1]);
            op_us[i] = -1.0f;          // marks "events n_ev, n_ev+1 belong to this op"
            n_ev += 2;
        }
        rc = sat_run_ops_parity(op, 1, parity, stream);
        sat_conv_arm_timer(nullptr, nullptr);     // a launch path that ignored the timer must not leave it armed
    }
    hipError_t e = hipStreamSynchronize(s);
    if (rc == SAT_OK && e != hipSuccess) rc = (int)e;
    int k = 0;
    for (int i = 0; i < n_ops; ++i) {
        if (op_us[i] >= 0.0f) continue;
        if (k + 2 > n_ev) { op_us[i] = 0.0f; continue; }       // event creation failed part-way
        float ms = 0.0f;
        if (rc == SAT_OK && hipEventElapsedTime(&ms, ev[k], ev[k + 1]) == hipSuccess) op_us[i] = ms * 1e3f;
        else op_us[i] = 0.0f;
        k += 2;
    }
    for (int j = 0; j < n_ev; ++j) (void)hipEventDestroy(ev[j]);
    delete[] ev;
    return rc;
}

// ------------------------------------------------------------------------------------------------------
// encoder head
static int fc_split(int F) { return F >= 1024 ? 8 : (F >= 256 ? 4 : 1); }

extern "C" int64_t sat_fc_bn1d_ws_bytes(int B, int F, int E) {
    return (int64_t)(fc_split(F) + 1) * B * E * sizeof(float);
}

extern "C" int sat_fc_bn1d_fwd(const float* pooled, const float* w_fc, const float* b_fc, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               int training, int B, int F, int E, float* feats, float* xhat, float* rstd,
                               float* workspace, int64_t ws_bytes, sat_stream_t stream) {
    if (!pooled || !w_fc || !b_fc || !gamma || !beta || !running_mean || !running_var || !feats || !xhat || !rstd ||
        !workspace)
        return SAT_ERR_ARG;
    if (ws_bytes < sat_fc_bn1d_ws_bytes(B, F, E)) return SAT_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nz = fc_split(F);
    float* part = workspace;                     // [nz][B][E]
    float* zbuf = workspace + (long)nz * B * E;  // [B][E]
    SAT_TRY(sat_skinny_store(pooled, F, w_fc, F, 0, B, E, F, nz, part, E, (long)B * E, nullptr, s));
    return sat_bn1d_fwd_launch(part, nz, (long)B * E, b_fc, gamma, beta, running_mean, running_var, momentum, eps,
                               training, B, E, zbuf, feats, xhat, rstd, s);
}

extern "C" int sat_fc_bn1d_bwd(const float* dy, const float* pooled, const float* xhat, const float* rstd,
                               const float* gamma, int B, int F, int E, float* dw_fc, float* db_fc, float* dgamma,
                               float* dbeta, float* workspace, int64_t ws_bytes, sat_stream_t stream) {
    if (!dy || !pooled || !xhat || !rstd || !gamma || !dw_fc || !db_fc || !dgamma || !dbeta || !workspace)
        return SAT_ERR_ARG;
    if (ws_bytes < (int64_t)B * E * (int64_t)sizeof(float)) return SAT_ERR_WORKSPACE;
    if ((E & 3) || (F & 3)) return SAT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    float* dz = workspace;
    SAT_TRY(sat_bn1d_bwd_launch(dy, xhat, rstd, gamma, B, E, dz, dgamma, dbeta, db_fc, s));
    // dW_fc[E,F] = dz^T[E,B] * pooled[B,F]: K = B is short -- a row-ordered fmaf chain per output (the MFMA GEMM took 35 us here)
    if (B <= 1024) return sat_outer_wgrad_launch(dz, pooled, B, E, F, dw_fc, s);
    return sat_gemm_f32(2, 1, dz, E, pooled, F, dw_fc, F, nullptr, nullptr, E, F, B, stream);
}

// ------------------------------------------------------------------------------------------------------
// LSTM layer over a packed batch
bool sat_lstm_persist_ok(int B, int H, int T, int n_cu);          // sat_lstm_persist.hip
int sat_lstm_persist_rows(int B, int H, int T, int n_cu);
int sat_lstm_persist_launch(float* GA, const float* W, float* CS, float* HS, float* HP, const int32_t* batch_sizes, int T,
                            int H, int rows, void* workspace, int64_t ws_bytes, hipStream_t s);
int64_t sat_lstm_persist_bwd_ws_bytes(int B, int H);
int sat_lstm_persist_bwd_launch(const float* dHS, const float* GA, const float* CS, const float* W, float* DG,
                                const int32_t* batch_sizes, int T, int H, void* xch, unsigned* err, hipStream_t s);

static int device_cu_count() {
    static int n = -1;
    if (n < 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) n = v;
        else n = 0;
    }
    return n;
}

static int lstm_fwd_impl(const float* X, const float* w_ih, const float* w_hh, const float* b_ih,
                         const float* b_hh, const int32_t* batch_sizes, int T, int In, int H, float* GA,
                         float* CS, float* HS, float* HP, float* c_state, void* workspace, int64_t ws_bytes,
                         void* mixed, int64_t mixed_bytes, sat_stream_t stream) {
    if (!X || !w_ih || !w_hh || !b_ih || !b_hh || !batch_sizes || !GA || !CS || !HS || !HP || !c_state || T < 1)
        return SAT_ERR_ARG;
    if ((In & 3) || (H & 3)) return SAT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int B = batch_sizes[0];
    long N = 0;
    for (int t = 0; t < T; ++t) {
        if (batch_sizes[t] < 1 || (t > 0 && batch_sizes[t] > batch_sizes[t - 1])) return SAT_ERR_ARG;
        N += batch_sizes[t];
    }
    // x-gates for every packed row in one batched MFMA GEMM: GA = X * W_ih^T + b_ih + b_hh
    // (bf16 throughput mode: the same product on the bf16 matrix pipe from bf16 copies of X and W_ih, f32 accumulate and output)
    if (mixed && mixed_bytes >= sat_gemm_mixed_scratch_bytes((int)N, 4 * H, In))
        SAT_TRY(sat_gemm_mixed_nt(X, In, 0, w_ih, In, 0, GA, 4L * H, b_ih, b_hh, (int)N, 4 * H, In, mixed, mixed_bytes, s));
    else
        SAT_TRY(sat_gemm_f32(0, 0, X, In, w_ih, In, GA, 4L * H, b_ih, b_hh, (int)N, 4 * H, In, stream));
    hipError_t e = hipMemsetAsync(HP, 0, (size_t)B * H * sizeof(float), s);   // h_{-1} = 0 for the rows of step 0
    if (e != hipSuccess) return (int)e;
    // the recurrence: ONE persistent launch (W_hh in registers, per-group hidden-state exchange) when every workgroup
    // can be resident and the caller brought the exchange workspace; otherwise one launch per step
    static const int persist_env = getenv("SAT_LSTM_PERSIST") ? atoi(getenv("SAT_LSTM_PERSIST")) : 1;
    const int prow = (persist_env && workspace && ws_bytes >= sat_lstm_fwd_ws_bytes(B, H)) ? sat_lstm_persist_rows(B, H, T, device_cu_count()) : 0;
    if (prow) return sat_lstm_persist_launch(GA, w_hh, CS, HS, HP, batch_sizes, T, H, prow, workspace, ws_bytes, s);
    // one launch per step: the status word the caller reads back (sat_lstm_fwd_status_offset) reports a clean run
    const int64_t soff = sat_lstm_fwd_status_offset(B, H);
    if (workspace && soff >= 0 && ws_bytes >= soff + 64) {
        e = hipMemsetAsync((char*)workspace + soff, 0, 64, s);
        if (e != hipSuccess) return (int)e;
    }
    e = hipMemsetAsync(c_state, 0, (size_t)B * H * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    long off = 0;
    for (int t = 0; t < T; ++t) {
        const int n = batch_sizes[t];
        const int n_next = (t + 1 < T) ? batch_sizes[t + 1] : 0;
        float* ga = GA + off * 4 * H;
        // recurrent step: gates = x-gates + h_{t-1} * W_hh^T; activated gates overwrite the x-gates in place
        SAT_TRY(sat_skinny_lstm(HP + off * H, w_hh, nullptr, nullptr, 0, nullptr, nullptr, ga, 4L * H, n, H, c_state,
                                ga, 4L * H, CS + off * H, HS + off * H, n_next ? HP + (off + n) * H : nullptr, n_next, s));
        off += n;
    }
    return SAT_OK;
}

extern "C" int sat_lstm_fwd(const float* X, const float* w_ih, const float* w_hh, const float* b_ih,
                            const float* b_hh, const int32_t* batch_sizes, int T, int In, int H, float* GA,
                            float* CS, float* HS, float* HP, float* c_state, void* workspace, int64_t ws_bytes,
                            sat_stream_t stream) {
    return lstm_fwd_impl(X, w_ih, w_hh, b_ih, b_hh, batch_sizes, T, In, H, GA, CS, HS, HP, c_state, workspace, ws_bytes, nullptr, 0, stream);
}

// scratch for the bf16-pipe forms of an LSTM layer's batched GEMMs (x-gates, dW_ih, dW_hh, dX): N = packed rows
extern "C" int64_t sat_lstm_mixed_ws_bytes(int N, int In, int H) {
    int64_t m = sat_gemm_mixed_scratch_bytes(N, 4 * H, In);
    const int64_t c[3] = {sat_gemm_mixed_scratch_bytes(4 * H, In, N), sat_gemm_mixed_scratch_bytes(4 * H, H, N),
                          sat_gemm_mixed_scratch_bytes(N, In, 4 * H)};
    for (int i = 0; i < 3; ++i) m = c[i] > m ? c[i] : m;
    return m;
}

// sat_lstm_fwd with the x-gates GEMM (models.py:52, the input half of nn.LSTM) on the bf16 matrix pipe; the recurrence stays f32
extern "C" int sat_lstm_fwd_bf16(const float* X, const float* w_ih, const float* w_hh, const float* b_ih,
                                 const float* b_hh, const int32_t* batch_sizes, int T, int In, int H, float* GA,
                                 float* CS, float* HS, float* HP, float* c_state, void* workspace, int64_t ws_bytes,
                                 void* mixed_ws, int64_t mixed_bytes, sat_stream_t stream) {
    if (!mixed_ws) return SAT_ERR_ARG;
    return lstm_fwd_impl(X, w_ih, w_hh, b_ih, b_hh, batch_sizes, T, In, H, GA, CS, HS, HP, c_state, workspace, ws_bytes, mixed_ws,
                         mixed_bytes, stream);
}

static int lstm_bwd_split(int H) {
    const int ncg = sat_cdiv(H, 16);
    int nz = 256 / (ncg > 0 ? ncg : 1);
    return nz < 1 ? 1 : (nz > 16 ? 16 : nz);
}

extern "C" int64_t sat_lstm_bwd_ws_bytes(int B, int H) {
    return (int64_t)(lstm_bwd_split(H) + 1) * B * H * sizeof(float);
}

// split-K factor that fills the chip when a GEMM has fewer than ~192 tiles of 64x64 (the batched dX / dW_ih GEMMs at
// small In): the K range is dealt over grid.y and the slabs are summed in fixed order
static int fill_split(long M, long Nn, long K) {
    const long tiles = (long)sat_cdiv(M, 64) * sat_cdiv(Nn, 64);
    if (tiles >= 192 || K < 256) return 1;
    long ks = 384 / (tiles > 0 ? tiles : 1);
    const long max_by_k = K / 128;
    if (ks > max_by_k) ks = max_by_k;
    return (int)(ks < 1 ? 1 : (ks > 8 ? 8 : ks));
}

// Layout of the FULL backward workspace (round 5: the exchange FIRST, at offsets that depend on (B, H) only, so that one buffer
// sized for the longest batch serves every N -- with real captions N changes almost every step):
//   [0, xb - 64)      granule exchange of the persistent backward recurrence: a region of its own, nothing else writes there;
//   [xb - 64, xb)     the recurrence's status word          (xb = sat_lstm_persist_bwd_ws_bytes(B, H))
//   [head, head + m)  dh_part / dc_state of the per-step form, then the split-K slabs of the batched GEMMs (m depends on N)
// (an H the persistent recurrence does not run has no exchange: the head is then the status word alone)
static int64_t lstm_bwd_xb(int B, int H) {
    const int64_t xb = sat_lstm_persist_bwd_ws_bytes(B, H);
    return xb >= 64 ? xb : 64;
}
static int64_t lstm_bwd_head_bytes(int B, int H) { return (lstm_bwd_xb(B, H) + 255) / 256 * 256; }
static int64_t lstm_bwd_slab_bytes(int N, int B, int In, int H) {
    const int64_t base = sat_lstm_bwd_ws_bytes(B, H);
    const int64_t dx = (int64_t)fill_split(N, In, 4L * H) * N * In * sizeof(float);
    const int64_t dw = (int64_t)fill_split(4L * H, In, N) * 4 * H * In * sizeof(float);
    int64_t m = dx > dw ? dx : dw;
    if (base > m) m = base;
    return (m + 255) / 256 * 256;
}

// workspace that additionally lets sat_lstm_bwd run its under-filled weight/input-gradient GEMMs split-K and its recurrence as
// ONE persistent launch, for exactly N packed rows ...
extern "C" int64_t sat_lstm_bwd_ws_bytes_full(int N, int B, int In, int H) {
    return lstm_bwd_head_bytes(B, H) + lstm_bwd_slab_bytes(N, B, In, H);
}
// ... and for ANY N <= n_max (split-K factors are not monotonic in N: the bound takes the largest factor): size the buffer once with
// this, hand it to every call
extern "C" int64_t sat_lstm_bwd_ws_bytes_max(int n_max, int B, int In, int H) {
    int64_t m = sat_lstm_bwd_ws_bytes(B, H);
    const int64_t dx = 8LL * n_max * In * sizeof(float), dw = 8LL * 4 * H * In * sizeof(float);
    if (dx > m) m = dx;
    if (dw > m) m = dw;
    return lstm_bwd_head_bytes(B, H) + (m + 255) / 256 * 256;
}

// Byte offset of the backward recurrence's STATUS WORD (uint32) in the full workspace -- a function of (B, H) only (N and In are
// kept in the signature for ABI stability): zeroed by every sat_lstm_bwd call that got the full workspace, set non-zero when the
// persistent backward recurrence gave up waiting for its group (DG and every gradient of that call are then INVALID) -- read it back
// like sat_lstm_fwd_status_offset's word.
extern "C" int64_t sat_lstm_bwd_status_offset(int N, int B, int In, int H) {
    (void)N; (void)In;
    return lstm_bwd_xb(B, H) - 64;
}

static int lstm_bwd_impl(const float* dHS, const float* X, const float* w_ih, const float* w_hh, const float* GA,
                         const float* CS, const float* HP, const int32_t* batch_sizes, int T, int In, int H,
                         float* DG, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, float* dX,
                         float* workspace, int64_t ws_bytes, void* mixed, int64_t mixed_bytes, sat_stream_t stream) {
    if (!dHS || !X || !w_ih || !w_hh || !GA || !CS || !HP || !batch_sizes || !DG || !dw_ih || !dw_hh || !db_ih ||
        !db_hh || !workspace || T < 1)
        return SAT_ERR_ARG;
    if ((In & 3) || (H & 3)) return SAT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int B = batch_sizes[0];
    if (ws_bytes < sat_lstm_bwd_ws_bytes(B, H)) return SAT_ERR_WORKSPACE;
    long N = 0;
    for (int t = 0; t < T; ++t) N += batch_sizes[t];
    const int nz = lstm_bwd_split(H);
    const long slab = (long)B * H;
    const int64_t full = sat_lstm_bwd_ws_bytes_full((int)N, B, In, H);
    const bool roomy = ws_bytes >= full;
    char* const ws0 = (char*)workspace;
    if (roomy) workspace = (float*)(ws0 + lstm_bwd_head_bytes(B, H));       // the slab region sits behind the exchange + status head
    float* dc_state = workspace + nz * slab; // [B][H]  (the nz slabs in front of it: room kept for the split-K GEMMs below)
    hipError_t e = hipMemsetAsync(dc_state, 0, (size_t)slab * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    // the recurrence: ONE persistent launch (W_hh in registers, per-group exchange of the d(pre-activation) rows) when every
    // workgroup can be resident and the caller brought the full workspace (its last 64 bytes = the status word); otherwise one
    // launch per step
    static const int persist_bwd = getenv("SAT_LSTM_PERSIST_BWD") ? atoi(getenv("SAT_LSTM_PERSIST_BWD")) : 1;
    bool recurrence_done = false;
    if (roomy) {
        unsigned* status = (unsigned*)(ws0 + sat_lstm_bwd_status_offset((int)N, B, In, H));
        if (persist_bwd && sat_lstm_persist_bwd_ws_bytes(B, H) > 0 && sat_lstm_persist_ok(B, H, T, device_cu_count())) {
            SAT_TRY(sat_lstm_persist_bwd_launch(dHS, GA, CS, w_hh, DG, batch_sizes, T, H, ws0, status, s));
            recurrence_done = true;
        } else {
            e = hipMemsetAsync(status, 0, 64, s);
            if (e != hipSuccess) return (int)e;
        }
    }
    long off = N;
    for (int t = T - 1; t >= 0 && !recurrence_done; --t) {
        const int n = batch_sizes[t];
        off -= n;
        const int n_next = (t + 1 < T) ? batch_sizes[t + 1] : 0;
        const float* cs_prev = (t > 0) ? CS + (off - batch_sizes[t - 1]) * H : nullptr;
        // ONE launch per step: dh_t = dHS_t + DG_{t+1} W_hh (K = 4H inside the workgroup) and the gate backward of step t
        SAT_TRY(sat_lstm_bwd_step(dHS + off * H, n_next ? DG + (off + n) * 4 * H : nullptr, n_next, w_hh, GA + off * 4 * H,
                                  CS + off * H, cs_prev, dc_state, DG + off * 4 * H, n, H, s));
    }
    // batched weight gradients over all packed rows (the recurrence above is done with the workspace: it is free for
    // split-K slabs when the caller sized it with sat_lstm_bwd_ws_bytes_full)
    if (mixed && mixed_bytes >= sat_lstm_mixed_ws_bytes((int)N, In, H)) {
        // bf16 throughput mode: the three batched GEMMs on the bf16 matrix pipe (f32 accumulate / output); DG, X, HP are read
        // "k-major" (the packed-row axis is the contraction), W_ih k-major for dX
        SAT_TRY(sat_gemm_mixed_nt(DG, 4L * H, 1, X, In, 1, dw_ih, In, nullptr, nullptr, 4 * H, In, (int)N, mixed, mixed_bytes, s));
        SAT_TRY(sat_gemm_mixed_nt(DG, 4L * H, 1, HP, H, 1, dw_hh, H, nullptr, nullptr, 4 * H, H, (int)N, mixed, mixed_bytes, s));
        SAT_TRY(sat_colsum_f32(DG, 4L * H, (int)N, 4 * H, db_ih, stream));
        e = hipMemcpyAsync(db_hh, db_ih, (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
        if (dX) SAT_TRY(sat_gemm_mixed_nt(DG, 4L * H, 0, w_ih, In, 1, dX, In, nullptr, nullptr, (int)N, In, 4 * H, mixed, mixed_bytes, s));
        return SAT_OK;
    }
    const int ks_dw = roomy ? fill_split(4L * H, In, N) : 1;
    if (ks_dw > 1) {
        SAT_TRY(sat_gemm_f32_splitk(2, 1, DG, 4L * H, X, In, workspace, In, nullptr, nullptr, 4 * H, In, (int)N, ks_dw,
                                    (int64_t)4 * H * In, stream));
        SAT_TRY(sat_sum_slabs_f32(workspace, ks_dw, (int64_t)4 * H * In, (int64_t)4 * H * In, dw_ih, stream));
    } else {
        SAT_TRY(sat_gemm_f32(2, 1, DG, 4L * H, X, In, dw_ih, In, nullptr, nullptr, 4 * H, In, (int)N, stream));
    }
    SAT_TRY(sat_gemm_f32(2, 1, DG, 4L * H, HP, H, dw_hh, H, nullptr, nullptr, 4 * H, H, (int)N, stream));
    SAT_TRY(sat_colsum_f32(DG, 4L * H, (int)N, 4 * H, db_ih, stream));
    e = hipMemcpyAsync(db_hh, db_ih, (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
    if (dX) {
        const int ks_dx = roomy ? fill_split(N, In, 4L * H) : 1;
        if (ks_dx > 1) {
            SAT_TRY(sat_gemm_f32_splitk(0, 1, DG, 4L * H, w_ih, In, workspace, In, nullptr, nullptr, (int)N, In, 4 * H, ks_dx,
                                        (int64_t)N * In, stream));
            SAT_TRY(sat_sum_slabs_f32(workspace, ks_dx, (int64_t)N * In, (int64_t)N * In, dX, stream));
        } else {
            SAT_TRY(sat_gemm_f32(0, 1, DG, 4L * H, w_ih, In, dX, In, nullptr, nullptr, (int)N, In, 4 * H, stream));
        }
    }
    return SAT_OK;
}

extern "C" int sat_lstm_bwd(const float* dHS, const float* X, const float* w_ih, const float* w_hh, const float* GA,
                            const float* CS, const float* HP, const int32_t* batch_sizes, int T, int In, int H,
                            float* DG, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, float* dX,
                            float* workspace, int64_t ws_bytes, sat_stream_t stream) {
    return lstm_bwd_impl(dHS, X, w_ih, w_hh, GA, CS, HP, batch_sizes, T, In, H, DG, dw_ih, dw_hh, db_ih, db_hh, dX, workspace, ws_bytes,
                         nullptr, 0, stream);
}

// sat_lstm_bwd with dW_ih, dW_hh and dX on the bf16 matrix pipe (the recurrence and the bias gradients stay f32)
extern "C" int sat_lstm_bwd_bf16(const float* dHS, const float* X, const float* w_ih, const float* w_hh, const float* GA,
                                 const float* CS, const float* HP, const int32_t* batch_sizes, int T, int In, int H,
                                 float* DG, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, float* dX,
                                 float* workspace, int64_t ws_bytes, void* mixed_ws, int64_t mixed_bytes, sat_stream_t stream) {
    if (!mixed_ws) return SAT_ERR_ARG;
    return lstm_bwd_impl(dHS, X, w_ih, w_hh, GA, CS, HP, batch_sizes, T, In, H, DG, dw_ih, dw_hh, db_ih, db_hh, dX, workspace, ws_bytes,
                         mixed_ws, mixed_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------
// vocab projection + CE
extern "C" int sat_vocab_logits_fwd(const float* Hs, const float* w, const float* b, int N, int H, int V,
                                    float* logits, int64_t ldl, sat_stream_t stream) {
    if (!Hs || !w || !b || !logits || ldl < V) return SAT_ERR_ARG;
    return sat_gemm_f32(0, 0, Hs, H, w, H, logits, ldl, b, nullptr, N, V, H, stream);
}

static int vocab_bwd_split(int N, int H, int V) {
    // dHs[N,H] = dlogits[N,V] * W[V,H]: K = V is long and M*N small -> deal K over enough slices to fill the chip
    const long tiles = (long)sat_cdiv(N, 64) * sat_cdiv(H, 64);
    int ks = (int)(512 / (tiles > 0 ? tiles : 1));
    const int nk = sat_cdiv(V, 32);
    if (ks > nk / 8) ks = nk / 8;
    return ks < 1 ? 1 : (ks > 16 ? 16 : ks);
}

extern "C" int64_t sat_vocab_ce_bwd_ws_bytes(int N, int H, int V) {
    const int ks = vocab_bwd_split(N, H, V);
    return ks > 1 ? (int64_t)ks * N * H * sizeof(float) : 0;
}

extern "C" int sat_sum_slabs_f32(const float* in, int nslab, int64_t slab_stride, int64_t n, float* out, sat_stream_t stream);
extern "C" int sat_gemm_f32_splitk(int amode, int bmode, const float* A, int64_t lda, const float* B, int64_t ldb,
                                   float* C, int64_t ldc, const float* bias, const float* bias2, int M, int N, int K,
                                   int ksplit, int64_t slab_stride, sat_stream_t stream);

extern "C" int sat_vocab_ce_bwd(const float* dlogits, int64_t ldl, const float* Hs, const float* w, int N, int H, int V,
                                float* dw, float* db, float* dHs, float* workspace, int64_t ws_bytes,
                                sat_stream_t stream) {
    if (!dlogits || !Hs || !w || !dw || !db || !dHs) return SAT_ERR_ARG;
    // any V: the dlogits rows are padded to a multiple of 4 floats (ldl), pad columns must hold zeros
    if ((H & 3) || (ldl & 3) || ldl < ((V + 3) & ~3)) return SAT_ERR_UNSUPPORTED;
    const int ks = vocab_bwd_split(N, H, V);
    if (ks > 1 && (!workspace || ws_bytes < sat_vocab_ce_bwd_ws_bytes(N, H, V))) return SAT_ERR_WORKSPACE;
    // dW[V,H] = dlogits^T * Hs ;  db = colsum(dlogits) ;  dHs[N,H] = dlogits * W
    SAT_TRY(sat_gemm_f32(2, 1, dlogits, ldl, Hs, H, dw, H, nullptr, nullptr, V, H, N, stream));
    SAT_TRY(sat_colsum_f32(dlogits, ldl, N, V, db, stream));
    if (ks == 1) return sat_gemm_f32(0, 1, dlogits, ldl, w, H, dHs, H, nullptr, nullptr, N, H, V, stream);
    SAT_TRY(sat_gemm_f32_splitk(0, 1, dlogits, ldl, w, H, workspace, H, nullptr, nullptr, N, H, V, ks, (int64_t)N * H, stream));
    if (((long)N * H) & 3) return SAT_ERR_UNSUPPORTED;
    return sat_sum_slabs_f32(workspace, ks, (int64_t)N * H, (int64_t)N * H, dHs, stream);
}
