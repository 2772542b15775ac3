/*
 * lnsfaid_capi.hip — host side of the C ABI declared in include/lnsfaid.h.
 *
 * Owns the context (device tables, per-codeword state in HBM, one HIP stream, events) and drives the
 * relaunch loop described at the top of lnsfaid_kernels.hip.  There is no CPU decode path in this
 * library: without a GPU lnsfaid_create() fails with LNSFAID_E_NODEVICE.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h> /* types and enums only: the library is bound at run time (lf_rccl below) */

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "lnsfaid_device.h"
#include "lnsfaid_swar.h" /* sw_nms_fits / sw_nms_tables (host side) */

extern "C" hipError_t lf_launch_decode(int method, int uniform_w, const LfKernelArgs* args, size_t lds_bytes,
                                       hipStream_t stream);
extern "C" const void* lf_decode_func(int method, int uniform_w);
extern "C" int lf_decode_threads(void);
extern "C" const void* lf_decode4_func(int method, int ef, int rm);
extern "C" int lf_decode4_threads(void);
extern "C" hipError_t lf_launch_decode4(int method, int ef, int rm, const LfKernelArgs* args, size_t lds_bytes, hipStream_t stream);
extern "C" int lf_decode4_rm_layers(void);
extern "C" const void* lf_decode5_func(int method);
extern "C" int lf_decode5_threads(void);
extern "C" hipError_t lf_launch_decode5(int method, const LfKernelArgs* args, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t lf_launch_count_errors(const int8_t* decoded, const int8_t* input_bits, int n_var, int k_info,
                                             size_t n_cw, unsigned long long* out, hipStream_t stream);

extern "C" hipError_t lf_launch_frontend(const uint32_t* d_seeds, const unsigned long long* d_draws, int n_streams, int mod_type,
                                         float sigma_ch, float scale, const int8_t* d_codeword, const int8_t* d_frames, int n_var,
                                         int n_check, int interleave, int fast, int8_t* d_fix, hipStream_t stream);
extern "C" hipError_t lf_frontend_fastpath_scan(double* d_out2, hipStream_t stream);
extern "C" void lf_frontend_fastpath_assumed(double* eps2);

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
struct LfCombiner;
static thread_local char g_hip_err[256] = "";
/* Contexts alive in this process.  A host thread that waits for its stream by spinning (hipStreamSynchronize) is the
 * lowest-latency choice for a few contexts, and the worst one for the reference's call shape - one context per worker thread,
 * 64 of them (reference main.cpp:31-34, :164-172): the spinning threads then fight for the cores the HIP runtime itself needs.
 * From LF_SPIN_CONTEXTS live contexts on, waits go through an event created with hipEventBlockingSync (the thread sleeps
 * until the interrupt); LNSFAID_SYNC=spin / block overrides. */
static std::atomic<int> g_live_contexts(0);
#define LF_SPIN_CONTEXTS 4
static bool sync_blocking()
{
    static const char* e = getenv("LNSFAID_SYNC");
    if (e && e[0] == 's') return false;
    if (e && e[0] == 'b') return true;
    return g_live_contexts.load(std::memory_order_relaxed) > LF_SPIN_CONTEXTS;
}

static int hip_fail(hipError_t e, const char* what)
{
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
    return LNSFAID_E_HIP;
}
#define HIP_TRY(call)                                      \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return hip_fail(e_, #call);  \
    } while (0)

#define LF_IO_CHUNKS 8 /* pieces the host-pointer path is cut into when both host buffers are pinned */
#define LF_MAX_CHAIN 6 /* decode launches queued back to back before the host looks at their "codewords left" counters */

struct lnsfaid_ctx {
    int device = 0;
    size_t max_groups = 0;
    LfDevCode hcode;
    LfDevCfg hcfg;
    int n_var = 0, n_check = 0, k_info = 0;
    size_t lds_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_chain[LF_MAX_CHAIN + 1] = {}; /* time stamps around the launches of a chain */
    hipEvent_t ev_block = nullptr;              /* hipEventBlockingSync: where the host waits when many contexts are alive */
    uint32_t slot_seen[LF_MAX_CHAIN] = {};      /* last value read of each running "codewords left" counter */
    int predicted_launches = 1;                 /* launches the previous batch needed: how many this one queues at once */
    hipStream_t s_in = nullptr, s_out = nullptr; /* copy streams of the pipelined host-pointer path */
    hipEvent_t ev_in[LF_IO_CHUNKS] = {};
    LfDevCode* d_code = nullptr;
    LfDevCfg* d_cfg = nullptr;
    int8_t* d_en = nullptr;
    uint4* d_rows = nullptr;
    uint32_t* d_bits = nullptr;
    LfLaneState* d_lane = nullptr;
    int32_t* d_status[2] = { nullptr, nullptr };
    uint32_t* d_remaining = nullptr;
    int32_t* d_live = nullptr;
    unsigned long long* d_counters = nullptr;
    uint32_t* h_remaining = nullptr;          /* pinned, LF_MAX_CHAIN words */
    unsigned long long* h_counters = nullptr; /* pinned */
    /* staging for the host-pointer entry points */
    int8_t* d_io_in = nullptr;
    int8_t* d_io_out = nullptr;
    lnsfaid_group_stats* d_io_stats = nullptr;
    int rows_per_lane = 0; /* 0: pick per configuration; 2 / 4: forced (lnsfaid_select_kernel) */
    int waves_per_cw = 0;  /* 0 / 1: one wave per codeword; 2: lnsfaid_kernel5.hip where it applies (lnsfaid_select_waves) */
    int msg_store = 0;     /* 0: pick per code; 1: registers; 2: streamed through HBM (lnsfaid_select_message_store) */
    struct LfCombiner* comb = nullptr; /* call combiner this one-group context is a member of (see below) */
    int comb_slot = -1;
    const void* checked_fn = nullptr; /* kernel instance kernel_check() last looked at */
    int resident_wg = 0, lds_wg = 0;  /* its workgroups per CU: what the occupancy query says / what its LDS alone allows */
    void* comm = nullptr;      /* ncclComm_t for lnsfaid_allreduce_counters */
    bool comm_owned = false;
    unsigned long long* d_reduce = nullptr;
    double kernel_ms = 0.0;
    uint64_t kernel_launches = 0;
    /* device front-end scratch: seeds, draw counters, transmitted codeword */
    uint32_t* d_fe_seeds = nullptr;
    unsigned long long* d_fe_draws = nullptr;
    int8_t* d_fe_codeword = nullptr;
    int8_t* d_fe_frames = nullptr; /* per-stream sent frames (lnsfaid_frontend_set_frames), encoder output layout */
    int8_t* d_fe_input = nullptr;  /* their information bits, [stream][32][K] */
    size_t fe_frames_streams = 0;  /* 0: frames not in use */
    int fe_interleave = 1;         /* InterleaveModType of the device front-end */
    int fe_exact = 0;              /* 1: every symbol through the double-precision chain (lnsfaid_frontend_set_exact) */
};

/* wait until everything queued on the context's stream so far has finished */
static int stream_wait(lnsfaid_ctx* ctx)
{
    if (sync_blocking()) {
        HIP_TRY(hipEventRecord(ctx->ev_block, ctx->stream));
        HIP_TRY(hipEventSynchronize(ctx->ev_block));
    } else {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return LNSFAID_OK;
}

/* ---- code analysis: PosNoeudsVariable -> circulants ------------------------------------------------- */
static int weight_class(int w) { return w == 3 ? 0 : (w == 6 ? 1 : (w == 11 ? 2 : 3)); } /* CDecoder_FAID.cpp:692-705 */

static int build_code(const lnsfaid_code* code, LfDevCode* out)
{
    memset(out, 0, sizeof(*out));
    if (!code || !code->pos_vn || !code->deg || !code->deg_rows) return LNSFAID_E_INVAL;
    const int Z = code->z, N = code->n_var, M = code->n_check, E = code->n_edges;
    if (Z != LF_Z) return LNSFAID_E_CODE; /* kernels map the 256 rows of a circulant onto 128 threads x 2 rows */
    if (N <= 0 || M <= 0 || N % Z || M % Z || M >= N) return LNSFAID_E_CODE;
    const int nbr = M / Z, nbc = N / Z, K = N - M;
    if (nbr > LF_MAX_BR || nbc > LF_MAX_BC) return LNSFAID_E_CODE;
    if (K % 4 || M % 4 || N % LF_T) return LNSFAID_E_CODE; /* dword staging, 64-wide ballots over VNs */
    if (code->puncture_tail < 0 || code->puncture_tail > N) return LNSFAID_E_CODE;
    /* degree of each check row from the DEG_k / DEG_k_COMPUTATIONS classes */
    std::vector<int> row_deg;
    row_deg.reserve(M);
    long e_total = 0;
    for (int k = 0; k < code->nb_degres; ++k) {
        if (code->deg_rows[k] < 0 || code->deg[k] < 2 || code->deg[k] > LF_MAX_DEG) return LNSFAID_E_CODE;
        for (int i = 0; i < code->deg_rows[k]; ++i) { row_deg.push_back(code->deg[k]); e_total += code->deg[k]; }
    }
    if ((int)row_deg.size() != M || e_total != E) return LNSFAID_E_CODE;

    size_t e = 0;
    for (int br = 0; br < nbr; ++br) {
        const int deg = row_deg[(size_t)br * Z];
        out->deg[br] = deg;
        const uint16_t* row0 = code->pos_vn + e;
        for (int j = 0; j < 64; ++j) out->sbtab[br][j] = 0u;
        for (int j = 0; j < 32; ++j) out->sbplain[br][j] = 0u;
        int prev_cb = -1;
        for (int j = 0; j < deg; ++j) {
            const int cb = row0[j] / Z, sh = row0[j] % Z;
            if (row0[j] >= N || cb <= prev_cb) return LNSFAID_E_CODE; /* ascending, no block column twice */
            prev_cb = cb;
            out->circ[br][j].sb = (uint32_t)cb * (uint32_t)Z + (uint32_t)sh;
            out->sbtab[br][LF_JCODE_A(j)] = out->sbtab[br][LF_JCODE_B(j)] = out->circ[br][j].sb;
            out->sbplain[br][j] = (((uint32_t)cb * (uint32_t)Z) << 16) | (((uint32_t)sh) << 2);
            out->s4tab[br][j] = ((uint32_t)sh) << 2;
            out->cbtab[br][j] = (uint32_t)cb * (uint32_t)Z;
            if (out->col_weight[cb] >= LF_MAX_COLW) return LNSFAID_E_CODE;
            out->colcirc[cb][out->col_weight[cb]++] = (uint32_t)br | ((uint32_t)sh << 8);
        }
        for (int i = 0; i < Z; ++i) {
            if (row_deg[(size_t)br * Z + i] != deg) return LNSFAID_E_CODE;
            for (int j = 0; j < deg; ++j) {
                const uint32_t sb = out->circ[br][j].sb;
                const int want = (int)(sb / (uint32_t)Z * (uint32_t)Z + (sb % (uint32_t)Z + (uint32_t)i) % (uint32_t)Z);
                if (code->pos_vn[e + (size_t)i * deg + j] != want) return LNSFAID_E_CODE; /* not quasi-cyclic */
            }
        }
        e += (size_t)deg * Z;
    }
    {   /* syndrome walk table: needs the LDS layout (n_words = N / 32, p_words = M / 32) */
        const int nw = N / 32, pw = M / 32;
        const uint32_t hard0 = lf_lds_off_hard(N), zero = lf_lds_off_zero(N, nw, pw);
        if (lf_lds_bytes(N, nw, pw) > 0xffffu) return LNSFAID_E_CODE; /* 16-bit addresses */
        for (int br = 0; br < LF_MAX_BR; ++br)
            for (int j = 0; j < LF_MAX_DEG; ++j)
                for (int k = 0; k < 8; ++k) {
                    uint2 e; e.x = zero | (zero << 16); e.y = 0;
                    if (br < nbr && j < out->deg[br]) {
                        const uint32_t sb = out->circ[br][j].sb, cb = sb / (uint32_t)Z, sh = sb % (uint32_t)Z;
                        const uint32_t o = (32u * (uint32_t)k + sh) & 255u, q = o >> 5;
                        e.x = (hard0 + 4u * (cb * 8u + q)) | ((hard0 + 4u * (cb * 8u + ((q + 1u) & 7u))) << 16);
                        e.y = o & 31u;
                    }
                    out->synw[br][j][k] = e;
                }
    }
    for (int br = 0; br < nbr; ++br)
        for (int j = 0; j < out->deg[br]; ++j)
            out->circ[br][j].wclass = (uint32_t)weight_class(out->col_weight[out->circ[br][j].sb / (uint32_t)Z]);
    out->n_var = N; out->n_check = M; out->k_info = K; out->nbr = nbr; out->nbc = nbc;
    out->puncture_tail = code->puncture_tail;
    out->n_words = N / 32; out->p_words = M / 32;
    return LNSFAID_OK;
}

/* block columns the bit-flipping stage may touch: column weight == REGULAR_COL_WEIGHT */
static void build_wcols(LfDevCode* code, int W)
{
    code->n_wcols = 0;
    for (int cb = 0; cb < code->nbc; ++cb)
        if (code->col_weight[cb] == W) code->wcol[code->n_wcols++] = cb;
    /* first occurrence of every weight-W block column in row order (reference era_[]: CDecoder_FAID.cpp:673-680) */
    bool seen[LF_MAX_BC] = {};
    for (int br = 0; br < LF_MAX_BR; ++br) {
        code->era_edges[br] = 0;
        for (int j = 0; br < code->nbr && j < code->deg[br]; ++j) {
            const int cb = (int)(code->circ[br][j].sb / LF_Z);
            if (code->col_weight[cb] == W && !seen[cb]) { seen[cb] = true; code->era_edges[br] |= 1u << j; }
        }
    }
}

static int build_cfg(const lnsfaid_cfg* cfg, LfDevCfg* out)
{
    if (!cfg) return LNSFAID_E_INVAL;
    if (cfg->decode_method < 0 || cfg->decode_method > 5) return LNSFAID_E_INVAL;
    if (cfg->max_iteration < 0 || cfg->max_iteration > (1 << 20)) return LNSFAID_E_INVAL;
    if (cfg->max_bf_iter < 0 || cfg->max_bf_iter > (1 << 20)) return LNSFAID_E_INVAL;
    if (cfg->regular_col_weight < 0 || cfg->regular_col_weight > LF_MAX_COLW) return LNSFAID_E_INVAL;
    memset(out, 0, sizeof(*out));
    out->method = cfg->decode_method;
    out->max_iter = cfg->max_iteration;
    /* OMS: VECTOR_SET1 int8 lanes; NMS: VECTOR_SET2 int16 lanes (CLDPC.cpp:337, :345) */
    out->factor_1 = cfg->decode_method == 0 ? (int16_t)cfg->factor_1 : (int8_t)cfg->factor_1;
    out->factor_2 = cfg->decode_method == 0 ? (int16_t)cfg->factor_2 : (int8_t)cfg->factor_2;
    out->floor_err_count = cfg->floor_err_count;
    out->floor_iter_thresh = cfg->floor_iter_thresh;
    out->ef = cfg->ef_elimination;
    out->max_bf = (cfg->decode_method == 0 || cfg->decode_method == 1) ? 0 : cfg->max_bf_iter; /* Decode / Decode_OMS have no BF stage */
    out->L0 = cfg->bf_L0; out->L1 = cfg->bf_L1; out->alpha = cfg->bf_alpha; out->delta = cfg->bf_delta;
    out->W = cfg->regular_col_weight;
    out->hard2_thr = cfg->hard2_threshold;
    out->vote_cap = (int8_t)cfg->bf_vote_cap;
    /* OMS offsets (CDecoder_OMS.cpp:388-425) leave the 3-bit message alphabet for Factor_1 < 0 or Factor_2 < 1: the
     * minimum 0 would become -1 */
    if ((cfg->decode_method == 1 || cfg->decode_method == 3 || cfg->decode_method == 4)
        && ((int8_t)cfg->factor_1 < 0 || (int8_t)cfg->factor_2 < 1)) return LNSFAID_E_INVAL;
    if (cfg->decode_method == 5 && cfg->ef_elimination != 1) return LNSFAID_E_INVAL;
    if (cfg->decode_method == 2 && (cfg->ef_elimination < 0 || cfg->ef_elimination > 2)) return LNSFAID_E_INVAL;
    if (cfg->decode_method != 2 && cfg->decode_method != 5 && cfg->ef_elimination != 0) return LNSFAID_E_INVAL;
    out->uniform_w = 1;
    for (int it = 0; it < 6; ++it)
        for (int w = 0; w < 4; ++w) {
            for (int a = 0; a < 8; ++a) {
                const int v = cfg->v2c_map[it][w][a], ve = cfg->v2c_map_ef[it][w][a];
                if ((cfg->decode_method == 2 || cfg->decode_method == 5) && (v < 0 || v > 7)) return LNSFAID_E_INVAL; /* 3-bit message alphabet */
                if ((cfg->decode_method == 5 || cfg->ef_elimination >= 1) && (ve < 0 || ve > 7)) return LNSFAID_E_INVAL;
                uint32_t* l = a < 4 ? &out->lut_lo[it][w] : &out->lut_hi[it][w];
                uint32_t* le = a < 4 ? &out->lut_ef_lo[it][w] : &out->lut_ef_hi[it][w];
                *l |= (uint32_t)(v & 0xff) << (8 * (a & 3));
                *le |= (uint32_t)(ve & 0xff) << (8 * (a & 3));
                /* a table that is not non-decreasing cannot be applied after the minimum search */
                if (a > 0 && (v < cfg->v2c_map[it][w][a - 1] || ((cfg->decode_method == 5 || cfg->ef_elimination >= 1) && ve < cfg->v2c_map_ef[it][w][a - 1])))
                    out->uniform_w = 0;
            }
            if (out->lut_lo[it][w] != out->lut_lo[it][0] || out->lut_hi[it][w] != out->lut_hi[it][0]) out->uniform_w = 0;
            if ((cfg->decode_method == 5 || cfg->ef_elimination >= 1)
                && (out->lut_ef_lo[it][w] != out->lut_ef_lo[it][0] || out->lut_ef_hi[it][w] != out->lut_ef_hi[it][0]))
                out->uniform_w = 0;
        }
    if (cfg->decode_method == 0) {
        out->uniform_w = (out->factor_1 == out->factor_2) ? 1 : 0; /* NMS: one factor -> patch path */
        out->nms_fits = sw_nms_fits(out->factor_1, out->factor_2) ? 1 : 0;
        if (out->nms_fits) sw_nms_tables(out->factor_1, out->nms_t);
    }
    /* Decode_FAID with EF_ELIMINATION 1 / 2 exists in the four-rows-per-lane kernel only, which needs uniform tables */
    if (cfg->decode_method == 2 && cfg->ef_elimination >= 1 && !out->uniform_w) return LNSFAID_E_INVAL;
    out->bf_fast = (out->W == 3 && ((int8_t)out->alpha == 0 || (int8_t)out->alpha == 1)) ? 1 : 0;
    return LNSFAID_OK;
}

/* DecodeMethod 3 counts the votes of a variable node in four bit planes (bf_step_plain): column weight <= 15.  Checked when a
 * context is created and again when lnsfaid_set_cfg switches an existing context to DecodeMethod 3. */
static int check_code_for_method(const LfDevCode* code, int method)
{
    if (method == 3)
        for (int cb = 0; cb < code->nbc; ++cb)
            if (code->col_weight[cb] > 15) return LNSFAID_E_CODE;
    return LNSFAID_OK;
}

/* ---- context ------------------------------------------------------------------------------------------- */
extern "C" int lnsfaid_comm_destroy(lnsfaid_ctx* ctx);
static int comb_join(lnsfaid_ctx* ctx);
static void comb_leave(lnsfaid_ctx* ctx, int slot);
extern "C" void lnsfaid_destroy(lnsfaid_ctx* ctx)
{
    if (!ctx) return;
    if (ctx->comb_slot >= 0) { comb_leave(ctx, ctx->comb_slot); ctx->comb_slot = -1; }
    g_live_contexts.fetch_sub(1, std::memory_order_relaxed);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)lnsfaid_comm_destroy(ctx);
    (void)hipFree(ctx->d_reduce);
    (void)hipFree(ctx->d_code); (void)hipFree(ctx->d_cfg); (void)hipFree(ctx->d_en); (void)hipFree(ctx->d_rows);
    (void)hipFree(ctx->d_bits); (void)hipFree(ctx->d_lane); (void)hipFree(ctx->d_status[0]); (void)hipFree(ctx->d_status[1]);
    (void)hipFree(ctx->d_remaining); (void)hipFree(ctx->d_live); (void)hipFree(ctx->d_counters);
    (void)hipFree(ctx->d_io_in); (void)hipFree(ctx->d_io_out); (void)hipFree(ctx->d_io_stats);
    (void)hipFree(ctx->d_fe_seeds); (void)hipFree(ctx->d_fe_draws); (void)hipFree(ctx->d_fe_codeword);
    (void)hipFree(ctx->d_fe_frames); (void)hipFree(ctx->d_fe_input);
    if (ctx->h_remaining) (void)hipHostFree(ctx->h_remaining);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (auto e : ctx->ev_chain) if (e) (void)hipEventDestroy(e);
    if (ctx->ev_block) (void)hipEventDestroy(ctx->ev_block);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->s_in) (void)hipStreamDestroy(ctx->s_in);
    if (ctx->s_out) (void)hipStreamDestroy(ctx->s_out);
    for (auto e : ctx->ev_in) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

static int alloc_state(lnsfaid_ctx* ctx);
static int create_impl(lnsfaid_ctx* ctx, const lnsfaid_code* code, const lnsfaid_cfg* cfg)
{
    int rc = build_code(code, &ctx->hcode);
    if (rc) return rc;
    rc = build_cfg(cfg, &ctx->hcfg);
    if (rc) return rc;
    ctx->n_var = ctx->hcode.n_var; ctx->n_check = ctx->hcode.n_check; ctx->k_info = ctx->hcode.k_info;
    rc = check_code_for_method(&ctx->hcode, ctx->hcfg.method);
    if (rc) return rc;
    build_wcols(&ctx->hcode, ctx->hcfg.W);
    return alloc_state(ctx);
}

/* device state of a context whose hcode / hcfg / device / max_groups are set */
static int alloc_state(lnsfaid_ctx* ctx)
{
    ctx->n_var = ctx->hcode.n_var; ctx->n_check = ctx->hcode.n_check; ctx->k_info = ctx->hcode.k_info;
    ctx->lds_bytes = lf_lds_bytes(ctx->n_var, ctx->hcode.n_words, ctx->hcode.p_words);
    if (ctx->lds_bytes > 64 * 1024) return LNSFAID_E_CODE;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || ctx->device < 0 || ctx->device >= ndev) {
        snprintf(g_hip_err, sizeof(g_hip_err), "no usable HIP device (count %d, asked for %d)", ndev, ctx->device);
        return LNSFAID_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&ctx->ev0));
    HIP_TRY(hipEventCreate(&ctx->ev1));
    const size_t n_cw = ctx->max_groups * LNSFAID_GROUP;
    HIP_TRY(hipMalloc(&ctx->d_code, sizeof(LfDevCode)));
    HIP_TRY(hipMalloc(&ctx->d_cfg, sizeof(LfDevCfg)));
    HIP_TRY(hipMalloc(&ctx->d_en, n_cw * (size_t)ctx->n_var));
    HIP_TRY(hipMalloc(&ctx->d_rows, n_cw * (size_t)ctx->hcode.nbr * LF_T * sizeof(uint4)));
    HIP_TRY(hipMalloc(&ctx->d_bits, n_cw * (size_t)3 * ctx->hcode.n_words * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&ctx->d_lane, n_cw * sizeof(LfLaneState)));
    HIP_TRY(hipMalloc(&ctx->d_status[0], n_cw * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&ctx->d_status[1], n_cw * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&ctx->d_remaining, LF_MAX_CHAIN * sizeof(uint32_t)));
    HIP_TRY(hipMemset(ctx->d_remaining, 0, LF_MAX_CHAIN * sizeof(uint32_t)));
    for (auto& e : ctx->ev_chain) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_block, hipEventBlockingSync | hipEventDisableTiming));
    HIP_TRY(hipMalloc(&ctx->d_live, n_cw * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&ctx->d_counters, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc((void**)&ctx->h_remaining, LF_MAX_CHAIN * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&ctx->h_counters, 4 * sizeof(unsigned long long), hipHostMallocDefault));
    HIP_TRY(hipMemcpy(ctx->d_code, &ctx->hcode, sizeof(LfDevCode), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_cfg, &ctx->hcfg, sizeof(LfDevCfg), hipMemcpyHostToDevice));
    return LNSFAID_OK;
}

extern "C" int lnsfaid_create(lnsfaid_ctx** out, const lnsfaid_code* code, const lnsfaid_cfg* cfg, int32_t device,
                              size_t max_groups)
{
    if (!out || !code || !cfg || max_groups == 0 || max_groups > ((size_t)1 << 24)) return LNSFAID_E_INVAL;
    *out = nullptr;
    lnsfaid_ctx* ctx = new (std::nothrow) lnsfaid_ctx();
    if (!ctx) return LNSFAID_E_NOMEM;
    ctx->device = device;
    ctx->max_groups = max_groups;
    if (const char* e = getenv("LNSFAID_ROWS_PER_LANE")) /* test / A-B switch: force the 2-rows-per-lane kernel for a whole run */
        ctx->rows_per_lane = (e[0] == '2') ? 2 : 0;
    if (const char* e = getenv("LNSFAID_WAVES_PER_CODEWORD")) /* test / A-B switch, see lnsfaid_select_waves */
        ctx->waves_per_cw = (e[0] == '2') ? 2 : 0;
    if (const char* e = getenv("LNSFAID_FRONTEND_EXACT")) /* test / A-B switch, see lnsfaid_frontend_set_exact */
        ctx->fe_exact = (e[0] == '1') ? 1 : 0;
    if (const char* e = getenv("LNSFAID_MSG_STORE")) /* test / A-B switch, see lnsfaid_select_message_store */
        ctx->msg_store = (e[0] == 'h') ? LNSFAID_MSG_HBM : ((e[0] == 'r') ? LNSFAID_MSG_REGISTERS : 0);
    g_live_contexts.fetch_add(1, std::memory_order_relaxed); /* (lnsfaid_destroy takes it back) */
    const int rc = create_impl(ctx, code, cfg);
    if (rc) { lnsfaid_destroy(ctx); return rc; }
    ctx->comb_slot = comb_join(ctx); /* also sets ctx->comb */
    *out = ctx;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_set_cfg(lnsfaid_ctx* ctx, const lnsfaid_cfg* cfg)
{
    if (!ctx || !cfg) return LNSFAID_E_INVAL;
    LfDevCfg n;
    int rc = build_cfg(cfg, &n);
    if (rc) return rc;
    rc = check_code_for_method(&ctx->hcode, n.method);
    if (rc) return rc;
    /* the reference re-reads Profile.txt in every decode call (CDecoder_FAID.cpp:178-179), so a drop-in binding calls this once
     * per call with what is almost always the same configuration: nothing to do then (no synchronisation, no upload) */
    if (memcmp(&n, &ctx->hcfg, sizeof(n)) == 0) return LNSFAID_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n.W != ctx->hcfg.W) { /* the bit-flipping column list depends on REGULAR_COL_WEIGHT */
        build_wcols(&ctx->hcode, n.W);
        HIP_TRY(hipMemcpy(ctx->d_code, &ctx->hcode, sizeof(LfDevCode), hipMemcpyHostToDevice));
    }
    ctx->hcfg = n;
    HIP_TRY(hipMemcpy(ctx->d_cfg, &ctx->hcfg, sizeof(LfDevCfg), hipMemcpyHostToDevice));
    return LNSFAID_OK;
}

/* the combiner's pool context takes the configuration of the batch it is about to decode (validated when the member built it) */
static void apply_devcfg(lnsfaid_ctx* ctx, const LfDevCfg& n)
{
    (void)hipStreamSynchronize(ctx->stream);
    ctx->hcfg = n; /* (same REGULAR_COL_WEIGHT as the pool's code: members are matched on the whole LfDevCode) */
    (void)hipMemcpy(ctx->d_cfg, &ctx->hcfg, sizeof(LfDevCfg), hipMemcpyHostToDevice);
}

/* ---- the hot path ------------------------------------------------------------------------------------ */
/* The four-rows-per-lane kernel (lnsfaid_kernel4.hip) covers DecodeMethods 1..5 with FAID tables that are uniform over the
 * weight classes and non-decreasing (OMS has no table), and DecodeMethod 0 with Factor_1 == Factor_2 >= 15 (sw_nms_fits); it
 * needs max degree <= 24 like the other one; everything else runs on the two-rows-per-lane kernel. */
static bool kernel4_possible(const lnsfaid_ctx* ctx)
{
    const int m = ctx->hcfg.method;
    if (m < 0 || m > 5) return false;
    if (m == 0) return ctx->hcfg.nms_fits != 0; /* one factor, at most 16 levels (the usual normalisation factors) */
    if ((m == 2 || m == 5) && !ctx->hcfg.uniform_w) return false;
    return true;
}
static bool use_kernel4(const lnsfaid_ctx* ctx)
{
    if (ctx->hcfg.method == 2 && ctx->hcfg.ef >= 1) return true; /* not built for the other kernel (build_cfg made sure it applies) */
    if (ctx->rows_per_lane == 2) return false;
    return kernel4_possible(ctx);
}

extern "C" int lnsfaid_select_kernel(lnsfaid_ctx* ctx, int32_t rows_per_lane)
{
    if (!ctx || (rows_per_lane != 0 && rows_per_lane != 2 && rows_per_lane != 4)) return LNSFAID_E_INVAL;
    if (rows_per_lane == 4 && !kernel4_possible(ctx)) return LNSFAID_E_INVAL;
    if (rows_per_lane == 2 && ctx->hcfg.method == 2 && ctx->hcfg.ef >= 1) return LNSFAID_E_INVAL;
    ctx->rows_per_lane = rows_per_lane;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_kernel_rows_per_lane(const lnsfaid_ctx* ctx) { return ctx ? (use_kernel4(ctx) ? 4 : 2) : LNSFAID_E_INVAL; }

/* Two wavefronts per codeword (lnsfaid_kernel5.hip): the four-rows kernel's configurations without DecodeMethod 0 and without the
 * erasing instance of EF_ELIMINATION 2; messages streamed through HBM. */
static bool kernel5_possible(const lnsfaid_ctx* ctx)
{
    const int m = ctx->hcfg.method;
    /* (the two waves exchange 7 dwords per lane through the LDS of the hard-decision plane: n_words words) */
    return kernel4_possible(ctx) && m >= 1 && m <= 5 && !(m == 2 && ctx->hcfg.ef == 2) && ctx->hcode.n_words >= 7 * 64;
}
static bool use_kernel5(const lnsfaid_ctx* ctx) { return ctx->waves_per_cw == 2 && use_kernel4(ctx) && kernel5_possible(ctx); }

extern "C" int lnsfaid_select_waves(lnsfaid_ctx* ctx, int32_t waves_per_codeword)
{
    if (!ctx || waves_per_codeword < 0 || waves_per_codeword > 2) return LNSFAID_E_INVAL;
    if (waves_per_codeword == 2 && !(use_kernel4(ctx) && kernel5_possible(ctx))) return LNSFAID_E_INVAL;
    ctx->waves_per_cw = waves_per_codeword;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_kernel_waves(const lnsfaid_ctx* ctx) { return ctx ? (use_kernel5(ctx) ? 2 : 1) : LNSFAID_E_INVAL; }

/* Where the four-rows kernel keeps the compressed check-to-variable messages between layers: in registers (codes of up to
 * lf_decode4_rm_layers() layers; not built for the erasing instance of EF_ELIMINATION 2) or streamed through HBM. */
static bool msg_registers_possible(const lnsfaid_ctx* ctx)
{
    if (ctx->hcfg.method == 0) return false; /* NMS: the 16-level minimum search leaves no room for them */
    return ctx->hcode.nbr <= lf_decode4_rm_layers() && !(ctx->hcfg.method == 2 && ctx->hcfg.ef == 2);
}
static bool use_msg_registers(const lnsfaid_ctx* ctx)
{
    if (!msg_registers_possible(ctx) || use_kernel5(ctx)) return false;
    return ctx->msg_store != LNSFAID_MSG_HBM;
}

static const void* selected_kernel(const lnsfaid_ctx* ctx, int* threads)
{
    if (use_kernel5(ctx)) {
        *threads = lf_decode5_threads();
        return lf_decode5_func(ctx->hcfg.method);
    }
    if (use_kernel4(ctx)) {
        *threads = lf_decode4_threads();
        return lf_decode4_func(ctx->hcfg.method, ctx->hcfg.ef, use_msg_registers(ctx) ? 1 : 0);
    }
    *threads = lf_decode_threads();
    return lf_decode_func(ctx->hcfg.method, ctx->hcfg.uniform_w);
}

extern "C" int lnsfaid_select_message_store(lnsfaid_ctx* ctx, int32_t where)
{
    if (!ctx || where < 0 || where > LNSFAID_MSG_HBM) return LNSFAID_E_INVAL;
    if (where == LNSFAID_MSG_REGISTERS && !msg_registers_possible(ctx)) return LNSFAID_E_INVAL;
    ctx->msg_store = where;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_message_store(const lnsfaid_ctx* ctx)
{
    if (!ctx) return LNSFAID_E_INVAL;
    if (!use_kernel4(ctx)) return LNSFAID_MSG_HBM; /* the two-rows kernel always streams them */
    return use_msg_registers(ctx) ? LNSFAID_MSG_REGISTERS : LNSFAID_MSG_HBM;
}

/* Build-time properties of the kernel instance the context is about to launch, looked at once per instance:
 *  - it must have no static LDS: the layer steps address En by its LDS offset, so the dynamic segment has to start at 0
 *    (adding a __shared__ variable to a decode kernel would otherwise corrupt En silently);
 *  - how many of its workgroups a CU holds.  The decoders are sized so that LDS alone decides that (50G-PON: 20 424 B per
 *    codeword, 8 per CU); one more register or LDS word in the wrong place halves it, which costs ~40 % of the throughput and
 *    nothing else would show.  Recorded for lnsfaid_kernel_residency, printed under LNSFAID_TRACE. */
static const void* selected_kernel(const lnsfaid_ctx* ctx, int* threads);
static int kernel_check(lnsfaid_ctx* ctx)
{
    int threads = 0;
    const void* fn = selected_kernel(ctx, &threads);
    if (!fn) return LNSFAID_E_INTERNAL;
    if (fn == ctx->checked_fn) return LNSFAID_OK;
    hipFuncAttributes at;
    HIP_TRY(hipFuncGetAttributes(&at, fn));
    if (at.sharedSizeBytes != 0) {
        snprintf(g_hip_err, sizeof(g_hip_err), "decode kernel has %zu bytes of static LDS: its En image would not start at LDS offset 0",
                 (size_t)at.sharedSizeBytes);
        return LNSFAID_E_INTERNAL;
    }
    int wg = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg, fn, threads, ctx->lds_bytes));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    const size_t lds_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 160 * 1024;
    const size_t gran = 512; /* LDS allocation granularity */
    int by_lds = (int)(lds_cu / ((ctx->lds_bytes + gran - 1) / gran * gran));
    const int by_waves = 32 / ((threads + 63) / 64); /* 32 wave slots per CU */
    if (by_lds > by_waves) by_lds = by_waves;
    ctx->checked_fn = fn; ctx->resident_wg = wg; ctx->lds_wg = by_lds;
    static const bool trace = getenv("LNSFAID_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "[lnsfaid] decode kernel: %d threads, %d VGPRs, %zu B LDS per workgroup, %d workgroups per CU (LDS alone: %d)%s\n",
                threads, at.numRegs, ctx->lds_bytes, wg, by_lds, wg < by_lds ? "  ** residency lost **" : "");
    return LNSFAID_OK;
}

extern "C" int lnsfaid_kernel_residency(lnsfaid_ctx* ctx, int32_t* workgroups_per_cu, int32_t* lds_limit)
{
    if (!ctx) return LNSFAID_E_INVAL;
    HIP_TRY(hipSetDevice(ctx->device));
    const int rc = kernel_check(ctx);
    if (rc) return rc;
    if (workgroups_per_cu) *workgroups_per_cu = ctx->resident_wg;
    if (lds_limit) *lds_limit = ctx->lds_wg;
    return LNSFAID_OK;
}

/* status_preloaded: ctx->d_status[0] already holds the decision point of every codeword (the call combiner marks the groups
 * that take no part in a batch as finished); otherwise every codeword is fresh */
static int decode_device_impl(lnsfaid_ctx* ctx, const int8_t* d_fixInput, size_t n_groups, int8_t* d_decodedBits,
                              lnsfaid_group_stats* d_stats, bool status_preloaded);
extern "C" int lnsfaid_decode_device(lnsfaid_ctx* ctx, const int8_t* d_fixInput, size_t n_groups, int8_t* d_decodedBits,
                                     lnsfaid_group_stats* d_stats)
{
    return decode_device_impl(ctx, d_fixInput, n_groups, d_decodedBits, d_stats, false);
}

static int decode_device_impl(lnsfaid_ctx* ctx, const int8_t* d_fixInput, size_t n_groups, int8_t* d_decodedBits,
                              lnsfaid_group_stats* d_stats, bool status_preloaded)
{
    if (!ctx || (n_groups && (!d_fixInput || !d_decodedBits))) return LNSFAID_E_INVAL;
    if (n_groups > ctx->max_groups) return LNSFAID_E_INVAL;
    if (n_groups == 0) return LNSFAID_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    {
        const int rc = kernel_check(ctx);
        if (rc) return rc;
    }
    const size_t n_cw = n_groups * LNSFAID_GROUP;
#ifdef LF4_LIVE_PROOF /* experiment build only (lnsfaid_kernel4.hip publish_pass) */
    HIP_TRY(hipMemsetAsync(ctx->d_live, 0, n_cw * sizeof(int32_t), ctx->stream));
#endif

    LfKernelArgs a;
    a.code = ctx->d_code; a.cfg = ctx->d_cfg;
    a.fix_input = d_fixInput; a.decoded = d_decodedBits;
    a.st_en = ctx->d_en; a.st_rows = ctx->d_rows; a.st_bits = ctx->d_bits; a.st_lane = ctx->d_lane;
    a.live = ctx->d_live; a.stats = d_stats; a.n_cw = (int32_t)n_cw;

    /* Every launch moves each unfinished group's front forward or finishes it; the time line has max_iter + max_bf + 1
     * points and a group needs at most two launches per point.  How many launches a batch takes is data dependent (one when
     * nothing converges, three or four with early stop) and only known from the "codewords left" counter of the previous
     * launch, so a host round trip per launch would sit between them - a quarter of the time of a one-group call.  Instead the
     * launches the PREVIOUS batch needed are queued back to back (a launch on a finished batch is a grid of immediate exits,
     * microseconds), their counters are read in one go, and only a batch that needs more continues one launch at a time.
     * No memset in front: the first launch takes "every codeword fresh" from a null status pointer, and the counters are
     * running sums of which the host remembers the last value it saw. */
    const long max_launches = 2L * ((long)ctx->hcfg.max_iter + ctx->hcfg.max_bf + 2) + 2 + LF_MAX_CHAIN;
    static const bool trace = getenv("LNSFAID_TRACE") != nullptr; /* read once: per-launch timing on stderr */
    int cur = 0;
    long launch = 0, needed = 0;
    bool done = false;
    while (!done) {
        if (launch >= max_launches) return LNSFAID_E_INTERNAL;
        int chain = launch == 0 ? ctx->predicted_launches : 1;
        if (chain < 1) chain = 1;
        if (chain > LF_MAX_CHAIN) chain = LF_MAX_CHAIN;
        for (int j = 0; j < chain; ++j) {
            a.status_cur = (launch + j == 0 && !status_preloaded) ? nullptr : ctx->d_status[cur];
            a.status_next = ctx->d_status[cur ^ 1];
            a.remaining = ctx->d_remaining + j;
            HIP_TRY(hipEventRecord(ctx->ev_chain[j], ctx->stream));
            if (use_kernel5(ctx)) HIP_TRY(lf_launch_decode5(ctx->hcfg.method, &a, ctx->lds_bytes, ctx->stream));
            else if (use_kernel4(ctx)) HIP_TRY(lf_launch_decode4(ctx->hcfg.method, ctx->hcfg.ef, use_msg_registers(ctx) ? 1 : 0, &a, ctx->lds_bytes, ctx->stream));
            else HIP_TRY(lf_launch_decode(ctx->hcfg.method, ctx->hcfg.uniform_w, &a, ctx->lds_bytes, ctx->stream));
            cur ^= 1;
        }
        HIP_TRY(hipEventRecord(ctx->ev_chain[chain], ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->h_remaining, ctx->d_remaining, (size_t)chain * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        { const int rcw = stream_wait(ctx); if (rcw) return rcw; }
        for (int j = 0; j < chain; ++j) {
            const uint32_t left = ctx->h_remaining[j] - ctx->slot_seen[j]; /* running counter, modulo 2^32 */
            ctx->slot_seen[j] = ctx->h_remaining[j];
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_chain[j], ctx->ev_chain[j + 1]));
            ctx->kernel_ms += ms;
            ctx->kernel_launches += 1;
            if (trace) fprintf(stderr, "[lnsfaid] launch %ld: %.3f ms, %u codewords left%s\n", launch + j, ms, left, done ? " (queued ahead, batch already complete)" : "");
            if (!done && left == 0) { done = true; needed = launch + j + 1; }
        }
        launch += chain;
    }
    ctx->predicted_launches = (int)needed;
    return LNSFAID_OK;
}

static int ensure_io(lnsfaid_ctx* ctx)
{
    if (ctx->d_io_in) return LNSFAID_OK;
    const size_t bytes = ctx->max_groups * LNSFAID_GROUP * (size_t)ctx->n_var;
    HIP_TRY(hipMalloc(&ctx->d_io_in, bytes));
    HIP_TRY(hipMalloc(&ctx->d_io_out, bytes));
    HIP_TRY(hipMalloc(&ctx->d_io_stats, ctx->max_groups * sizeof(lnsfaid_group_stats)));
    return LNSFAID_OK;
}

/* ---- the call combiner: many contexts, one group per call -> few launches ------------------------------------------------
 * The reference decodes ONE group of 32 frames per call from T worker threads, each owning a CLDPC (reference
 * CSimulate.cpp:136-164, main.cpp:164-172), and the drop-in binding gives every CLDPC its own context.
 * Taken literally that is T streams with a 32-wave launch each: the runtime multiplexes the streams onto a handful of hardware
 * queues, so only a few of those launches are on the chip at a time, and every call pays its own copies, launches and waits
 * (measured, profiles/r03_dropin: 64 threads reach 1.5 Gb/s of 70).  So contexts of ONE group on the same device that decode
 * the same code share a combiner: a call copies its fixInput into a slot of a pinned staging area and sleeps; one worker thread
 * per device gathers the calls that arrive within LF_COMB_WINDOW_US (or until every member has one pending), moves their slots
 * to the device in one copy, decodes them in ONE launch sequence on a pool context of LF_COMB_SLOTS groups - groups without a
 * pending call are marked finished, their workgroups exit at once -, copies the decisions back in one copy and wakes the
 * callers, each of which copies its slot out.  Results are those of the direct path bit for bit (groups never interact).
 * A context that is alone on its device, or whose kernel / message-store selection was changed by hand, takes the direct path;
 * LNSFAID_COALESCE=0 switches the combiner off. */
#define LF_COMB_SLOTS 128
#define LF_COMB_QUIET_US 150  /* a batch is closed when no further call has arrived for this long ...                      */
#define LF_COMB_WINDOW_US 1500 /* ... or this long after its first call, or as soon as every member has a call pending       */
#define LF_COMB_DEVICES 16
#define LF_COMB_WORKERS 4     /* upper limit of the batches in flight (LNSFAID_COMB_WORKERS, default 2): the transfers of one overlap
                               * the decode of the others */
#define LF_COMB_MIN_MEMBERS 4 /* fewer one-group contexts than this on a device: each keeps to its own stream (measured faster) */
enum { LF_SLOT_FREE = 0, LF_SLOT_IDLE, LF_SLOT_PENDING, LF_SLOT_RUNNING, LF_SLOT_DONE };
struct LfSlot {
    int state = LF_SLOT_FREE;
    int rc = 0;
    LfDevCfg cfg;
    lnsfaid_group_stats stats;
};
struct LfCombiner {
    int device = 0;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::thread worker[LF_COMB_WORKERS];
    bool stop = false, dead = false;
    bool gathering = false;                        /* one worker at a time collects a batch */
    int running = 0;                               /* calls claimed by a worker and not yet answered */
    lnsfaid_ctx* pool[LF_COMB_WORKERS] = {};       /* one per worker, created for its first batch */
    LfDevCode code;              /* what every member decodes (incl. the bit-flipping column list) */
    size_t group_bytes = 0;
    int8_t *h_in = nullptr, *h_out = nullptr; /* pinned, LF_COMB_SLOTS groups each; allocated when the second member joins */
    int8_t *d_in_map = nullptr, *d_out_map = nullptr; /* the same memory as the device sees it (zero copy), null if not mapped */
    int32_t* h_status = nullptr;              /* pinned, LF_COMB_WORKERS x LF_COMB_SLOTS * 32 words */
    lnsfaid_group_stats* h_stats = nullptr;   /* pinned, LF_COMB_WORKERS x LF_COMB_SLOTS */
    LfSlot slots[LF_COMB_SLOTS];
    int members = 0, pending = 0;
    char err[256] = "";              /* lnsfaid_last_hip_error text of the last failed batch (the worker's is thread-local) */
    uint64_t batches = 0, calls = 0; /* statistics (LNSFAID_TRACE at shutdown) */
    double gather_ms = 0, device_ms = 0;
};
static std::mutex g_comb_mutex;
static LfCombiner* g_comb[LF_COMB_DEVICES] = {};

static int comb_workers()
{
    static const int n = [] {
        const char* e = getenv("LNSFAID_COMB_WORKERS");
        const int v = e ? atoi(e) : 2;
        return v < 1 ? 1 : (v > LF_COMB_WORKERS ? LF_COMB_WORKERS : v);
    }();
    return n;
}

/* into how many batches the members' calls are cut (>= the number of workers; LNSFAID_COMB_BATCHES): more, smaller batches than
 * workers keep the launches out of phase - a worker that comes back from the device finds the next batch waiting */
static int comb_batches()
{
    static const int v = [] {
        const char* e = getenv("LNSFAID_COMB_BATCHES");
        const int n = e ? atoi(e) : 0;
        return n > 64 ? 64 : n;
    }();
    return v > comb_workers() ? v : comb_workers();
}

static bool comb_enabled()
{
    static const char* e = getenv("LNSFAID_COALESCE");
    return !(e && e[0] == '0');
}

static void apply_devcfg(lnsfaid_ctx* ctx, const LfDevCfg& n); /* pool only: no validation, same code */

static void comb_run_batch(LfCombiner* cb, int w, const int* batch, int nb)
{
    lnsfaid_ctx* P = cb->pool[w];
    int32_t* h_status = cb->h_status + (size_t)w * LF_COMB_SLOTS * LNSFAID_GROUP;
    lnsfaid_group_stats* h_stats = cb->h_stats + (size_t)w * LF_COMB_SLOTS;
    int rc = LNSFAID_OK;
    auto fail = [&](hipError_t e, const char* what) { if (rc == LNSFAID_OK && e != hipSuccess) rc = hip_fail(e, what); };
    int lo = LF_COMB_SLOTS, hi = -1;
    for (int i = 0; i < nb; ++i) { lo = batch[i] < lo ? batch[i] : lo; hi = batch[i] > hi ? batch[i] : hi; }
    const size_t n = (size_t)hi + 1;
    for (size_t g = 0; g < n; ++g) { /* groups without a call in this batch: finished before they start */
        bool active = false;
        for (int i = 0; i < nb; ++i) active = active || (size_t)batch[i] == g;
        for (int l = 0; l < LNSFAID_GROUP; ++l) h_status[g * LNSFAID_GROUP + l] = active ? 0 : LF_DONE;
    }
    fail(hipSetDevice(cb->device), "hipSetDevice");
    if (rc == LNSFAID_OK && memcmp(&P->hcfg, &cb->slots[batch[0]].cfg, sizeof(LfDevCfg)) != 0) apply_devcfg(P, cb->slots[batch[0]].cfg);
    const size_t gb = cb->group_bytes, span = (size_t)(hi - lo + 1);
    fail(hipMemcpyAsync(P->d_status[0], h_status, n * LNSFAID_GROUP * sizeof(int32_t), hipMemcpyHostToDevice, P->stream), "status upload");
    /* The staging area is pinned, device-mapped host memory.  zero copy: the kernel reads every LLR once (input staging) and
     * writes every decision once, so it can do both straight over PCIe - no separate copy phases in front of and behind the
     * launch, and the transfers of one batch run under the compute of the other.  LNSFAID_COMB_COPY=1: explicit copies. */
    static const bool copy_mode = getenv("LNSFAID_COMB_COPY") != nullptr;
    const int8_t* k_in = cb->d_in_map;
    int8_t* k_out = cb->d_out_map;
    if (copy_mode || !k_in || !k_out) {
        fail(hipMemcpyAsync(P->d_io_in + (size_t)lo * gb, cb->h_in + (size_t)lo * gb, span * gb, hipMemcpyHostToDevice, P->stream), "fixInput upload");
        k_in = P->d_io_in; k_out = P->d_io_out;
    }
    if (rc == LNSFAID_OK) rc = decode_device_impl(P, k_in, n, k_out, P->d_io_stats, true);
    if (rc == LNSFAID_OK) {
        if (k_out == P->d_io_out)
            fail(hipMemcpyAsync(cb->h_out + (size_t)lo * gb, P->d_io_out + (size_t)lo * gb, span * gb, hipMemcpyDeviceToHost, P->stream), "decodedBits download");
        fail(hipMemcpyAsync(h_stats + lo, P->d_io_stats + lo, span * sizeof(lnsfaid_group_stats), hipMemcpyDeviceToHost, P->stream), "stats download");
        if (rc == LNSFAID_OK) rc = stream_wait(P);
    }
    std::lock_guard<std::mutex> lk(cb->m);
    if (rc != LNSFAID_OK) snprintf(cb->err, sizeof(cb->err), "%s", g_hip_err);
    for (int i = 0; i < nb; ++i) {
        LfSlot& s = cb->slots[batch[i]];
        s.rc = rc;
        s.stats = h_stats[batch[i]];
        s.state = LF_SLOT_DONE;
    }
    cb->running -= nb;
    cb->batches += 1; cb->calls += (uint64_t)nb;
    cb->cv_done.notify_all();
    cb->cv_work.notify_all(); /* a gathering worker counts the calls in flight */
}

static void comb_worker(LfCombiner* cb, int w)
{
    std::unique_lock<std::mutex> lk(cb->m);
    for (;;) {
        cb->cv_work.wait(lk, [&] { return cb->stop || (cb->pending > 0 && !cb->gathering); });
        if (cb->stop) break;
        /* gather.  The decode time of a batch hardly depends on its size (a launch of 64 groups just fills the chip) and
         * LF_COMB_WORKERS batches are in flight at a time, so a batch should be a 1 / LF_COMB_WORKERS share of the members: wait
         * until that many calls are pending, or every member is accounted for (pending, or in flight with another worker), or no
         * further call has arrived for LF_COMB_QUIET_US, or LF_COMB_WINDOW_US have passed since the first one was seen */
        cb->gathering = true;
        const auto t_first = std::chrono::steady_clock::now();
        const auto deadline = t_first + std::chrono::microseconds(LF_COMB_WINDOW_US);
        for (;;) {
            const int share = (cb->members + comb_batches() - 1) / comb_batches();
            if (cb->stop || cb->pending >= share || cb->pending + cb->running >= cb->members) break;
            const int seen = cb->pending, seen_running = cb->running;
            auto quiet = std::chrono::steady_clock::now() + std::chrono::microseconds(LF_COMB_QUIET_US);
            if (quiet > deadline) quiet = deadline;
            cb->cv_work.wait_until(lk, quiet, [&] { return cb->pending != seen || cb->running != seen_running || cb->stop; });
            if (cb->pending == seen && cb->running == seen_running) break; /* quiet (or the window is over) */
        }
        if (cb->stop) { cb->gathering = false; break; }
        cb->gather_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_first).count();
        int batch[LF_COMB_SLOTS], nb = 0;
        const int cap = cb->members > 8 * comb_workers() ? (cb->members + comb_batches() - 1) / comb_batches() : LF_COMB_SLOTS;
        const LfDevCfg* cfg = nullptr; /* one configuration per batch: the first pending one's */
        for (int i = 0; i < LF_COMB_SLOTS && nb < cap; ++i) {
            LfSlot& s = cb->slots[i];
            if (s.state != LF_SLOT_PENDING) continue;
            if (!cfg) cfg = &s.cfg;
            if (memcmp(cfg, &s.cfg, sizeof(LfDevCfg)) != 0) continue; /* next batch */
            s.state = LF_SLOT_RUNNING;
            batch[nb++] = i;
        }
        cb->pending -= nb;
        cb->running += nb;
        cb->gathering = false;
        cb->cv_work.notify_all(); /* the next batch may be gathered by another worker while this one is on the device */
        if (nb == 0) continue;
        if (!cb->pool[w]) { /* this worker's first batch: its pool context (device state for LF_COMB_SLOTS groups) */
            lk.unlock();
            int rc = LNSFAID_OK;
            lnsfaid_ctx* P = new (std::nothrow) lnsfaid_ctx();
            if (!P) rc = LNSFAID_E_NOMEM;
            if (!rc) {
                P->device = cb->device; P->max_groups = LF_COMB_SLOTS;
                P->hcode = cb->code; P->hcfg = cb->slots[batch[0]].cfg;
                g_live_contexts.fetch_add(1, std::memory_order_relaxed);
                rc = alloc_state(P);
                if (!rc) rc = ensure_io(P);
                if (rc) { lnsfaid_destroy(P); P = nullptr; }
            }
            lk.lock();
            if (rc) { /* the members fall back to their own contexts from now on */
                cb->dead = true;
                for (int i = 0; i < nb; ++i) { cb->slots[batch[i]].rc = rc; cb->slots[batch[i]].state = LF_SLOT_DONE; }
                cb->running -= nb;
                cb->cv_done.notify_all();
                continue;
            }
            cb->pool[w] = P;
        }
        lk.unlock();
        const auto t_dev = std::chrono::steady_clock::now();
        comb_run_batch(cb, w, batch, nb);
        lk.lock();
        cb->device_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dev).count();
    }
}

/* a one-group context joins the combiner of its device (created on demand); -1: it stays on its own */
static int comb_join(lnsfaid_ctx* ctx)
{
    if (!comb_enabled() || ctx->max_groups != 1 || ctx->device < 0 || ctx->device >= LF_COMB_DEVICES) return -1;
    std::lock_guard<std::mutex> g(g_comb_mutex);
    LfCombiner* cb = g_comb[ctx->device];
    if (!cb) { /* the first one-group context of the device: a record only - staging area and worker come with the second */
        cb = new (std::nothrow) LfCombiner();
        if (!cb) return -1;
        cb->device = ctx->device;
        cb->code = ctx->hcode;
        cb->group_bytes = (size_t)LNSFAID_GROUP * (size_t)ctx->n_var;
        g_comb[ctx->device] = cb;
    }
    std::lock_guard<std::mutex> lk(cb->m);
    if (cb->dead || memcmp(&cb->code, &ctx->hcode, sizeof(LfDevCode)) != 0) return -1; /* another code: on its own */
    if (cb->members >= 1 && !cb->h_in) { /* there is something to combine from now on */
        const size_t bytes = (size_t)LF_COMB_SLOTS * cb->group_bytes;
        if (hipSetDevice(cb->device) != hipSuccess
            || hipHostMalloc((void**)&cb->h_in, bytes, hipHostMallocMapped) != hipSuccess
            || hipHostMalloc((void**)&cb->h_out, bytes, hipHostMallocMapped) != hipSuccess
            || hipHostMalloc((void**)&cb->h_status, (size_t)LF_COMB_WORKERS * LF_COMB_SLOTS * LNSFAID_GROUP * sizeof(int32_t), hipHostMallocDefault) != hipSuccess
            || hipHostMalloc((void**)&cb->h_stats, (size_t)LF_COMB_WORKERS * LF_COMB_SLOTS * sizeof(lnsfaid_group_stats), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            if (cb->h_in) (void)hipHostFree(cb->h_in);
            if (cb->h_out) (void)hipHostFree(cb->h_out);
            if (cb->h_status) (void)hipHostFree(cb->h_status);
            if (cb->h_stats) (void)hipHostFree(cb->h_stats);
            cb->h_in = cb->h_out = nullptr; cb->h_status = nullptr; cb->h_stats = nullptr;
            cb->dead = true; /* everybody stays on the direct path */
            return -1;
        }
        if (hipHostGetDevicePointer((void**)&cb->d_in_map, cb->h_in, 0) != hipSuccess
            || hipHostGetDevicePointer((void**)&cb->d_out_map, cb->h_out, 0) != hipSuccess) {
            (void)hipGetLastError();
            cb->d_in_map = cb->d_out_map = nullptr; /* explicit copies then */
        }
        for (int w = 0; w < comb_workers(); ++w) cb->worker[w] = std::thread(comb_worker, cb, w);
    }
    for (int i = 0; i < LF_COMB_SLOTS; ++i)
        if (cb->slots[i].state == LF_SLOT_FREE) { cb->slots[i].state = LF_SLOT_IDLE; cb->members += 1; ctx->comb = cb; return i; }
    return -1; /* more contexts than slots */
}

static void comb_leave(lnsfaid_ctx* ctx, int slot)
{
    LfCombiner* cb = nullptr;
    {
        std::lock_guard<std::mutex> g(g_comb_mutex);
        cb = ctx->comb;
        if (!cb) return;
        bool last;
        {
            std::lock_guard<std::mutex> lk(cb->m);
            cb->slots[slot].state = LF_SLOT_FREE;
            cb->members -= 1;
            last = cb->members == 0;
            if (last) cb->stop = true;
            cb->cv_work.notify_all();
        }
        if (!last) return;
        if (g_comb[cb->device] == cb) g_comb[cb->device] = nullptr; /* a later context starts a new one */
    }
    for (auto& th : cb->worker) if (th.joinable()) th.join();
    static const bool trace = getenv("LNSFAID_TRACE") != nullptr;
    if (trace && cb->batches)
        fprintf(stderr, "[lnsfaid] call combiner of device %d: %llu calls in %llu batches (%.1f per batch); per batch %.3f ms gathering, %.3f ms copies + decode\n",
                cb->device, (unsigned long long)cb->calls, (unsigned long long)cb->batches, (double)cb->calls / (double)cb->batches,
                cb->gather_ms / (double)cb->batches, cb->device_ms / (double)cb->batches);
    for (auto P : cb->pool) if (P) lnsfaid_destroy(P);
    if (cb->h_in) { (void)hipHostFree(cb->h_in); (void)hipHostFree(cb->h_out); (void)hipHostFree(cb->h_status); (void)hipHostFree(cb->h_stats); }
    delete cb;
}

/* 1: decoded through the combiner (*rc_out = result); 0: not applicable now, take the direct path */
static int comb_decode(lnsfaid_ctx* ctx, const int8_t* fixInput, int8_t* decodedBits, lnsfaid_group_stats* stats, int* rc_out)
{
    if (ctx->comb_slot < 0 || ctx->rows_per_lane != 0 || ctx->msg_store != 0 || ctx->waves_per_cw != 0) return 0;
    LfCombiner* cb = ctx->comb;
    const int slot = ctx->comb_slot;
    {
        std::lock_guard<std::mutex> lk(cb->m);
        if (cb->dead || cb->members < LF_COMB_MIN_MEMBERS || !cb->h_in) return 0; /* too few to gain from combining: direct path */
        if (memcmp(&cb->code, &ctx->hcode, sizeof(LfDevCode)) != 0) return 0; /* lnsfaid_set_cfg changed REGULAR_COL_WEIGHT */
    }
    memcpy(cb->h_in + (size_t)slot * cb->group_bytes, fixInput, cb->group_bytes); /* every caller copies its own slot, in parallel */
    int rc;
    {
        std::unique_lock<std::mutex> lk(cb->m);
        LfSlot& s = cb->slots[slot];
        s.cfg = ctx->hcfg;
        s.state = LF_SLOT_PENDING;
        cb->pending += 1;
        cb->cv_work.notify_all(); /* (at most LF_COMB_WORKERS waiters) */
        if (!cb->cv_done.wait_for(lk, std::chrono::seconds(120), [&] { return s.state == LF_SLOT_DONE; })) {
            snprintf(g_hip_err, sizeof(g_hip_err), "call combiner: no result after 120 s");
            *rc_out = LNSFAID_E_INTERNAL;
            return 1;
        }
        rc = s.rc;
        if (rc != LNSFAID_OK) snprintf(g_hip_err, sizeof(g_hip_err), "%s", cb->err);
        if (stats) *stats = s.stats;
        s.state = LF_SLOT_IDLE;
    }
    if (rc == LNSFAID_OK) memcpy(decodedBits, cb->h_out + (size_t)slot * cb->group_bytes, cb->group_bytes);
    *rc_out = rc;
    return 1;
}


extern "C" int lnsfaid_frontend_set_interleave(lnsfaid_ctx* ctx, int32_t interleave_mod_type)
{
    if (!ctx || interleave_mod_type < 1 || ctx->n_var % interleave_mod_type != 0) return LNSFAID_E_INVAL;
    ctx->fe_interleave = interleave_mod_type;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_frontend_set_frames(lnsfaid_ctx* ctx, const int8_t* outputBits, const int8_t* inputBits, size_t n_streams)
{
    if (!ctx || n_streams > ctx->max_groups) return LNSFAID_E_INVAL;
    if (!outputBits || n_streams == 0) { ctx->fe_frames_streams = 0; return LNSFAID_OK; } /* back to one codeword for all */
    if (!inputBits) return LNSFAID_E_INVAL;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->d_fe_frames) {
        HIP_TRY(hipMalloc(&ctx->d_fe_frames, ctx->max_groups * LNSFAID_GROUP * (size_t)ctx->n_var));
        HIP_TRY(hipMalloc(&ctx->d_fe_input, ctx->max_groups * LNSFAID_GROUP * (size_t)ctx->k_info));
    }
    HIP_TRY(hipMemcpyAsync(ctx->d_fe_frames, outputBits, n_streams * LNSFAID_GROUP * (size_t)ctx->n_var, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->d_fe_input, inputBits, n_streams * LNSFAID_GROUP * (size_t)ctx->k_info, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->fe_frames_streams = n_streams;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_frontend_input_bits(lnsfaid_ctx* ctx, const int8_t** d_inputBits)
{
    if (!ctx || !d_inputBits) return LNSFAID_E_INVAL;
    *d_inputBits = ctx->fe_frames_streams ? ctx->d_fe_input : nullptr;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_io_buffers(lnsfaid_ctx* ctx, int8_t** d_fixInput, int8_t** d_decodedBits, lnsfaid_group_stats** d_stats)
{
    if (!ctx) return LNSFAID_E_INVAL;
    HIP_TRY(hipSetDevice(ctx->device));
    const int rc = ensure_io(ctx);
    if (rc) return rc;
    if (d_fixInput) *d_fixInput = ctx->d_io_in;
    if (d_decodedBits) *d_decodedBits = ctx->d_io_out;
    if (d_stats) *d_stats = ctx->d_io_stats;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_read_stats(lnsfaid_ctx* ctx, lnsfaid_group_stats* stats, size_t n_groups)
{
    if (!ctx || (n_groups && !stats) || n_groups > ctx->max_groups) return LNSFAID_E_INVAL;
    if (n_groups == 0) return LNSFAID_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const int rc = ensure_io(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stats, ctx->d_io_stats, n_groups * sizeof(lnsfaid_group_stats), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return LNSFAID_OK;
}

static bool host_pinned(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; } /* plain malloc memory */
    return at.type == hipMemoryTypeHost;
}

extern "C" int lnsfaid_decode(lnsfaid_ctx* ctx, const int8_t* fixInput, size_t n_groups, int8_t* decodedBits,
                              lnsfaid_group_stats* stats)
{
    if (!ctx || (n_groups && (!fixInput || !decodedBits))) return LNSFAID_E_INVAL;
    if (n_groups > ctx->max_groups) return LNSFAID_E_INVAL;
    if (n_groups == 0) return LNSFAID_OK;
    {   /* a one-group context among others on its device: through the call combiner */
        int rcc = 0;
        if (comb_decode(ctx, fixInput, decodedBits, stats, &rcc)) return rcc;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_io(ctx);
    if (rc) return rc;
    const size_t group_bytes = LNSFAID_GROUP * (size_t)ctx->n_var;
    const size_t bytes = n_groups * group_bytes;
    /* Pinned host buffers (hipHostMalloc / lnsfaid_host_register): cut the batch into pieces of whole groups and overlap
     * the copy-in of piece c + 1 and the copy-out of piece c - 1 with the decode of piece c (groups are independent, and
     * a piece is decoded to completion before the next one starts, so the pieces share the per-codeword state buffers).
     * Pageable buffers cannot overlap (the runtime stages them synchronously): one copy in, one decode, one copy out. */
    size_t chunk = ((n_groups + LF_IO_CHUNKS - 1) / LF_IO_CHUNKS + 63) / 64 * 64; /* >= one full wave of workgroups */
    if (chunk >= n_groups || !host_pinned(fixInput) || !host_pinned(decodedBits)) { /* (small batches: no pointer query at all) */
        HIP_TRY(hipMemcpyAsync(ctx->d_io_in, fixInput, bytes, hipMemcpyHostToDevice, ctx->stream));
        rc = lnsfaid_decode_device(ctx, ctx->d_io_in, n_groups, ctx->d_io_out, ctx->d_io_stats);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(decodedBits, ctx->d_io_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        if (!ctx->s_in) {
            HIP_TRY(hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking));
            for (auto& e : ctx->ev_in) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const size_t n_chunks = (n_groups + chunk - 1) / chunk; /* <= LF_IO_CHUNKS */
        for (size_t c = 0; c < n_chunks; ++c) { /* all copies in are queued at once, one event per piece */
            const size_t g0 = c * chunk, ng = g0 + chunk <= n_groups ? chunk : n_groups - g0;
            HIP_TRY(hipMemcpyAsync(ctx->d_io_in + g0 * group_bytes, fixInput + g0 * group_bytes, ng * group_bytes, hipMemcpyHostToDevice, ctx->s_in));
            HIP_TRY(hipEventRecord(ctx->ev_in[c], ctx->s_in));
        }
        for (size_t c = 0; c < n_chunks; ++c) {
            const size_t g0 = c * chunk, ng = g0 + chunk <= n_groups ? chunk : n_groups - g0;
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_in[c], 0));
            rc = lnsfaid_decode_device(ctx, ctx->d_io_in + g0 * group_bytes, ng, ctx->d_io_out + g0 * group_bytes, ctx->d_io_stats + g0);
            if (rc) { (void)hipStreamSynchronize(ctx->s_in); (void)hipStreamSynchronize(ctx->s_out); return rc; }
            /* decode_device returns with the piece finished: its copy out needs no further ordering */
            HIP_TRY(hipMemcpyAsync(decodedBits + g0 * group_bytes, ctx->d_io_out + g0 * group_bytes, ng * group_bytes, hipMemcpyDeviceToHost, ctx->s_out));
        }
        HIP_TRY(hipStreamSynchronize(ctx->s_out));
    }
    if (stats)
        HIP_TRY(hipMemcpyAsync(stats, ctx->d_io_stats, n_groups * sizeof(lnsfaid_group_stats), hipMemcpyDeviceToHost,
                               ctx->stream));
    { const int rcw = stream_wait(ctx); if (rcw) return rcw; }
    return LNSFAID_OK;
}

/* ---- pinned host memory for callers without a HIP toolchain (the reference is plain C++) ---------------- */
extern "C" int lnsfaid_host_register(void* ptr, size_t bytes)
{
    if (!ptr || !bytes) return LNSFAID_E_INVAL;
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable)); /* valid on every device of the process */
    return LNSFAID_OK;
}

extern "C" int lnsfaid_host_unregister(void* ptr)
{
    if (!ptr) return LNSFAID_E_INVAL;
    HIP_TRY(hipHostUnregister(ptr));
    return LNSFAID_OK;
}

extern "C" int lnsfaid_count_errors_device(lnsfaid_ctx* ctx, const int8_t* d_decodedBits, const int8_t* d_inputBits,
                                           size_t n_groups, uint64_t out[4])
{
    if (!ctx || !out || (n_groups && !d_decodedBits)) return LNSFAID_E_INVAL;
    if (n_groups == 0) return LNSFAID_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(lf_launch_count_errors(d_decodedBits, d_inputBits, ctx->n_var, ctx->k_info, n_groups * LNSFAID_GROUP,
                                   ctx->d_counters, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           ctx->stream));
    { const int rcw = stream_wait(ctx); if (rcw) return rcw; }
    for (int i = 0; i < 4; ++i) out[i] += ctx->h_counters[i];
    return LNSFAID_OK;
}

extern "C" int lnsfaid_count_errors(lnsfaid_ctx* ctx, const int8_t* decodedBits, const int8_t* inputBits, size_t n_groups,
                                    uint64_t out[4])
{
    if (!ctx || !out || (n_groups && !decodedBits)) return LNSFAID_E_INVAL;
    if (n_groups > ctx->max_groups) return LNSFAID_E_INVAL;
    if (n_groups == 0) return LNSFAID_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_io(ctx);
    if (rc) return rc;
    const size_t n_cw = n_groups * LNSFAID_GROUP;
    HIP_TRY(hipMemcpyAsync(ctx->d_io_out, decodedBits, n_cw * (size_t)ctx->n_var, hipMemcpyHostToDevice, ctx->stream));
    const int8_t* d_in = nullptr;
    if (inputBits) {
        /* [32][K] per group packed back to back fits in the (larger) fixInput staging buffer */
        HIP_TRY(hipMemcpyAsync(ctx->d_io_in, inputBits, n_cw * (size_t)ctx->k_info, hipMemcpyHostToDevice, ctx->stream));
        d_in = ctx->d_io_in;
    }
    return lnsfaid_count_errors_device(ctx, ctx->d_io_out, d_in, n_groups, out);
}

/* ---- front-end on the device ------------------------------------------------------------------------ */
extern "C" uint64_t lnsfaid_frontend_draws_per_group(const lnsfaid_ctx* ctx, int32_t mod_type)
{
    if (!ctx || (mod_type != 2 && mod_type != 4 && mod_type != 6 && mod_type != 8)) return 0;
    /* 32 * n_var / mod_type symbols, 2 normals per symbol, 2 uniforms per normal */
    return (uint64_t)32 * (uint64_t)ctx->n_var / (uint64_t)mod_type * 4u;
}

extern "C" int lnsfaid_frontend_device_states(lnsfaid_ctx* ctx, const uint32_t* states, const uint64_t* draws_before, size_t n_streams,
                                              int32_t mod_type, float sigma, float scale, const int8_t* codeword, int8_t* d_fixInput)
{
    if (!ctx || !states || !draws_before || !d_fixInput || (mod_type != 2 && mod_type != 4 && mod_type != 6 && mod_type != 8)) return LNSFAID_E_INVAL;
    if (n_streams == 0) return LNSFAID_OK;
    if (n_streams > ctx->max_groups || (32L * ctx->n_var) % mod_type != 0) return LNSFAID_E_INVAL;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->d_fe_seeds) {
        HIP_TRY(hipMalloc(&ctx->d_fe_seeds, ctx->max_groups * 3 * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&ctx->d_fe_draws, ctx->max_groups * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc(&ctx->d_fe_codeword, (size_t)ctx->n_var));
    }
    HIP_TRY(hipMemcpyAsync(ctx->d_fe_seeds, states, n_streams * 3 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->d_fe_draws, draws_before, n_streams * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (codeword) HIP_TRY(hipMemcpyAsync(ctx->d_fe_codeword, codeword, (size_t)ctx->n_var, hipMemcpyHostToDevice, ctx->stream));
    /* AWGNChannel(ModSeq, sigma / sqrt(2)): float / double -> double, narrowed to float (CSimulate.cpp:126) */
    const float sigma_ch = (float)((double)sigma / 1.4142135623730951);
    HIP_TRY(lf_launch_frontend(ctx->d_fe_seeds, ctx->d_fe_draws, (int)n_streams, mod_type, sigma_ch, scale,
                               codeword ? ctx->d_fe_codeword : nullptr,
                               (!codeword && ctx->fe_frames_streams >= n_streams) ? ctx->d_fe_frames : nullptr, ctx->n_var,
                               ctx->n_check, ctx->fe_interleave, ctx->fe_exact ? 0 : 1, d_fixInput, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* states / draws_before may be reused by the caller */
    return LNSFAID_OK;
}

extern "C" int lnsfaid_frontend_set_exact(lnsfaid_ctx* ctx, int32_t exact)
{
    if (!ctx || exact < 0 || exact > 1) return LNSFAID_E_INVAL;
    ctx->fe_exact = exact;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_frontend_fastpath_bounds(lnsfaid_ctx* ctx, double measured[2], double assumed[2])
{
    if (!ctx || !measured || !assumed) return LNSFAID_E_INVAL;
    HIP_TRY(hipSetDevice(ctx->device));
    double* d = nullptr;
    HIP_TRY(hipMalloc(&d, 2 * sizeof(double)));
    hipError_t e = lf_frontend_fastpath_scan(d, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(measured, d, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    HIP_TRY(e);
    lf_frontend_fastpath_assumed(assumed);
    return LNSFAID_OK;
}

extern "C" int lnsfaid_frontend_device(lnsfaid_ctx* ctx, const uint32_t* seeds, const uint64_t* draws_before, size_t n_streams,
                                       int32_t mod_type, float sigma, float scale, const int8_t* codeword, int8_t* d_fixInput)
{
    if (!ctx || !seeds || !draws_before || !d_fixInput) return LNSFAID_E_INVAL;
    if (n_streams > ctx->max_groups) return LNSFAID_E_INVAL; /* before anything is sized by it */
    if (n_streams == 0) return LNSFAID_OK;
    std::vector<uint32_t> st; /* CChannel::Initial without CONTINUE_SEED: IX = IY = IZ = seed (CChannel.cpp:121) */
    try {
        st.resize(3 * n_streams);
    } catch (...) {
        return LNSFAID_E_NOMEM; /* nothing is thrown across the C boundary */
    }
    for (size_t i = 0; i < n_streams; ++i) st[3 * i] = st[3 * i + 1] = st[3 * i + 2] = seeds[i];
    return lnsfaid_frontend_device_states(ctx, st.data(), draws_before, n_streams, mod_type, sigma, scale, codeword, d_fixInput);
}

/* ---- multi-GPU: one process (or host thread) per GPU, the four counters summed over RCCL ---------------------------
 * The reference sums its workers' counters on the main thread after pthread_join (main.cpp:174-182); with one context per
 * GPU the same sum is ONE all-reduce of 4 x uint64 on the context's stream.  RCCL is bound at run time - first the copy
 * that is already in the process (a PyTorch process carries its own), then the system library - so that the decode library
 * has no link-time dependency on it and a communicator handed in by the caller belongs to the same RCCL instance. */
struct LfRccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
static LfRccl* lf_rccl()
{
    static LfRccl r;
    static bool tried = false;
    if (tried) return r.ok ? &r : nullptr;
    tried = true;
    void* h = RTLD_DEFAULT;
    if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { snprintf(g_hip_err, sizeof(g_hip_err), "RCCL not found: %s", dlerror()); return nullptr; }
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
    if (!r.ok) snprintf(g_hip_err, sizeof(g_hip_err), "RCCL symbols missing");
    return r.ok ? &r : nullptr;
}
static int rccl_fail(LfRccl* r, ncclResult_t e, const char* what)
{
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, r && r->GetErrorString ? r->GetErrorString(e) : "RCCL error");
    return LNSFAID_E_HIP;
}

extern "C" int lnsfaid_comm_unique_id(uint8_t id[LNSFAID_COMM_ID_BYTES])
{
    if (!id) return LNSFAID_E_INVAL;
    static_assert(LNSFAID_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    LfRccl* r = lf_rccl();
    if (!r) return LNSFAID_E_NODEVICE;
    ncclUniqueId u;
    const ncclResult_t e = r->GetUniqueId(&u);
    if (e != ncclSuccess) return rccl_fail(r, e, "ncclGetUniqueId");
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return LNSFAID_OK;
}

extern "C" int lnsfaid_comm_destroy(lnsfaid_ctx* ctx)
{
    if (!ctx) return LNSFAID_E_INVAL;
    if (ctx->comm && ctx->comm_owned) {
        LfRccl* r = lf_rccl();
        if (r) { (void)hipSetDevice(ctx->device); (void)r->CommDestroy((ncclComm_t)ctx->comm); }
    }
    ctx->comm = nullptr; ctx->comm_owned = false;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_comm_init(lnsfaid_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t id[LNSFAID_COMM_ID_BYTES])
{
    if (!ctx || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return LNSFAID_E_INVAL;
    LfRccl* r = lf_rccl();
    if (!r) return LNSFAID_E_NODEVICE;
    (void)lnsfaid_comm_destroy(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t e = r->CommInitRank(&c, n_ranks, u, rank);
    if (e != ncclSuccess) return rccl_fail(r, e, "ncclCommInitRank");
    ctx->comm = (void*)c; ctx->comm_owned = true;
    return LNSFAID_OK;
}

extern "C" int lnsfaid_comm_attach(lnsfaid_ctx* ctx, void* nccl_comm)
{
    if (!ctx) return LNSFAID_E_INVAL;
    (void)lnsfaid_comm_destroy(ctx);
    ctx->comm = nccl_comm; ctx->comm_owned = false; /* the caller keeps ownership */
    return LNSFAID_OK;
}

extern "C" int lnsfaid_allreduce_counters(lnsfaid_ctx* ctx, uint64_t counters[4])
{
    if (!ctx || !counters) return LNSFAID_E_INVAL;
    if (!ctx->comm) return LNSFAID_E_INVAL; /* lnsfaid_comm_init / lnsfaid_comm_attach first */
    LfRccl* r = lf_rccl();
    if (!r) return LNSFAID_E_NODEVICE;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->d_reduce) HIP_TRY(hipMalloc(&ctx->d_reduce, 4 * sizeof(unsigned long long)));
    for (int i = 0; i < 4; ++i) ctx->h_counters[i] = counters[i];
    HIP_TRY(hipMemcpyAsync(ctx->d_reduce, ctx->h_counters, 4 * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
    const ncclResult_t e = r->AllReduce(ctx->d_reduce, ctx->d_reduce, 4, ncclUint64, ncclSum, (ncclComm_t)ctx->comm, ctx->stream);
    if (e != ncclSuccess) return rccl_fail(r, e, "ncclAllReduce");
    HIP_TRY(hipMemcpyAsync(ctx->h_counters, ctx->d_reduce, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; ++i) counters[i] = ctx->h_counters[i];
    return LNSFAID_OK;
}

/* ---- measurement hooks / misc ---------------------------------------------------------------------- */
extern "C" int lnsfaid_kernel_time(lnsfaid_ctx* ctx, double* out_ms, uint64_t* out_launches, int32_t reset)
{
    if (!ctx) return LNSFAID_E_INVAL;
    if (out_ms) *out_ms = ctx->kernel_ms;
    if (out_launches) *out_launches = ctx->kernel_launches;
    if (reset) { ctx->kernel_ms = 0.0; ctx->kernel_launches = 0; }
    return LNSFAID_OK;
}

extern "C" void* lnsfaid_stream(lnsfaid_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" const char* lnsfaid_strerror(int err)
{
    switch (err) {
    case LNSFAID_OK: return "ok";
    case LNSFAID_E_INVAL: return "invalid argument or unsupported configuration";
    case LNSFAID_E_CODE: return "code table is not a supported quasi-cyclic code (Z = 256, degree 2..24, distinct block columns per block row)";
    case LNSFAID_E_NOMEM: return "out of memory";
    case LNSFAID_E_HIP: return "HIP runtime error";
    case LNSFAID_E_NODEVICE: return "no usable GPU";
    case LNSFAID_E_INTERNAL: return "internal error: relaunch loop did not converge";
    default: return "unknown error";
    }
}

extern "C" const char* lnsfaid_last_hip_error(void) { return g_hip_err; }
#ifndef LNSFAID_EXTRA_FLAGS
#define LNSFAID_EXTRA_FLAGS ""
#endif
/* the experiment switches of the build (Makefile EXTRA) are part of the version: "lnsfaid-amd 0.3 (gfx950)" is the default build */
extern "C" const char* lnsfaid_version(void)
{
    return sizeof(LNSFAID_EXTRA_FLAGS) > 1 ? "lnsfaid-amd 0.3 (gfx950) [" LNSFAID_EXTRA_FLAGS "]" : "lnsfaid-amd 0.3 (gfx950)";
}
