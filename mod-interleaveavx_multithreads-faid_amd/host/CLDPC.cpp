#include "CLDPC.h"

#include <fstream>
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <cstring>

static void die(const char* what, int rc)
{
    /* the reference's fatal path is exit(EXIT_FAILURE) (CTool.cpp:591-596); errors never pass silently */
    fprintf(stderr, "%s failed: %d (%s; %s)\n", what, rc, lnsfaid_strerror(rc), lnsfaid_last_hip_error());
    exit(EXIT_FAILURE);
}

CLDPC::CLDPC()
    : m_Rate(0), inputBits(nullptr), outputBits(nullptr), decodedBits(nullptr), fixInput(nullptr), nb_iteration(0),
      m_M(0), m_N(0), m_K(0), m_PunLen(0), m_ShortenLen(0), m_frame(0), m_groups(0), m_stats(nullptr), m_device(0),
      m_factor_1(1), m_factor_2(6), m_device_io(false)
{
    memset(m_ctx, 0, sizeof(m_ctx));
}

CLDPC::~CLDPC()
{
    for (auto& c : m_ctx) lnsfaid_destroy(c);
    if (fixInput) (void)lnsfaid_host_unregister(fixInput);       /* harmless error when it was never registered */
    if (decodedBits) (void)lnsfaid_host_unregister(decodedBits);
    free(inputBits); free(outputBits); free(decodedBits); free(fixInput); free(m_stats);
}

void CLDPC::Initial(int nb_frame, int MaxItertion, int groups, int device)
{
    if (nb_frame != LNSFAID_GROUP) { fprintf(stderr, "noFrames must be 32 (one group = 32 codewords)\n"); exit(EXIT_FAILURE); }
    m_M = _NoCheck; m_K = NmoinsK; m_N = _NoVar; m_PunLen = _PunctureBits; m_ShortenLen = _ShortenBits;
    m_Rate = 0.8444444; /* reference CLDPC.cpp:4780 */
    m_frame = nb_frame; nb_iteration = MaxItertion; m_groups = groups; m_device = device;
    const size_t frames = (size_t)groups * nb_frame;
    inputBits = (int8_t*)calloc(frames * m_K, 1);
    outputBits = (int8_t*)calloc(frames * m_N, 1);
    decodedBits = (int8_t*)calloc(frames * m_N, 1);
    fixInput = (int8_t*)calloc(frames * m_N, 1);
    m_stats = (lnsfaid_group_stats*)calloc(groups, sizeof(lnsfaid_group_stats));
    if (!inputBits || !outputBits || !decodedBits || !fixInput || !m_stats) die("allocation", LNSFAID_E_NOMEM);
    /* page-locked transfer buffers: lnsfaid_decode then overlaps its copies with the decode */
    m_pinned = lnsfaid_host_register(fixInput, frames * m_N) == LNSFAID_OK && lnsfaid_host_register(decodedBits, frames * m_N) == LNSFAID_OK;
    /* the code definition comes from the Constants_SSE.h-format header, exactly as in the reference */
    static const int32_t deg[] = { DEG_1, DEG_2, DEG_3 };
    static const int32_t rows[] = { DEG_1_COMPUTATIONS, DEG_2_COMPUTATIONS, DEG_3_COMPUTATIONS };
    for (int k = 0; k < NB_DEGRES; ++k) { m_deg[k] = deg[k]; m_deg_rows[k] = rows[k]; }
    m_code.n_var = _NoVar; m_code.n_check = _NoCheck; m_code.n_edges = _NoOnes; m_code.z = 256;
    m_code.puncture_tail = 384; /* reference CDecoder_FAID.cpp:253-255 */
    m_code.nb_degres = NB_DEGRES; m_code.deg = m_deg; m_code.deg_rows = m_deg_rows; m_code.pos_vn = PosNoeudsVariable;
}

void CLDPC::FakeEncoder(const int* CodeWord_sym)
{
    /* reference CLDPC.cpp:163-207: every frame carries the same fixed codeword.  The reference refills its 32 frames on every
     * Run(); with thousands of streams per object that is gigabytes of stores per round, so the buffers are left alone while they
     * still hold this codeword (GenMsgSeq / Encode mark them changed) */
    if (m_fake_filled && m_fake_codeword == CodeWord_sym) return;
    m_fake_filled = true;
    m_fake_codeword = CodeWord_sym;
    for (int g = 0; g < m_groups; ++g) {
        int8_t* in = inputBits + (size_t)g * 32 * m_K;
        int8_t* out = outputBits + (size_t)g * 32 * m_N;
        for (int l = 0; l < 32; ++l) {
            for (int j = 0; j < m_K; ++j) in[(size_t)l * m_K + j] = out[(size_t)l * m_K + j] = (int8_t)(CodeWord_sym ? CodeWord_sym[j] : 0);
            for (int j = 0; j < m_M; ++j) out[(size_t)32 * m_K + (size_t)l * m_M + j] = (int8_t)(CodeWord_sym ? CodeWord_sym[m_K + j] : 0);
        }
    }
}

void CLDPC::GenMsgSeq()
{
    m_fake_filled = false;
    for (size_t i = 0; i < (size_t)m_groups * m_frame * m_K; ++i) inputBits[i] = (int8_t)(rand() % 2);
}

const CEncoder& CLDPC::Encoder()
{
    if (!m_encoder_ready) {
        std::vector<int> row_deg;
        for (int k = 0; k < NB_DEGRES; ++k) row_deg.insert(row_deg.end(), (size_t)m_deg_rows[k], m_deg[k]);
        if (!m_encoder.Initial(m_N, m_M, row_deg.data(), PosNoeudsVariable)) die("Encode: the parity part of H is singular", LNSFAID_E_CODE);
        m_encoder_ready = true;
    }
    return m_encoder;
}

void CLDPC::Encode()
{
    m_fake_filled = false;
    const CEncoder& enc = Encoder();
#pragma omp parallel for schedule(dynamic, 1)
    for (int g = 0; g < m_groups; ++g) enc.Encode32(inputBits + (size_t)g * 32 * m_K, outputBits + (size_t)g * 32 * m_N);
}

void CLDPC::float2LimitChar_4bit(int8_t* output, const float* input, float scale, size_t length)
{
    /* reference CLDPC.cpp:4553-4573: float multiply, truncate toward zero, saturating packs, clamp to +-7 */
    for (size_t i = 0; i < length; ++i) {
        const float y = input[i] * scale;
        int q = (y > -2147483648.0f && y < 2147483648.0f) ? (int)y : (int)0x80000000;
        q = q > 127 ? 127 : (q < -128 ? -128 : q);
        output[i] = (int8_t)(q > 7 ? 7 : (q < -7 ? -7 : q));
    }
}

lnsfaid_ctx* CLDPC::context(int method)
{
    lnsfaid_cfg cfg;
    int rc = lnsfaid_cfg_default(&cfg, method, nb_iteration);
    if (rc) die("lnsfaid_cfg_default", rc);
    cfg.factor_1 = m_factor_1;
    cfg.factor_2 = m_factor_2;
    if (!m_ctx[method]) {
        rc = lnsfaid_create(&m_ctx[method], &m_code, &cfg, m_device, (size_t)m_groups);
        if (rc) die("lnsfaid_create", rc);
    } else {
        rc = lnsfaid_set_cfg(m_ctx[method], &cfg);
        if (rc) die("lnsfaid_set_cfg", rc);
    }
    return m_ctx[method];
}

void CLDPC::decode_with(int method)
{
    lnsfaid_ctx* ctx = context(method);
    m_last = ctx;
    int rc;
    if (m_device_io) {
        int8_t *d_fix = nullptr, *d_out = nullptr;
        lnsfaid_group_stats* d_st = nullptr;
        rc = lnsfaid_io_buffers(ctx, &d_fix, &d_out, &d_st);
        if (rc) die("lnsfaid_io_buffers", rc);
        rc = lnsfaid_decode_device(ctx, d_fix, (size_t)m_groups, d_out, d_st);
        if (!rc) rc = lnsfaid_read_stats(ctx, m_stats, (size_t)m_groups); /* 8 bytes per group: iteration counts, BFiter */
    } else {
        rc = lnsfaid_decode(ctx, fixInput, (size_t)m_groups, decodedBits, m_stats);
    }
    if (rc) die("lnsfaid_decode", rc);
}

void CLDPC::DeviceFrames(int decode_method, bool per_stream_frames, int interleave_mod_type)
{
    lnsfaid_ctx* ctx = context(decode_method);
    if (lnsfaid_frontend_set_interleave(ctx, interleave_mod_type)) die("lnsfaid_frontend_set_interleave", LNSFAID_E_INVAL);
    const int rc = per_stream_frames ? lnsfaid_frontend_set_frames(ctx, outputBits, inputBits, (size_t)m_groups)
                                     : lnsfaid_frontend_set_frames(ctx, nullptr, nullptr, 0);
    if (rc) die("lnsfaid_frontend_set_frames", rc);
}

void CLDPC::DeviceChannel(int decode_method, const uint32_t* states, const uint64_t* draws_before, int mod_type, float sigma,
                          float scale)
{
    lnsfaid_ctx* ctx = context(decode_method);
    int8_t* d_fix = nullptr;
    int rc = lnsfaid_io_buffers(ctx, &d_fix, nullptr, nullptr);
    if (rc) die("lnsfaid_io_buffers", rc);
    m_last = ctx;
    rc = lnsfaid_frontend_device_states(ctx, states, draws_before, (size_t)m_groups, mod_type, sigma, scale, nullptr, d_fix);
    if (rc) die("lnsfaid_frontend_device_states", rc);
    m_device_io = true;
}

uint64_t CLDPC::DrawsPerGroup(int mod_type)
{
    return (uint64_t)32 * (uint64_t)m_N / (uint64_t)mod_type * 4u; /* 2 normals per symbol, 2 uniforms per normal */
}

void CLDPC::Decode() { decode_with(0); }
void CLDPC::Decode_OMS() { decode_with(1); }
void CLDPC::Decode_FAID() { decode_with(2); }
void CLDPC::Decode_FAID_2B1C() { decode_with(5); }
int CLDPC::Decode_OMSBF() { decode_with(3); return m_stats[0].bf_iterations; }
int CLDPC::Decode_OMS_DTBF() { decode_with(4); return m_stats[0].bf_iterations; }

/* The collect-flag dump of reference CLDPC.cpp:4877-4983: one record per frame with information-bit errors appended to
 * errorindex.txt (block / index of every wrong information and parity bit), errorfloat.txt (channel output and 4-bit
 * input of the frame) and errordecode.txt (decoded, sent information and sent code bits).  bpskinput has the fixInput
 * layout ([32][K] then [32][M] per group); the reference's Z is Profile.txt's. */
void CLDPC::CollectErrors(const float* bpskinput, int Z)
{
    if (m_device_io) {
        static bool warned = false;
        if (!warned) fprintf(stderr, "collectflag: frames stay on the device with --device-frontend, no error dumps are written\n");
        warned = true;
        return;
    }
    const int N = m_N, K = m_K, M = m_M;
    std::vector<int> bit_block, bit_index, chk_block, chk_index;
    for (int g = 0; g < m_groups; ++g) {
        const int8_t* dec = decodedBits + (size_t)g * 32 * N;
        const int8_t* in = inputBits + (size_t)g * 32 * K;
        const int8_t* outb = outputBits + (size_t)g * 32 * N;
        const int8_t* chr = fixInput + (size_t)g * 32 * N;
        const float* flt = bpskinput ? bpskinput + (size_t)g * 32 * N : nullptr;
        for (int i = 0; i < 32; ++i) {
            bit_block.clear(); bit_index.clear(); chk_block.clear(); chk_index.clear();
            for (int j = 0; j < K; ++j)
                if (dec[(size_t)i * N + j] != in[(size_t)i * K + j]) { bit_block.push_back(j / Z + 1); bit_index.push_back(j % Z); }
            if (bit_block.empty()) continue;
            for (int j = K; j < N; ++j)
                if (dec[(size_t)i * N + j] != outb[(size_t)32 * K + (size_t)i * M + (j - K)]) { chk_block.push_back(j / Z + 1); chk_index.push_back(j % Z); }
            std::ofstream eout("errorindex.txt", std::ios::app), nout("errorfloat.txt", std::ios::app), dout("errordecode.txt", std::ios::app);
            eout << "ErrorFrame: " << i << std::endl << "ErrorBit Num: " << bit_block.size() << std::endl << "Errorbit Block: ";
            for (int v : bit_block) eout << v << "\t";
            eout << std::endl << "Errobit Index: ";
            for (int v : bit_index) eout << v << "\t";
            eout << std::endl << "Errorcheck Num: " << chk_block.size() << std::endl << "Errorcheck Block: ";
            for (int v : chk_block) eout << v << "\t";
            eout << std::endl << "Errorcheck Index: ";
            for (int v : chk_index) eout << v << "\t";
            eout << std::endl;
            nout << "ErrorFloat=[ ";
            if (flt) {
                for (int j = 0; j < K; ++j) nout << flt[(size_t)i * K + j] << "\t";
                for (int j = 0; j < M; ++j) nout << flt[(size_t)32 * K + (size_t)i * M + j] << "\t";
            }
            nout << "];" << std::endl << "ErrorChar=[";
            for (int j = 0; j < K; ++j) nout << (int)chr[(size_t)i * K + j] << "\t";
            for (int j = 0; j < M; ++j) nout << (int)chr[(size_t)32 * K + (size_t)i * M + j] << "\t";
            nout << "];" << std::endl << std::endl;
            dout << "Decodedbits=[";
            for (int j = 0; j < N; ++j) dout << (int)dec[(size_t)i * N + j] << "\t";
            dout << "];" << std::endl << "inputbits=[";
            for (int j = 0; j < K; ++j) dout << (int)in[(size_t)i * K + j] << "\t";
            dout << "];" << std::endl << "outputbits=[";
            for (int j = 0; j < K; ++j) dout << (int)outb[(size_t)i * K + j] << "\t";
            for (int j = 0; j < M; ++j) dout << (int)outb[(size_t)32 * K + (size_t)i * M + j] << "\t";
            dout << "];" << std::endl << std::endl;
        }
    }
}

Statistic CLDPC::CalculateErrors()
{
    lnsfaid_ctx* ctx = m_last; /* the decoder that produced decodedBits (device mode: whose buffers hold them) */
    if (!ctx) die("CalculateErrors before any Decode_*", LNSFAID_E_INVAL);
    uint64_t out[4] = { 0, 0, 0, 0 };
    int rc;
    if (m_device_io) {
        int8_t* d_out = nullptr;
        rc = lnsfaid_io_buffers(ctx, nullptr, &d_out, nullptr);
        const int8_t* d_in = nullptr; /* NULL = all-zero codeword; the sent information bits when frames are set */
        if (!rc) rc = lnsfaid_frontend_input_bits(ctx, &d_in);
        if (!rc) rc = lnsfaid_count_errors_device(ctx, d_out, d_in, (size_t)m_groups, out);
    } else {
        rc = lnsfaid_count_errors(ctx, decodedBits, inputBits, (size_t)m_groups, out);
    }
    if (rc) die("lnsfaid_count_errors", rc);
    Statistic s;
    s.ErrorFrame = out[1]; s.ErrorBits = out[2]; s.LT3ErrBitFrame = out[3];
    return s;
}

void CLDPC::CommInit(int decode_method, int n_ranks, int rank, const uint8_t id[LNSFAID_COMM_ID_BYTES])
{
    m_comm_ctx = context(decode_method);
    const int rc = lnsfaid_comm_init(m_comm_ctx, n_ranks, rank, id);
    if (rc) die("lnsfaid_comm_init", rc);
}

void CLDPC::AllReduceCounters(unsigned long counters[4])
{
    if (!m_comm_ctx) die("AllReduceCounters before CommInit", LNSFAID_E_INVAL);
    uint64_t c[4] = { counters[0], counters[1], counters[2], counters[3] };
    const int rc = lnsfaid_allreduce_counters(m_comm_ctx, c);
    if (rc) die("lnsfaid_allreduce_counters", rc);
    for (int i = 0; i < 4; ++i) counters[i] = (unsigned long)c[i];
}

double CLDPC::KernelMs(bool reset)
{
    double total = 0;
    for (auto c : m_ctx)
        if (c) { double ms = 0; uint64_t n = 0; lnsfaid_kernel_time(c, &ms, &n, reset); total += ms; }
    return total;
}
