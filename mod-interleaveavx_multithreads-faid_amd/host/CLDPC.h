/*
 * CLDPC.h — host-side mirror of the reference's `class CLDPC` (reference CLDPC.h:110-171) for the decode
 * path.  Same member names and meaning for what callers touch (fixInput, decodedBits, inputBits, outputBits,
 * nb_iteration, m_N/m_K/m_M/m_frame/m_Rate) and the same entry points; the bodies of Decode_OMS /
 * Decode_FAID / Decode_FAID_2B1C / CalculateErrors are calls into the C ABI (include/lnsfaid.h).
 *
 * One extension: a CLDPC object can carry `m_groups` consecutive groups of 32 frames (the reference always
 * has one), so that one Decode_*() call feeds the GPU a whole batch.  Buffers are then the reference's
 * layout repeated per group.
 */
#ifndef CLDPC_H
#define CLDPC_H
#include <cstddef>
#include <cstdint>

#include "./Constants/Constants_SSE.h" /* generated: same macros and PosNoeudsVariable as the reference's header */
#include "CEncoder.h"
#include "CTool.h"
#include "lnsfaid.h"

struct Statistic { /* reference CLDPC.h:103-108 */
    unsigned long ErrorFrame;
    unsigned long ErrorBits;
    unsigned long LT3ErrBitFrame;
};

class CLDPC {
public:
    double m_Rate;
    int8_t* inputBits;   /* [groups][32][K] information bits                                */
    int8_t* outputBits;  /* [groups]([32][K] then [32][M]) encoder output, reference layout  */
    int8_t* decodedBits; /* [groups][32][N] hard decisions                                   */
    int8_t* fixInput;    /* [groups]([32][K] then [32][M]) quantised LLRs in [-7, 7]         */
    int nb_iteration;
    int m_M, m_N, m_K, m_PunLen, m_ShortenLen, m_frame;
    int m_groups;

    CLDPC();
    ~CLDPC();
    CLDPC(const CLDPC&) = delete;
    CLDPC& operator=(const CLDPC&) = delete;

    /* reference CLDPC::Initial(nb_frame, MaxIteration) plus the batch size and the GPU to use */
    void Initial(int nb_frame, int MaxItertion, int groups = 1, int device = 0);
    void FakeEncoder(const int* CodeWord_sym = nullptr); /* nullptr = the shipped all-zero CodeWord_sym */
    void GenMsgSeq();  /* reference CLDPC.cpp:60-66: inputBits[i] = rand() % 2 */
    void Encode();     /* reference CLDPC.cpp:68-155, with the generator derived from the code table (CEncoder.h) */
    const CEncoder& Encoder(); /* derived on first use (about a second) */
    void float2LimitChar_4bit(int8_t* output, const float* input, float scale, size_t length);

    void Decode();           /* DecodeMethod 0 and the switch default: normalised min-sum (reference CLDPC.cpp:214) */
    void Decode_OMS();       /* DecodeMethod 1 */
    void Decode_FAID();      /* DecodeMethod 2 */
    void Decode_FAID_2B1C(); /* DecodeMethod 5 */
    int Decode_OMSBF();      /* DecodeMethod 3; returns the bit-flipping iterations of the first group */
    int Decode_OMS_DTBF();   /* DecodeMethod 4; returns the bit-flipping iterations of the first group (reference: BFiter) */
    Statistic CalculateErrors();
    /* collectflag == 1 part of the reference's CalculateErrors(bpskinput, charinput, collectflag) (CLDPC.h:169):
     * appends the error frames of the last decode to errorindex.txt / errorfloat.txt / errordecode.txt */
    void CollectErrors(const float* bpskinput, int Z);

    /* Device-resident mode (--device-frontend): the channel output of `m_groups` reference worker threads is
     * generated on the GPU straight into the decoder's input buffer, Decode_*() and CalculateErrors() then work on
     * device buffers and fixInput / decodedBits on the host are not touched.  Sends the all-zero codeword (FakeEncoder
     * with the shipped CodeWord_sym) unless DeviceFrames() installed the encoder's output. */
    /* with per_stream_frames the device front-end sends outputBits (after Encode) instead of the all-zero codeword and
     * CalculateErrors compares with inputBits */
    void DeviceFrames(int decode_method, bool per_stream_frames, int interleave_mod_type = 1);
    /* states: RS.IX, RS.IY, RS.IZ per group (3 words each) at draws_before = 0 */
    void DeviceChannel(int decode_method, const uint32_t* states, const uint64_t* draws_before, int mod_type, float sigma,
                       float scale);
    uint64_t DrawsPerGroup(int mod_type);

    /* Factor_1 / Factor_2 as the reference re-reads them from Profile.txt on every decode call */
    void SetFactors(int factor_1, int factor_2) { m_factor_1 = factor_1; m_factor_2 = factor_2; }
    const lnsfaid_group_stats* GroupStats() const { return m_stats; }
    double KernelMs(bool reset);

    /* one process per GPU: bind an RCCL communicator to the decoder of `decode_method` (collective over the ranks) and sum
     * {TestFrame, ErrorFrame, ErrorBits, LT3ErrBitFrame} over them, as reference main.cpp:174-182 does over its threads */
    void CommInit(int decode_method, int n_ranks, int rank, const uint8_t id[LNSFAID_COMM_ID_BYTES]);
    void AllReduceCounters(unsigned long counters[4]);

private:
    void decode_with(int method);
    lnsfaid_ctx* context(int method);
    lnsfaid_ctx* m_ctx[6]; /* one context per DecodeMethod, created on first use */
    lnsfaid_ctx* m_last = nullptr; /* the context of the last decode / device channel call: CalculateErrors counts ITS output */
    lnsfaid_ctx* m_comm_ctx = nullptr;
    lnsfaid_code m_code;
    int32_t m_deg[NB_DEGRES], m_deg_rows[NB_DEGRES];
    lnsfaid_group_stats* m_stats;
    int m_device, m_factor_1, m_factor_2;
    bool m_device_io;
    bool m_pinned = false;
    CEncoder m_encoder;
    bool m_encoder_ready = false;
    bool m_fake_filled = false;            /* inputBits / outputBits hold FakeEncoder(m_fake_codeword) for every group */
    const int* m_fake_codeword = nullptr;
};
#endif
