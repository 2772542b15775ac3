/*
 * CSimulate.h — the per-worker simulation loop of the reference (CSimulate.{h,cpp}) re-hosted for a GPU.
 *
 * The reference runs one CSimulate per hardware thread; each owns a channel seeded with seed[index]
 * (CSimulate.cpp:11-17, :57) and decodes one group of 32 frames per call, 50 calls per Run()
 * (CSimulate.cpp:118-169).  Here one CSimulate owns `streams` such workers ("virtual threads"
 * first_index .. first_index+streams-1): their 32-frame groups are generated on the host cores (OpenMP, one
 * stream per core at a time) and decoded together as ONE batch of `streams` groups per call.  Stream s sees
 * exactly the LLRs reference thread first_index+s would see, so the summed counters equal the reference's.
 */
#ifndef CSIMULATE_H
#define CSIMULATE_H
#include <vector>

#include "CChannel.h"
#include "CLDPC.h"
#include "CTool.h"

class CSimulate {
public:
    CLDPC* ldpc = nullptr;
    std::vector<CChannel> channel; /* one per stream */
    unsigned long TestFrame = 0, ErrorFrame = 0, ErrorBits = 0, LT3ErrBitFrame = 0;
    float sigma = 0, snr = 0, scale = 0;
    int decode_method = 0, ModulationType = 2, InterleaveModType = 1;
    double decode_seconds = 0; /* host wall time spent inside Decode_*() */
    bool device_frontend = false; /* generate the channel output on the GPU (lnsfaid_frontend_device) */
    bool encode = false;          /* GenMsgSeq + Encode instead of FakeEncoder (reference FAKE_ENCODE 0) */
    unsigned long sum_iterations = 0, sum_bf_iterations = 0, decoded_groups = 0;

    ~CSimulate();
    void Initial(Parameter_Simulation& p, int first_index, int streams, int device);
    void Configure(float Eb_N0, int decode_method);
    void Run(); /* 50 decode calls of `streams` groups, reference CSimulate.cpp:92-180 */

private:
    std::vector<Complex8> ModSeq;   /* modulated fixed codeword of one group (QPSK) */
    std::vector<float> BPSKModSeq;
    int m_first = 0, m_streams = 0, m_Z = 256;
    std::vector<uint64_t> m_draws; /* all zero: the device front-end is handed the current generator states */
};

extern const char* g_dump_fixinput; /* --dump-fixinput F: every call appends the batch's fixInput to F (tests: BPSK noise is not reproducible elsewhere) */
int SimulationSeed(int index); /* the reference's seed table, CSimulate.cpp:11-17 */
#endif
