#include "CSimulate.h"

#include <algorithm>
#include <fstream>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#define BitsOverChannelLocal (_NoVar - _PunctureBits - _ShortenBits)

/* seed table of the reference's worker threads (data, reference CSimulate.cpp:11-17) */
static const int seed_table[] = { 101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191,
    193, 197, 199, 211, 223, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307, 311, 313, 317, 331, 337,
    347, 349, 353, 359, 367, 373, 379, 383, 389, 397, 401, 409, 419, 421, 431, 433, 439, 443, 449, 457, 461, 463, 467, 479,
    487, 491, 499, 503, 507, 521, 523, 541, 547, 563, 569, 571, 577, 587, 593, 599, 601, 607, 613, 617, 619, 631, 641, 643,
    647, 653, 659, 661, 673, 677, 683, 691, 701, 709, 719, 727, 733, 739, 743, 751, 757, 761, 769, 773, 787, 797, 809, 811,
    821, 823, 827, 829, 839, 953, 857, 859, 863, 877, 881, 883, 887, 907, 911, 919, 929, 937, 941, 947, 953, 967, 971, 977,
    983, 991, 997, 1009, 1013, 1019 };

const char* g_dump_fixinput = nullptr;

int SimulationSeed(int index)
{
    const int n = (int)(sizeof(seed_table) / sizeof(seed_table[0]));
    /* beyond the reference's table (it would read out of bounds): keep going with distinct odd seeds */
    return index < n ? seed_table[index] : 1021 + 2 * (index - n);
}

CSimulate::~CSimulate() { delete ldpc; }

void CSimulate::Initial(Parameter_Simulation& p, int first_index, int streams, int device)
{
    scale = p.scale;
    m_first = first_index;
    m_streams = streams;
    m_Z = p.Z > 0 ? p.Z : 256;
    ModulationType = p.mod_type;
    InterleaveModType = p.interleavemod_type;
    const bool qam = ModulationType == 2 || ModulationType == 4 || ModulationType == 6 || ModulationType == 8;
    if ((ModulationType != 1 && !qam) || InterleaveModType < 1 || _NoVar % InterleaveModType != 0
        || (qam && (32L * _NoVar) % ModulationType != 0)) {
        fprintf(stderr, "front-end: modType 1 (BPSK), 2, 4, 6, 8 (QPSK, 16-, 64-, 256-QAM); InterleaveModType must divide the frame length\n");
        exit(EXIT_FAILURE);
    }
    ldpc = new CLDPC();
    ldpc->Initial(p.nb_frames, p.Max_Iteration, streams, device);
    ldpc->SetFactors(p.Factor_1, p.Factor_2);
    const unsigned long SourceLen = (unsigned long)ldpc->m_frame * BitsOverChannelLocal;
    const unsigned long SymbolLen = SourceLen / (unsigned long)ModulationType; /* reference CModulate.cpp:66-78 */
    channel.resize(streams);
    m_draws.assign(streams, 0);
    for (int s = 0; s < streams; ++s) {
        channel[s].RandomSeed = SimulationSeed(first_index + s);
        channel[s].Initial(SymbolLen, first_index + s);
    }
}

void CSimulate::Configure(float Eb_N0, int _decode_method)
{
    snr = Eb_N0; /* reference CSimulate.cpp:67-74 */
    if (ModulationType == 1) sigma = (float)(1.0 / sqrt(2.0 * ldpc->m_Rate * ModulationType * pow(10.0, 0.1 * snr)));
    else sigma = (float)(1.0 / sqrt(ldpc->m_Rate * ModulationType * pow(10.0, 0.1 * snr)));
    /* the reference's switch sends every value outside 1..5 to Decode() (CSimulate.cpp:161-163) */
    decode_method = (_decode_method >= 1 && _decode_method <= 5) ? _decode_method : 0;
    TestFrame = ErrorFrame = ErrorBits = LT3ErrBitFrame = 0;
}

void CSimulate::Run()
{
    /* Gray-mapped amplitude tables per axis and the max-log demapper's fold constants (reference CModulate.cpp:4-7, :273-362) */
    static const float table_qpsk[2] = { -0.707107f, 0.707107f };
    static const float table_16qam[4] = { -0.316228f, -0.948683f, 0.316228f, 0.948683f };
    static const float table_64qam[8] = { -0.462910f, -0.154303f, -0.771517f, -1.08012f, 0.462910f, 0.154303f, 0.771517f, 1.08012f };
    static const float table_256qam[16] = { -0.383482f, -0.536875f, -0.230089f, -0.076696f, -0.843661f, -0.690268f, -0.997054f, -1.150447f,
                                            0.383482f, 0.536875f, 0.230089f, 0.076696f, 0.843661f, 0.690268f, 0.997054f, 1.150447f };
    static const double fold_16[1] = { 0.6324555 }, fold_64[2] = { 0.6172134, 0.3086067 }, fold_256[3] = { 0.613568, 0.306784, 0.153392 };
    const float* axis = ModulationType == 2 ? table_qpsk : ModulationType == 4 ? table_16qam : ModulationType == 6 ? table_64qam : table_256qam;
    const double* fold = ModulationType == 4 ? fold_16 : ModulationType == 6 ? fold_64 : fold_256;
    const int N = ldpc->m_N, K = ldpc->m_K, M = ldpc->m_M;
    const int Q = ModulationType, half = Q / 2, I = InterleaveModType;
    /* block interleaver inside every frame (reference CModulate.cpp:134-146, :152-166): position p carries code bit k(p) */
    auto code_bit = [N, I](int p) { return (N / I) * (p % I) + p / I; };
    /* FAKE_ENCODE (the reference's default, CSimulate.cpp:103-104: GenMatrix is not shipped) or, with --encode, random
     * information bits through the encoder derived from the code table (reference #else branch :106-107) */
    if (encode) { ldpc->GenMsgSeq(); ldpc->Encode(); }
    else ldpc->FakeEncoder();
    /* interleave + modulate once per 50 calls, reference CSimulate.cpp:111-116.
     * outputBits of a group is [32][K] then [32][M]; frame m's bit k sits at m*N + k after
     * BeforeModulationInterleaver (CModulate.cpp:95-148).  With FakeEncoder every stream sends the same 32 frames:
     * one modulated sequence serves all of them. */
    const size_t bits = (size_t)32 * N;
    const int n_seq = encode ? m_streams : 1;
    const size_t sym = ModulationType == 1 ? bits : bits / (size_t)ModulationType;
    if (ModulationType == 1) BPSKModSeq.resize((size_t)n_seq * sym);
    else ModSeq.resize((size_t)n_seq * sym);
    for (int g = 0; g < n_seq; ++g) {
        const int8_t* ob = ldpc->outputBits + (size_t)g * bits;
        auto tx_bit = [&](int m, int k) { return k < K ? ob[(size_t)m * K + k] : ob[(size_t)32 * K + (size_t)m * M + (k - K)]; };
        if (ModulationType == 1) {
            float* dst = BPSKModSeq.data() + (size_t)g * sym;
            for (int m = 0; m < 32; ++m) for (int k = 0; k < N; ++k) dst[(size_t)m * N + k] = 2.0f * tx_bit(m, k) - 1.0f; /* CModulate.cpp:368 */
        } else { /* Modulation (reference CModulate.cpp:216-264): even positions index the in-phase entry MSB first, odd ones the quadrature entry */
            Complex8* dst = ModSeq.data() + (size_t)g * sym;
            for (size_t i = 0; i < bits / (size_t)Q; ++i) {
                int idx_i = 0, idx_q = 0;
                for (int u = 0; u < Q; ++u) {
                    const size_t pos = (size_t)Q * i + u;
                    const int b = tx_bit((int)(pos / N), code_bit((int)(pos % N)));
                    if (u & 1) idx_q += b << (half - u / 2 - 1); else idx_i += b << (half - u / 2 - 1);
                }
                dst[i].real = axis[idx_i];
                dst[i].imag = axis[idx_q];
            }
        }
    }
    if (device_frontend) ldpc->DeviceFrames(decode_method, encode, InterleaveModType); /* the 32 frames of every stream, once per 50 calls */
    std::vector<float> llr(device_frontend ? 0 : (size_t)m_streams * bits);
    std::vector<int> BFiters_((size_t)m_streams * 51, 0); /* per stream, reference CSimulate.cpp:99 */
    std::vector<uint32_t> states(3 * (size_t)m_streams);
    for (int call = 0; call < 50; ++call) {
        TestFrame += 32ul * m_streams;
        if (device_frontend) {
            if (ModulationType == 1) { fprintf(stderr, "--device-frontend needs a QAM modType (2, 4, 6, 8)\n"); exit(EXIT_FAILURE); }

            /* every call starts from the streams' current generator states (RS), so a resumed run continues seamlessly */
            for (int s = 0; s < m_streams; ++s) {
                states[3 * (size_t)s] = (uint32_t)channel[s].RS.IX;
                states[3 * (size_t)s + 1] = (uint32_t)channel[s].RS.IY;
                states[3 * (size_t)s + 2] = (uint32_t)channel[s].RS.IZ;
            }
            ldpc->DeviceChannel(decode_method, states.data(), m_draws.data(), ModulationType, sigma, scale);
            const uint64_t n = ldpc->DrawsPerGroup(ModulationType);
            for (int s = 0; s < m_streams; ++s) {
                /* keep RS (the resume table of Temp.txt) where the host generator would be: X <- X * a^n mod m */
                auto jump = [n](unsigned long x, unsigned long a, unsigned long m) {
                    unsigned long r = 1, b = a % m; uint64_t e = n;
                    while (e) { if (e & 1) r = r * b % m; b = b * b % m; e >>= 1; }
                    return x % m * r % m;
                };
                channel[s].RS.IX = jump(channel[s].RS.IX, 249, 61967);
                channel[s].RS.IY = jump(channel[s].RS.IY, 251, 63443);
                channel[s].RS.IZ = jump(channel[s].RS.IZ, 252, 63599);
            }
        }
#pragma omp parallel for schedule(dynamic, 1) if (!device_frontend)
        for (int s = 0; s < (device_frontend ? 0 : m_streams); ++s) {
            float* dst = llr.data() + (size_t)s * bits;
            CChannel& ch = channel[s];
            int8_t* fix = ldpc->fixInput + (size_t)s * bits;
            if (ModulationType == 1) {
                ch.BPSKAWGNChannel(BPSKModSeq.data() + (encode ? (size_t)s * sym : 0), sigma);
                for (size_t i = 0; i < bits; ++i) dst[i] = ch.BPSKSymbol[i];
            } else {
                ch.AWGNChannel(ModSeq.data() + (encode ? (size_t)s * sym : 0), (float)(sigma / sqrt(2))); /* reference CSimulate.cpp:126 */
                for (size_t i = 0; i < bits / (size_t)Q; ++i) { /* max-log Demodulation, every level stored as float (CModulate.cpp:273-362) */
                    float l[8];
                    l[0] = ch.SymbolSeq[i].real;
                    l[1] = ch.SymbolSeq[i].imag;
                    for (int n = 1; n < half; ++n) {
                        l[2 * n] = fabs(l[2 * n - 2]) - fold[n - 1];
                        l[2 * n + 1] = fabs(l[2 * n - 1]) - fold[n - 1];
                    }
                    for (int u = 0; u < Q; ++u) { /* de-interleave: frame-major by code bit */
                        const size_t pos = (size_t)Q * i + u;
                        dst[pos / N * N + (size_t)code_bit((int)(pos % N))] = l[u];
                    }
                }
            }
            /* AfterDeModulationDeInterleaver (CModulate.cpp:152-212) + float2LimitChar_4bit, frame-major -> [32][K] | [32][M] */
            for (int m = 0; m < 32; ++m) {
                ldpc->float2LimitChar_4bit(fix + (size_t)m * K, dst + (size_t)m * N, scale, (size_t)K);
                ldpc->float2LimitChar_4bit(fix + (size_t)32 * K + (size_t)m * M, dst + (size_t)m * N + K, scale, (size_t)M);
            }
        }
        if (g_dump_fixinput && !device_frontend) {
            std::ofstream dump(g_dump_fixinput, std::ios::binary | std::ios::app);
            dump.write((const char*)ldpc->fixInput, (std::streamsize)((size_t)m_streams * bits));
        }
        const auto t0 = std::chrono::steady_clock::now();
        switch (decode_method) { /* reference CSimulate.cpp:136-164 */
        case 0: ldpc->Decode(); break;
        case 1: ldpc->Decode_OMS(); break;
        case 2: ldpc->Decode_FAID(); break;
        case 3: (void)ldpc->Decode_OMSBF(); break;
        case 4: (void)ldpc->Decode_OMS_DTBF(); break;
        case 5: ldpc->Decode_FAID_2B1C(); break;
        default: ldpc->Decode(); break; /* as the reference: any other value runs NMS */
        }
        decode_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int s = 0; s < m_streams; ++s) {
            const lnsfaid_group_stats& st = ldpc->GroupStats()[s];
            sum_iterations += st.iterations; sum_bf_iterations += st.bf_iterations;
            /* BF_ITER_COUNT (reference CSimulate.cpp:146-156): only DecodeMethod 3 and 4 return BFiter */
            if ((decode_method == 3 || decode_method == 4) && st.bf_iterations >= 0 && st.bf_iterations <= 50) BFiters_[(size_t)s * 51 + st.bf_iterations]++;
        }
        decoded_groups += m_streams;
        const Statistic Test = ldpc->CalculateErrors();
        ErrorFrame += Test.ErrorFrame;
        ErrorBits += Test.ErrorBits;
        LT3ErrBitFrame += Test.LT3ErrBitFrame;
        if (collectflag == 1 && Test.ErrorFrame > 0) { /* reference CLDPC.cpp:4877: set by main once FER < 1e-5 */
            std::vector<float> fl; /* DeInterLeaveSeq layout: [32][K] then [32][M] per group */
            if (!device_frontend) {
                fl.resize((size_t)m_streams * bits);
                for (int s = 0; s < m_streams; ++s)
                    for (int m = 0; m < 32; ++m) {
                        const float* src = llr.data() + (size_t)s * bits + (size_t)m * N;
                        std::copy(src, src + K, fl.data() + (size_t)s * bits + (size_t)m * K);
                        std::copy(src + K, src + N, fl.data() + (size_t)s * bits + (size_t)32 * K + (size_t)m * M);
                    }
            }
            ldpc->CollectErrors(fl.empty() ? nullptr : fl.data(), m_Z);
        }
    }
    /* reference CSimulate.cpp:171-179: every worker appends its histogram after its 50 calls */
    std::ofstream iterOut("iterCount.txt", std::ios::app);
    for (int s = 0; s < m_streams; ++s)
        for (int i = 1; i <= 50; i++)
            if (BFiters_[(size_t)s * 51 + i] != 0) iterOut << i << ": " << BFiters_[(size_t)s * 51 + i] << std::endl;
}
