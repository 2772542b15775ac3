/*
 * CChannel.h — AWGN channel of one simulation stream, mirroring reference CChannel.{h,cpp}.
 * QPSK and higher use the reference's own generator (Wichmann-Hill triple + Box-Muller, CChannel.cpp:71-97),
 * restated bit for bit so that a stream seeded like reference thread `index` sees the same noise.
 * BPSK in the reference draws from Intel MKL's MT2203 (CChannel.cpp:49,105), which is not available here:
 * BPSKAWGNChannel uses std::mt19937_64 + std::normal_distribution instead (statistically equivalent, not the
 * same stream; the decoder boundary is where BPSK parity is pinned).
 */
#ifndef CCHANNEL_H
#define CCHANNEL_H
#include <random>
#include <vector>

struct Complex8 { float real, imag; }; /* layout of MKL_Complex8 */

struct RandSeed { unsigned long IX, IY, IZ; }; /* reference CChannel.h:15-20 */

class CChannel {
public:
    std::vector<Complex8> SymbolSeq;
    std::vector<float> BPSKSymbol;
    unsigned long SymbolLen = 0;
    RandSeed RS{ 0, 0, 0 };
    int RandomSeed = 0;

    float Random_Uniform(RandSeed& rs);
    float Random_Norm(double sigma, RandSeed& rs);
    void AWGNChannel(const Complex8* inSymbolSeq, float sigma);
    void BPSKAWGNChannel(const float* inSymbolSeq, float sigma);
    void Initial(unsigned long len, int index);

private:
    std::mt19937_64 m_bpsk_rng;
};
#endif
