#include "CChannel.h"

#include <cmath>

float CChannel::Random_Uniform(RandSeed& rs)
{
    /* reference CChannel.cpp:71-80 (float arithmetic) */
    float temp = 0.0;
    rs.IX = (rs.IX * 249) % 61967;
    rs.IY = (rs.IY * 251) % 63443;
    rs.IZ = (rs.IZ * 252) % 63599;
    temp = (((float)rs.IX) / ((float)61967)) + (((float)rs.IY) / ((float)63443)) + (((float)rs.IZ) / ((float)63599));
    temp -= (int)temp;
    return temp;
}

float CChannel::Random_Norm(double sigma, RandSeed& rs)
{
    /* reference CChannel.cpp:82-89 (double arithmetic, float result) */
    float u1, u2, u;
    u1 = Random_Uniform(rs);
    u2 = Random_Uniform(rs);
    u = sigma * cos(2 * 3.1415926535897932384626433832795 * u2) * sqrt(-2.0 * log(1.0 - u1));
    return u;
}

void CChannel::AWGNChannel(const Complex8* in, float sigma)
{
    for (unsigned long i = 0; i < SymbolLen; i++) { /* reference CChannel.cpp:90-97 */
        SymbolSeq[i].real = Random_Norm(sigma, RS) + in[i].real;
        SymbolSeq[i].imag = Random_Norm(sigma, RS) + in[i].imag;
    }
}

void CChannel::BPSKAWGNChannel(const float* in, float sigma)
{
    std::normal_distribution<float> g(0.0f, sigma);
    for (unsigned long i = 0; i < SymbolLen; ++i) BPSKSymbol[i] = in[i] + g(m_bpsk_rng);
}

void CChannel::Initial(unsigned long len, int index)
{
    SymbolLen = len;
    SymbolSeq.assign(len, Complex8{ 0, 0 });
    BPSKSymbol.assign(len, 0.f);
    RS.IX = RS.IY = RS.IZ = (unsigned long)RandomSeed; /* reference CChannel.cpp:121, CONTINUE_SEED 0 */
    m_bpsk_rng.seed(0x9E3779B97F4A7C15ull ^ (unsigned long long)RandomSeed ^ ((unsigned long long)index << 32));
}
