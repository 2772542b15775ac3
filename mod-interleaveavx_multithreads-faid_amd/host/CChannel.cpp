#include "CChannel.h"

#include <cmath>

/* One step of the three Wichmann-Hill style congruential generators and their combined fractional part
 * (reference CChannel.cpp:71-80).  Everything after the integer update is float arithmetic, summed left to right. */
float CChannel::Random_Uniform(RandSeed& rs)
{
    static const unsigned long mul[3] = { 249, 251, 252 }, mod[3] = { 61967, 63443, 63599 };
    unsigned long* x[3] = { &rs.IX, &rs.IY, &rs.IZ };
    float sum = 0.0f;
    for (int g = 0; g < 3; ++g) {
        *x[g] = (*x[g] * mul[g]) % mod[g];
        sum = g == 0 ? (float)*x[g] / (float)mod[g] : sum + (float)*x[g] / (float)mod[g];
    }
    return sum - (float)(int)sum;
}

/* Box-Muller on two successive uniforms, evaluated in double and rounded to float once (reference CChannel.cpp:82-89) */
float CChannel::Random_Norm(double sigma, RandSeed& rs)
{
    const float first = Random_Uniform(rs), second = Random_Uniform(rs);
    const double two_pi = 2 * 3.1415926535897932384626433832795;
    return (float)(sigma * cos(two_pi * second) * sqrt(-2.0 * log(1.0 - first)));
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
