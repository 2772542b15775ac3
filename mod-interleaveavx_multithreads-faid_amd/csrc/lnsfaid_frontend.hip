/*
 * lnsfaid_frontend.hip — the reference's signal chain in front of the decoder, generated on the GPU
 * (SURVEY.md §8(f) N1).  One launch produces the `fixInput` of n_streams groups: stream s plays the reference's
 * worker thread with RandomSeed seeds[s] (CSimulate.cpp:11-17, :57) and has already consumed draws_before[s]
 * uniforms of its Wichmann-Hill generator.
 *
 *   CChannel::Random_Uniform  CChannel.cpp:71-80   three multiplicative congruential generators, float sum, fraction
 *   CChannel::Random_Norm     CChannel.cpp:82-89   Box-Muller in double from two consecutive uniforms
 *   CChannel::AWGNChannel     CChannel.cpp:90-97   real then imag of each symbol
 *   CModulate::Modulation / Demodulation, QPSK and 16-QAM   CModulate.cpp:216-293
 *   AfterDeModulationDeInterleaver (InterleaveModType 1)     CModulate.cpp:152-212
 *   CLDPC::float2LimitChar_4bit   CLDPC.cpp:4553-4573
 *
 * The sequential generator is parallelised exactly: the state after n draws is X0 * a^n mod m, so every thread
 * jumps to its first draw with a modular power and then steps its own run.  The integer and single-precision parts
 * are bit-exact; Box-Muller uses the device's double-precision log / cos, which are not guaranteed to round like
 * glibc's, so a quantised LLR can differ from the host generator's once in many millions (the host generator,
 * host/CChannel.cpp = oracle/frontend_oracle.c, stays the parity source; tests bound the mismatch rate).
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "lnsfaid.h"

#define FE_RUN 16 /* symbols (QPSK: LLR pairs, 16-QAM: LLR quadruples) per thread */

__device__ __forceinline__ uint32_t modpow(uint32_t a, unsigned long long e, uint32_t m)
{
    uint32_t r = 1u, b = a % m;
    while (e) {
        if (e & 1ull) r = (r * b) % m; /* operands < 2^16: no overflow */
        b = (b * b) % m;
        e >>= 1;
    }
    return r;
}

struct WH { uint32_t ix, iy, iz; };

__device__ __forceinline__ float wh_uniform(WH& s)
{
    s.ix = (s.ix * 249u) % 61967u;
    s.iy = (s.iy * 251u) % 63443u;
    s.iz = (s.iz * 252u) % 63599u;
    float temp = (((float)s.ix) / ((float)61967)) + (((float)s.iy) / ((float)63443)) + (((float)s.iz) / ((float)63599));
    temp -= (float)(int)temp;
    return temp;
}

__device__ __forceinline__ float wh_norm(double sigma, WH& s)
{
    const float u1 = wh_uniform(s);
    const float u2 = wh_uniform(s);
    return (float)(sigma * cos(2 * 3.1415926535897932384626433832795 * (double)u2) * sqrt(-2.0 * log(1.0 - (double)u1)));
}

__device__ __forceinline__ int8_t quantise_4bit(float x, float scale)
{
    const float y = x * scale;
    int q = (y > -2147483648.0f && y < 2147483648.0f) ? (int)y : (int)0x80000000; /* cvttps2dq */
    q = q > 127 ? 127 : (q < -128 ? -128 : q);                                      /* saturating packs */
    return (int8_t)(q > 7 ? 7 : (q < -7 ? -7 : q));
}

/* position of bit k of frame m inside one group's fixInput: [32][K] then [32][M] */
__device__ __forceinline__ size_t fix_pos(long pos, int n_var, int k_info, int n_check)
{
    const int m = (int)(pos / n_var), k = (int)(pos % n_var);
    return k < k_info ? (size_t)m * k_info + k : (size_t)32 * k_info + (size_t)m * n_check + (k - k_info);
}

__global__ __launch_bounds__(256) void lnsfaid_frontend_kernel(const uint32_t* __restrict__ states,
                                                               const unsigned long long* __restrict__ draws_before, int mod_type,
                                                               float sigma_ch, float scale, const int8_t* __restrict__ codeword,
                                                               const int8_t* __restrict__ frames, int n_var, int n_check,
                                                               int interleave, int8_t* __restrict__ fix_input)
{
    const int stream = (int)blockIdx.y;
    const long bits = 32L * n_var;
    const long symbols = bits / mod_type;
    const long first = ((long)blockIdx.x * 256 + threadIdx.x) * FE_RUN;
    if (first >= symbols) return;
    const int k_info = n_var - n_check;
    int8_t* out = fix_input + (size_t)stream * (size_t)bits;
    /* sent bit at position pos of the stream: its own 32 frames (encoder output layout = the fixInput layout), or one
     * codeword repeated in every frame (FakeEncoder), or all-zero */
    const int8_t* fr = frames ? frames + (size_t)stream * (size_t)bits : nullptr;
    auto tx = [&](long pos) -> int {
        if (fr) return fr[fix_pos(pos, n_var, k_info, n_check)];
        return codeword ? codeword[pos % n_var] : 0;
    };
    /* symbol i uses normals 2i and 2i+1, normal k uses uniforms 2k+1 and 2k+2 of the stream */
    const unsigned long long skip = draws_before[stream] + 4ull * (unsigned long long)first;
    /* generator state (IX, IY, IZ) of the stream at draws_before = 0: state after n draws = X0 * a^n mod m */
    const uint32_t x0 = states[3 * stream], y0 = states[3 * stream + 1], z0 = states[3 * stream + 2];
    WH s;
    s.ix = (uint32_t)(((unsigned long long)(x0 % 61967u) * modpow(249u, skip, 61967u)) % 61967u);
    s.iy = (uint32_t)(((unsigned long long)(y0 % 63443u) * modpow(251u, skip, 63443u)) % 63443u);
    s.iz = (uint32_t)(((unsigned long long)(z0 % 63599u) * modpow(252u, skip, 63599u)) % 63599u);
    const double sigma = (double)sigma_ch;
    const long last = first + FE_RUN < symbols ? first + FE_RUN : symbols;
    /* Modulation / Demodulation / (de)interleaver of reference CModulate.cpp:95-362 for QPSK, 16-, 64- and 256-QAM: symbol i
     * takes stream positions Q i .. Q i + Q - 1, position p of a frame carries code bit (N / I) (p mod I) + p div I. */
    const float t2[2] = { -0.707107f, 0.707107f };
    const float t4[4] = { -0.316228f, -0.948683f, 0.316228f, 0.948683f };
    const float t6[8] = { -0.462910f, -0.154303f, -0.771517f, -1.08012f, 0.462910f, 0.154303f, 0.771517f, 1.08012f };
    const float t8[16] = { -0.383482f, -0.536875f, -0.230089f, -0.076696f, -0.843661f, -0.690268f, -0.997054f, -1.150447f,
                           0.383482f, 0.536875f, 0.230089f, 0.076696f, 0.843661f, 0.690268f, 0.997054f, 1.150447f };
    const double f4[1] = { 0.6324555 }, f6[2] = { 0.6172134, 0.3086067 }, f8[3] = { 0.613568, 0.306784, 0.153392 };
    const int Q = mod_type, half = Q / 2;
    for (long i = first; i < last; ++i) {
        int idx_i = 0, idx_q = 0;
        long cpos[8]; /* frame * n_var + code bit of each position of the symbol */
        for (int u = 0; u < Q; ++u) {
            const long pos = (long)Q * i + u;
            const long m = pos / n_var, p = pos % n_var;
            cpos[u] = m * n_var + (n_var / interleave) * (p % interleave) + p / interleave;
            const int b = tx(cpos[u]);
            if (u & 1) idx_q += b << (half - u / 2 - 1); else idx_i += b << (half - u / 2 - 1);
        }
        const float ai = Q == 2 ? t2[idx_i] : Q == 4 ? t4[idx_i] : Q == 6 ? t6[idx_i] : t8[idx_i];
        const float aq = Q == 2 ? t2[idx_q] : Q == 4 ? t4[idx_q] : Q == 6 ? t6[idx_q] : t8[idx_q];
        float l[8];
        l[0] = wh_norm(sigma, s) + ai;
        l[1] = wh_norm(sigma, s) + aq;
        for (int n = 1; n < half; ++n) {
            const double c = Q == 4 ? f4[n - 1] : Q == 6 ? f6[n - 1] : f8[n - 1];
            l[2 * n] = (float)(fabs((double)l[2 * n - 2]) - c);
            l[2 * n + 1] = (float)(fabs((double)l[2 * n - 1]) - c);
        }
        for (int u = 0; u < Q; ++u) out[fix_pos(cpos[u], n_var, k_info, n_check)] = quantise_4bit(l[u], scale);
    }
}

extern "C" hipError_t lf_launch_frontend(const uint32_t* d_seeds, const unsigned long long* d_draws, int n_streams, int mod_type,
                                         float sigma_ch, float scale, const int8_t* d_codeword, const int8_t* d_frames, int n_var,
                                         int n_check, int interleave, int8_t* d_fix, hipStream_t stream)
{
    const long symbols = 32L * n_var / mod_type;
    const unsigned bx = (unsigned)((symbols + 256L * FE_RUN - 1) / (256L * FE_RUN));
    /* grid.y holds at most 65535 streams: more are launched in slices (pointers advanced per slice; the kernel indexes its
     * stream by blockIdx.y) */
    for (int s0 = 0; s0 < n_streams; s0 += 65535) {
        const int ns = n_streams - s0 < 65535 ? n_streams - s0 : 65535;
        hipLaunchKernelGGL(lnsfaid_frontend_kernel, dim3(bx, (unsigned)ns), dim3(256), 0, stream, d_seeds + 3 * (size_t)s0,
                           d_draws + s0, mod_type, sigma_ch, scale, d_codeword,
                           d_frames ? d_frames + (size_t)s0 * 32 * (size_t)n_var : nullptr, n_var, n_check, interleave,
                           d_fix + (size_t)s0 * 32 * (size_t)n_var);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
