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

#define FE_RUN 32 /* symbols (QPSK: LLR pairs, 16-QAM: LLR quadruples) per thread */

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

__device__ __forceinline__ int8_t quantise_4bit(float x, float scale)
{
    const float y = x * scale;
    int q = (y > -2147483648.0f && y < 2147483648.0f) ? (int)y : (int)0x80000000; /* cvttps2dq */
    q = q > 127 ? 127 : (q < -128 ? -128 : q);                                      /* saturating packs */
    return (int8_t)(q > 7 ? 7 : (q < -7 ? -7 : q));
}

/* ---- the fast path ------------------------------------------------------------------------------------------------
 * The chain above costs ~330 instructions per normal, nearly all of it the double-precision log / cos / sqrt and the IEEE float
 * divisions of the uniforms - for a result that is cut down to 4 bits.  The fast path computes the SAME uniforms (exactly: the
 * generator state and the three quotients are bit-exact in single precision, see below), the normal in single precision with the
 * hardware's transcendentals, and an upper bound of its distance from the double-precision value; the quantised LLR is taken from
 * it only when no quantiser threshold lies within that bound, otherwise the symbol is recomputed by the chain above (a few symbols
 * in ten thousand).  Output is therefore identical to the slow path's by construction; tests/test_gpu_frontend.py checks both
 * the identity on large batches and the error bounds it rests on (lnsfaid_frontend_fastpath_bounds scans EVERY float in [0, 1)).
 *
 * Exact pieces (exhaustive CPU test, tests/test_frontend_fastpath.py):
 *  - x * a mod m in float: x * a < 2^24 is exact, floor((x * a) * fl(1 / m)) is the true quotient for every x < m of the three
 *    generators, the remainder is one fma;
 *  - x / m correctly rounded for every x < m: q0 = x * fl(1 / m), e = fma(-q0, m, x), q = fma(e, fl(1 / m), q0). */
#define FE_M1 61967.0f
#define FE_M2 63443.0f
#define FE_M3 63599.0f
struct WHF { float ix, iy, iz; }; /* the generator state as floats (integers below 2^16) */

__device__ __forceinline__ float whf_mulmod(float x, float a, float m)
{
    const float cr = 1.0f / m; /* folded at compile time, correctly rounded */
    const float p = x * a;
    const float q = floorf(p * cr);
    return fmaf(-q, m, p);
}
__device__ __forceinline__ float whf_div(float x, float m)
{
    const float cr = 1.0f / m;
    const float q0 = x * cr;
    const float e = fmaf(-q0, m, x);
    return fmaf(e, cr, q0);
}
__device__ __forceinline__ float whf_uniform(WHF& s)
{
    s.ix = whf_mulmod(s.ix, 249.0f, FE_M1);
    s.iy = whf_mulmod(s.iy, 251.0f, FE_M2);
    s.iz = whf_mulmod(s.iz, 252.0f, FE_M3);
    float temp = (whf_div(s.ix, FE_M1) + whf_div(s.iy, FE_M2)) + whf_div(s.iz, FE_M3);
    temp -= truncf(temp);
    return temp;
}

/* sqrt(-2 ln(1 - u1)) and cos(2 pi u2) in single precision.  FE_EPS_R / FE_EPS_C bound their absolute distance from the
 * double-precision values over every float in [0, 1) (measured by fastpath_scan_kernel below measured on gfx950: 4.9e-7 and 1.3e-7; the GPU
 * test fails if a device ever exceeds half of these constants). */
#define FE_EPS_R 1.5e-6f
#define FE_EPS_C 4.0e-7f
__device__ __forceinline__ float fast_radius(float u1)
{
    /* -ln(1 - u): below 2^-6 the series (1 - u is not exact in float there), else log2 of the exactly rounded difference */
    const float ser = u1 * (1.0f + u1 * (0.5f + u1 * (0.33333334f + u1 * (0.25f + u1 * 0.2f))));
    const float lg = -0.69314718f * __builtin_amdgcn_logf(1.0f - u1);
    const float L = u1 < 0.015625f ? ser : lg;
    return __builtin_amdgcn_sqrtf(L + L);
}
__device__ __forceinline__ float fast_cos2pi(float u2) { return __builtin_amdgcn_cosf(u2); } /* v_cos_f32 takes revolutions */

/* Exhaustive scan of the two functions over all floats in [0, 1) against double precision (out: max |dR|, max |dc|, and where) */
__global__ __launch_bounds__(256) void fastpath_scan_kernel(double* out)
{
    double worst_r = 0.0, worst_c = 0.0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * 256 + threadIdx.x; b < 0x3f800000ull; b += (unsigned long long)gridDim.x * 256) {
        const float u = __uint_as_float((uint32_t)b);
        const double r = sqrt(-2.0 * log(1.0 - (double)u));
        const double c = cos(2 * 3.1415926535897932384626433832795 * (double)u);
        const double dr = fabs((double)fast_radius(u) - r), dc = fabs((double)fast_cos2pi(u) - c);
        worst_r = dr > worst_r ? dr : worst_r;
        worst_c = dc > worst_c ? dc : worst_c;
    }
    /* values are non-negative: their bit patterns order like integers */
    atomicMax((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(worst_r));
    atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(worst_c));
}

extern "C" hipError_t lf_frontend_fastpath_scan(double* d_out2, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_out2, 0, 2 * sizeof(double), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fastpath_scan_kernel, dim3(4096), dim3(256), 0, stream, d_out2);
    return hipGetLastError();
}
extern "C" void lf_frontend_fastpath_assumed(double* eps2) { eps2[0] = (double)FE_EPS_R; eps2[1] = (double)FE_EPS_C; }

/* true: no quantiser threshold (the integers +-1 .. +-7 of y = l * scale; beyond +-7 the value is clamped, inside (-1, 1) it truncates to 0)
 * within dy of y */
__device__ __forceinline__ bool quantiser_certain(float y, float dy)
{
    const float r = rintf(y);
    const float ar = fabsf(r);
    return !(fabsf(y - r) <= dy && ar >= 1.0f && ar <= 7.0f);
}

/* x * A^e mod M for e < 2^16: the powers A^(2^k) fold to constants once the loop is unrolled */
template <uint32_t A, uint32_t M>
__device__ __forceinline__ uint32_t jump_small(uint32_t x, uint32_t e)
{
    uint32_t b = A % M;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if ((e >> k) & 1u) x = (x * b) % M; /* operands < 2^16 */
        b = (b * b) % M;
    }
    return x;
}

/* MOD: Profile.txt modType (bits per symbol); FAST: the fast path above / every symbol through the double-precision chain */
template <int MOD, bool FAST>
__global__ __launch_bounds__(256) void lnsfaid_frontend_kernel(const uint32_t* __restrict__ states,
                                                               const unsigned long long* __restrict__ draws_before,
                                                               float sigma_ch, float scale, const int8_t* __restrict__ codeword,
                                                               const int8_t* __restrict__ frames, int n_var, int n_check,
                                                               int interleave, int8_t* __restrict__ fix_input)
{
    constexpr bool fast = FAST;
    /* (all positions of a stream fit 32 bits: 32 n_var <= 2^21) */
    const int stream = (int)blockIdx.y;
    const uint32_t N = (uint32_t)n_var, M = (uint32_t)n_check, K = N - M, I = (uint32_t)interleave;
    constexpr uint32_t Q = (uint32_t)MOD, half = Q / 2;
    const uint32_t bits = 32u * N;
    const uint32_t symbols = bits / Q;
    const uint32_t first = (blockIdx.x * 256u + threadIdx.x) * FE_RUN;
    if (first >= symbols) return;
    int8_t* out = fix_input + (size_t)stream * (size_t)bits;
    /* sent bit: the stream's own 32 frames (encoder output layout = the fixInput layout), or one codeword repeated in every frame
     * (FakeEncoder), or all-zero */
    const int8_t* fr = frames ? frames + (size_t)stream * (size_t)bits : nullptr;

    /* Generator state of the thread's first draw.  Symbol i uses normals 2i and 2i+1, normal k uniforms 2k+1 and 2k+2 of the stream;
     * the state after n draws is X0 * a^n mod m.  The exponent is split into the block's part (uniform: scalar unit) and the
     * thread's (< 2^16: constant powers). */
    const unsigned long long skip_block = draws_before[stream] + 4ull * FE_RUN * 256ull * (unsigned long long)blockIdx.x;
    const uint32_t skip_thread = 4u * FE_RUN * threadIdx.x;
    const uint32_t x0 = states[3 * stream], y0 = states[3 * stream + 1], z0 = states[3 * stream + 2];
    WH si; /* fast == 0: the generator in integers and IEEE divisions, exactly as written in the reference */
    si.ix = jump_small<249u, 61967u>((uint32_t)(((unsigned long long)(x0 % 61967u) * modpow(249u, skip_block, 61967u)) % 61967u), skip_thread);
    si.iy = jump_small<251u, 63443u>((uint32_t)(((unsigned long long)(y0 % 63443u) * modpow(251u, skip_block, 63443u)) % 63443u), skip_thread);
    si.iz = jump_small<252u, 63599u>((uint32_t)(((unsigned long long)(z0 % 63599u) * modpow(252u, skip_block, 63599u)) % 63599u), skip_thread);
    WHF sf;
    sf.ix = (float)si.ix; sf.iy = (float)si.iy; sf.iz = (float)si.iz;
    const double sigma = (double)sigma_ch;
    const uint32_t last = first + FE_RUN < symbols ? first + FE_RUN : symbols;
    /* Modulation / Demodulation / (de)interleaver of reference CModulate.cpp:95-362 for QPSK, 16-, 64- and 256-QAM: symbol i
     * takes stream positions Q i .. Q i + Q - 1, position p of a frame carries code bit (N / I) (p mod I) + p div I. */
    const float t2[2] = { -0.707107f, 0.707107f };
    const float t4[4] = { -0.316228f, -0.948683f, 0.316228f, 0.948683f };
    const float t6[8] = { -0.462910f, -0.154303f, -0.771517f, -1.08012f, 0.462910f, 0.154303f, 0.771517f, 1.08012f };
    const float t8[16] = { -0.383482f, -0.536875f, -0.230089f, -0.076696f, -0.843661f, -0.690268f, -0.997054f, -1.150447f,
                           0.383482f, 0.536875f, 0.230089f, 0.076696f, 0.843661f, 0.690268f, 0.997054f, 1.150447f };
    const double f4[1] = { 0.6324555 }, f6[2] = { 0.6172134, 0.3086067 }, f8[3] = { 0.613568, 0.306784, 0.153392 };
    /* where the thread's next LLR goes: frame fm, position fp inside it, and fp split by the interleaver (fp = fd I + fi); kept by
     * increments, so the loop has no division */
    const uint32_t pos0 = Q * first;
    uint32_t fm = pos0 / N, fp = pos0 - fm * N;
    uint32_t fd = fp / I, fi = fp - fd * I;
    const uint32_t stride = N / I;
    /* QPSK without interleaver (the shipped Profile.txt): a thread's 2 FE_RUN LLRs are consecutive bytes of one frame part and
     * start on a multiple of 2 FE_RUN (K, M and N are), so they leave as 16-byte stores, one per 8 symbols, instead of single bytes */
    const bool packed = Q == 2 && I == 1 && (K % (2 * FE_RUN)) == 0 && (N % (2 * FE_RUN)) == 0 && last - first == FE_RUN && ((size_t)out % 16) == 0;
    const uint32_t out0 = fp < K ? fm * K + fp : 32u * K + fm * M + (fp - K);
    /* higher orders without interleaver: the Q LLRs of a symbol are consecutive bytes of one frame part when Q divides K and N,
     * and leave as one 4- / 8-byte store (three 2-byte stores for 64-QAM) */
    const bool wide = Q > 2 && I == 1 && (K % Q) == 0 && (N % Q) == 0 && ((size_t)out % 8) == 0;
    uint32_t pack[4] = { 0u, 0u, 0u, 0u };
#pragma unroll 1
    for (uint32_t i = first; i < last; ++i) {
        uint32_t idx_i = 0, idx_q = 0;
        uint32_t cidx[8]; /* index of each LLR of the symbol in the stream's fixInput ([32][K] then [32][M]) */
#pragma unroll
        for (uint32_t u = 0; u < Q; ++u) {
            const uint32_t kbit = stride * fi + fd; /* code bit carried by this position */
            cidx[u] = kbit < K ? fm * K + kbit : 32u * K + fm * M + (kbit - K);
            const uint32_t b = fr ? (uint32_t)fr[cidx[u]] : (codeword ? (uint32_t)codeword[kbit] : 0u);
            if (u & 1u) idx_q += b << (half - u / 2 - 1); else idx_i += b << (half - u / 2 - 1);
            if (++fi == I) { fi = 0; ++fd; }
            if (++fp == N) { fp = 0; fi = 0; fd = 0; ++fm; }
        }
        const float ai = Q == 2 ? t2[idx_i] : Q == 4 ? t4[idx_i] : Q == 6 ? t6[idx_i] : t8[idx_i];
        const float aq = Q == 2 ? t2[idx_q] : Q == 4 ? t4[idx_q] : Q == 6 ? t6[idx_q] : t8[idx_q];
        /* the four uniforms of the symbol */
        float u1a, u2a, u1b, u2b;
        if (fast) { u1a = whf_uniform(sf); u2a = whf_uniform(sf); u1b = whf_uniform(sf); u2b = whf_uniform(sf); }
        else { u1a = wh_uniform(si); u2a = wh_uniform(si); u1b = wh_uniform(si); u2b = wh_uniform(si); }
        float l[8];
        bool certain = fast != 0;
        if (fast) {
            /* n = fl(fl(sigma c) R) against sigma cos() sqrt() in double, narrowed to float:
             * |dn| <= sigma (eps_c R + eps_R) + roundings; then l = n + a and y = l scale, one more rounding each */
            const float ra = fast_radius(u1a), ca = fast_cos2pi(u2a), rb = fast_radius(u1b), cb = fast_cos2pi(u2b);
            const float na = (sigma_ch * ca) * ra, nb = (sigma_ch * cb) * rb;
            l[0] = na + ai;
            l[1] = nb + aq;
            const float dl = sigma_ch * (FE_EPS_C * fmaxf(ra, rb) + FE_EPS_R)
                             + 4.0e-7f * (fmaxf(fabsf(na), fabsf(nb)) + fmaxf(fabsf(l[0]), fabsf(l[1])) + 1.0f);
#pragma unroll
            for (uint32_t n = 1; n < half; ++n) { /* the same double expressions as below on the approximate levels: 1-Lipschitz */
                const double c = Q == 4 ? f4[n - 1] : Q == 6 ? f6[n - 1] : f8[n - 1];
                l[2 * n] = (float)(fabs((double)l[2 * n - 2]) - c);
                l[2 * n + 1] = (float)(fabs((double)l[2 * n - 1]) - c);
            }
#pragma unroll
            for (uint32_t u = 0; u < Q; ++u) {
                const float y = l[u] * scale;
                /* (every level adds one float rounding of a value below 2) */
                certain = certain && quantiser_certain(y, fabsf(scale) * (dl + 2.5e-7f * (float)(u / 2)) + 2.5e-7f * fabsf(y));
            }
        }
        if (!certain) { /* the reference's chain in double precision */
            const double kTwoPi = 2 * 3.1415926535897932384626433832795;
            l[0] = (float)(sigma * cos(kTwoPi * (double)u2a) * sqrt(-2.0 * log(1.0 - (double)u1a))) + ai;
            l[1] = (float)(sigma * cos(kTwoPi * (double)u2b) * sqrt(-2.0 * log(1.0 - (double)u1b))) + aq;
#pragma unroll
            for (uint32_t n = 1; n < half; ++n) {
                const double c = Q == 4 ? f4[n - 1] : Q == 6 ? f6[n - 1] : f8[n - 1];
                l[2 * n] = (float)(fabs((double)l[2 * n - 2]) - c);
                l[2 * n + 1] = (float)(fabs((double)l[2 * n - 1]) - c);
            }
        }
        if (packed) {
            const uint32_t k = i - first;
            const uint32_t two = (uint32_t)(uint8_t)quantise_4bit(l[0], scale) | ((uint32_t)(uint8_t)quantise_4bit(l[1], scale) << 8);
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w)
                if (w == ((k & 7u) >> 1)) pack[w] |= two << (16 * (k & 1u));
            if ((k & 7u) == 7u) {
                *(uint4*)(out + out0 + 2u * (k - 7u)) = make_uint4(pack[0], pack[1], pack[2], pack[3]);
                pack[0] = pack[1] = pack[2] = pack[3] = 0u;
            }
        } else if (wide) {
            uint32_t w[2] = { 0u, 0u };
#pragma unroll
            for (uint32_t u = 0; u < Q; ++u) w[u / 4] |= (uint32_t)(uint8_t)quantise_4bit(l[u], scale) << (8 * (u % 4));
            int8_t* dst = out + cidx[0];
            if (Q == 4) *(uint32_t*)dst = w[0];
            else if (Q == 8) *(uint2*)dst = make_uint2(w[0], w[1]);
            else { *(uint16_t*)dst = (uint16_t)w[0]; *(uint16_t*)(dst + 2) = (uint16_t)(w[0] >> 16); *(uint16_t*)(dst + 4) = (uint16_t)w[1]; }
        } else {
#pragma unroll
            for (uint32_t u = 0; u < Q; ++u) out[cidx[u]] = quantise_4bit(l[u], scale);
        }
    }
}

extern "C" hipError_t lf_launch_frontend(const uint32_t* d_seeds, const unsigned long long* d_draws, int n_streams, int mod_type,
                                         float sigma_ch, float scale, const int8_t* d_codeword, const int8_t* d_frames, int n_var,
                                         int n_check, int interleave, int fast, int8_t* d_fix, hipStream_t stream)
{
    const long symbols = 32L * n_var / mod_type;
    const unsigned bx = (unsigned)((symbols + 256L * FE_RUN - 1) / (256L * FE_RUN));
    /* grid.y holds at most 65535 streams: more are launched in slices (pointers advanced per slice; the kernel indexes its
     * stream by blockIdx.y) */
    for (int s0 = 0; s0 < n_streams; s0 += 65535) {
        const int ns = n_streams - s0 < 65535 ? n_streams - s0 : 65535;
        const dim3 grid(bx, (unsigned)ns), block(256);
        const uint32_t* seeds = d_seeds + 3 * (size_t)s0;
        const unsigned long long* draws = d_draws + s0;
        const int8_t* fr = d_frames ? d_frames + (size_t)s0 * 32 * (size_t)n_var : nullptr;
        int8_t* fix = d_fix + (size_t)s0 * 32 * (size_t)n_var;
#define FE_LAUNCH(MOD)                                                                                                             \
    case MOD:                                                                                                                      \
        if (fast) hipLaunchKernelGGL((lnsfaid_frontend_kernel<MOD, true>), grid, block, 0, stream, seeds, draws, sigma_ch, scale, d_codeword, fr, n_var, n_check, interleave, fix); \
        else hipLaunchKernelGGL((lnsfaid_frontend_kernel<MOD, false>), grid, block, 0, stream, seeds, draws, sigma_ch, scale, d_codeword, fr, n_var, n_check, interleave, fix);     \
        break;
        switch (mod_type) {
            FE_LAUNCH(2) FE_LAUNCH(4) FE_LAUNCH(6) FE_LAUNCH(8)
        default: return hipErrorInvalidValue;
        }
#undef FE_LAUNCH
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
