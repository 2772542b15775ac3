/*
 * frontend_oracle.c — CPU restatement of the reference's signal chain in front of the decoder,
 * for ONE worker thread, QPSK or 16-QAM, InterleaveModType 1.  TEST INFRASTRUCTURE ONLY (same rule as
 * lnsfaid_oracle.c).  It exists so that the oracle can be driven with exactly the LLR stream the
 * reference's own run saw (seed table CSimulate.cpp:11-17), which is what lets the counters recorded
 * in SURVEY.md §6 pin the oracle.
 *
 * Restated stages, in the order of CSimulate::Run (CSimulate.cpp:103-132):
 *   FakeEncoder            CLDPC.cpp:163-207   every one of the 32 frames carries the same codeword
 *   BeforeModulationInterleaver, InterleaveModType 1   CModulate.cpp:95-148   -> identity per frame
 *   Modulation             CModulate.cpp:216-264, table_qpsk / table_16qam CModulate.cpp:4-5
 *   AWGNChannel            CChannel.cpp:71-97   Wichmann-Hill triple + Box-Muller
 *   Demodulation           CModulate.cpp:273-293   QPSK: LLR = (real, imag); 16-QAM adds |real| - c, |imag| - c
 *   AfterDeModulationDeInterleaver   CModulate.cpp:152-212   -> [32][K] then [32][M]
 *   float2LimitChar_4bit   CLDPC.cpp:4524-4582
 */
#include "lnsfaid_oracle.h"

#include <math.h>

void lnsfaid_frontend_seed(lnsfaid_frontend* fe, int seed)
{
    fe->IX = fe->IY = fe->IZ = (unsigned long)seed; /* CChannel.cpp:121 */
}

float lnsfaid_frontend_sigma(float eb_n0_db, int mod_type, double rate)
{
    /* CSimulate.cpp:69-74; snr is a float member, the arithmetic is double */
    float snr = eb_n0_db;
    if (mod_type == 1) return (float)(1.0 / sqrt(2.0 * rate * mod_type * pow(10.0, 0.1 * snr)));
    return (float)(1.0 / sqrt(rate * mod_type * pow(10.0, 0.1 * snr)));
}

/* CChannel::Random_Uniform (CChannel.cpp:71-80): float arithmetic throughout */
static float random_uniform(lnsfaid_frontend* rs)
{
    float temp = 0.0;
    rs->IX = (rs->IX * 249) % 61967;
    rs->IY = (rs->IY * 251) % 63443;
    rs->IZ = (rs->IZ * 252) % 63599;
    temp = (((float)rs->IX) / ((float)61967)) + (((float)rs->IY) / ((float)63443)) + (((float)rs->IZ) / ((float)63599));
    temp -= (int)temp;
    return temp;
}

/* CChannel::Random_Norm (CChannel.cpp:82-89): double arithmetic, result rounded to float */
static float random_norm(double sigma, lnsfaid_frontend* rs)
{
    float u1, u2, u;
    u1 = random_uniform(rs);
    u2 = random_uniform(rs);
    u = sigma * cos(2 * 3.1415926535897932384626433832795 * u2) * sqrt(-2.0 * log(1.0 - u1));
    return u;
}

/* float2LimitChar_4bit (CLDPC.cpp:4553-4573): float multiply, truncate toward zero (cvttps2dq),
 * saturating packs to int8, clamp to [-7, 7]. */
static int8_t quantise_4bit(float x, float scale)
{
    float y = x * scale;
    int q;
    if (!(y > -2147483648.0f && y < 2147483648.0f)) q = (int)0x80000000; /* cvttps2dq "integer indefinite" */
    else q = (int)y;
    if (q > 127) q = 127;   /* packs_epi32 + packs_epi16 */
    if (q < -128) q = -128;
    if (q > 7) q = 7;
    if (q < -7) q = -7;
    return (int8_t)q;
}

/* QPSK with 32 different frames ([32][n_var] bits, frame-major): what the driver sends with a real encoder
 * (reference CSimulate.cpp:106-107).  frame_stride = 0 repeats one codeword (FakeEncoder). */
static void qpsk_group(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* bits, int frame_stride, float sigma,
                       float scale, int8_t* fixInput);

void lnsfaid_frontend_qpsk_frames(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* frames, float sigma,
                                  float scale, int8_t* fixInput)
{
    qpsk_group(fe, n_var, n_check, frames, n_var, sigma, scale, fixInput);
}

void lnsfaid_frontend_qpsk_group(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* codeword, float sigma,
                                 float scale, int8_t* fixInput)
{
    qpsk_group(fe, n_var, n_check, codeword, 0, sigma, scale, fixInput);
}

static void qpsk_group(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* codeword, int frame_stride, float sigma,
                       float scale, int8_t* fixInput)
{
    static const float table_qpsk[2] = { -0.707107f, 0.707107f }; /* CModulate.cpp:4 */
    const int K = n_var - n_check;
    /* AWGNChannel(modulate->ModSeq, sigma / sqrt(2)) (CSimulate.cpp:126): float / double -> double,
     * narrowed to the float parameter of AWGNChannel, widened again for Random_Norm(double). */
    const float sigma_ch = (float)(sigma / sqrt(2));
    /* bit k of frame m sits at m*n_var + k in InterLeaveSeq; symbol i carries bits 2i (real), 2i+1 (imag)
     * (CModulate.cpp:253-259 with half_sym = 1); AWGNChannel draws real then imag (CChannel.cpp:94-95). */
    for (int m = 0; m < 32; ++m)
        for (int k = 0; k < n_var; ++k) {
            int bit = codeword ? codeword[(size_t)m * frame_stride + k] : 0;
            float rx = random_norm(sigma_ch, fe) + table_qpsk[bit];
            int8_t q = quantise_4bit(rx, scale);
            if (k < K) fixInput[(size_t)m * K + k] = q;
            else fixInput[(size_t)32 * K + (size_t)m * n_check + (k - K)] = q;
        }
}

/* Same for 16-QAM (modType 4): symbol i carries bits 4i..4i+3, I index = 2*b0 + b2, Q index = 2*b1 + b3
 * (CModulate.cpp:253-259 with half_sym = 2), max-log demapper of CModulate.cpp:283-293. */
void lnsfaid_frontend_qam16_group(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* codeword, float sigma,
                                  float scale, int8_t* fixInput)
{
    static const float table_16qam[4] = { -0.316228f, -0.948683f, 0.316228f, 0.948683f }; /* CModulate.cpp:5 */
    const int K = n_var - n_check;
    const float sigma_ch = (float)(sigma / sqrt(2));
    const long total = 32L * n_var; /* frame m, bit k at m * n_var + k */
    for (long i = 0; i < total / 4; ++i) {
        int b[4];
        for (int u = 0; u < 4; ++u) b[u] = codeword ? codeword[(4 * i + u) % n_var] : 0;
        const float re = random_norm(sigma_ch, fe) + table_16qam[2 * b[0] + b[2]];
        const float im = random_norm(sigma_ch, fe) + table_16qam[2 * b[1] + b[3]];
        float llr[4];
        llr[0] = re;
        llr[1] = im;
        llr[2] = fabs(re) - 0.6324555; /* double arithmetic, stored as float (CModulate.cpp:290-291) */
        llr[3] = fabs(im) - 0.6324555;
        for (int u = 0; u < 4; ++u) {
            const long pos = 4 * i + u;
            const int m = (int)(pos / n_var), k = (int)(pos % n_var);
            const int8_t q = quantise_4bit(llr[u], scale);
            if (k < K) fixInput[(size_t)m * K + k] = q;
            else fixInput[(size_t)32 * K + (size_t)m * n_check + (k - K)] = q;
        }
    }
}

/* ---- any modulation order the reference maps (QPSK, 16-, 64-, 256-QAM) and its block interleaver ----------------
 * BeforeModulationInterleaver (CModulate.cpp:134-146): inside every frame, position p = I * i + j carries code bit
 * k = (B / I) * j' ... precisely InterLeaveSeq[m B + j I + i] = ILSeq[m B + (B / I) i + j], I = InterleaveModType,
 * B = bits per frame; AfterDeModulationDeInterleaver (:152-166) is its inverse.  Modulation (:216-264): symbol s takes
 * positions M s .. M s + M - 1; the even ones index the in-phase table entry MSB first, the odd ones the quadrature
 * entry.  Demodulation (:273-362): max-log, l0 = real, l1 = imag, l(2n) = |l(2n-2)| - c_n, l(2n+1) = |l(2n-1)| - c_n with
 * every level stored as float before it feeds the next. */
static const float k_table_qpsk[2] = { -0.707107f, 0.707107f };                                   /* CModulate.cpp:4 */
static const float k_table_16qam[4] = { -0.316228f, -0.948683f, 0.316228f, 0.948683f };           /* :5 */
static const float k_table_64qam[8] = { -0.462910f, -0.154303f, -0.771517f, -1.08012f, 0.462910f, 0.154303f, 0.771517f, 1.08012f }; /* :6 */
static const float k_table_256qam[16] = { -0.383482f, -0.536875f, -0.230089f, -0.076696f, -0.843661f, -0.690268f, -0.997054f, -1.150447f,
                                          0.383482f, 0.536875f, 0.230089f, 0.076696f, 0.843661f, 0.690268f, 0.997054f, 1.150447f }; /* :7 */

int lnsfaid_frontend_group(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* bits, int frame_stride, int mod_type,
                           int interleave, float sigma, float scale, int8_t* fixInput)
{
    static const double c16[1] = { 0.6324555 }, c64[2] = { 0.6172134, 0.3086067 }, c256[3] = { 0.613568, 0.306784, 0.153392 };
    const float* table = mod_type == 2 ? k_table_qpsk : mod_type == 4 ? k_table_16qam : mod_type == 6 ? k_table_64qam
                       : mod_type == 8 ? k_table_256qam : 0;
    const double* cl = mod_type == 4 ? c16 : mod_type == 6 ? c64 : c256;
    const long total = 32L * n_var;
    if (!table || interleave < 1 || n_var % interleave != 0 || total % mod_type != 0) return -1;
    const int K = n_var - n_check, half = mod_type / 2;
    const float sigma_ch = (float)(sigma / sqrt(2));
    for (long s = 0; s < total / mod_type; ++s) {
        int idx_i = 0, idx_q = 0;
        long kk[8];
        for (int u = 0; u < mod_type; ++u) {
            const long pos = (long)mod_type * s + u;
            const int m = (int)(pos / n_var), p = (int)(pos % n_var);
            const int k = (n_var / interleave) * (p % interleave) + p / interleave; /* code bit at this position */
            kk[u] = (long)m * n_var + k;
            const int b = bits ? bits[(size_t)m * frame_stride + k] : 0;
            if (u & 1) idx_q += b << (half - u / 2 - 1); else idx_i += b << (half - u / 2 - 1);
        }
        float l[8];
        l[0] = random_norm(sigma_ch, fe) + table[idx_i];
        l[1] = random_norm(sigma_ch, fe) + table[idx_q];
        for (int n = 1; n < half; ++n) {
            l[2 * n] = fabs(l[2 * n - 2]) - cl[n - 1];
            l[2 * n + 1] = fabs(l[2 * n - 1]) - cl[n - 1];
        }
        for (int u = 0; u < mod_type; ++u) {
            const int m = (int)(kk[u] / n_var), k = (int)(kk[u] % n_var);
            const int8_t q = quantise_4bit(l[u], scale);
            if (k < K) fixInput[(size_t)m * K + k] = q;
            else fixInput[(size_t)32 * K + (size_t)m * n_check + (k - K)] = q;
        }
    }
    return 0;
}
