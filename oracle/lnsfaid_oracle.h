/*
 * lnsfaid_oracle.h — CPU oracle for the decode hot path.  TEST INFRASTRUCTURE ONLY (see the header
 * of lnsfaid_oracle.c): never included or linked by the product path.
 * Uses the product's public structs (include/lnsfaid.h) so both sides see identical inputs.
 */
#ifndef LNSFAID_ORACLE_H
#define LNSFAID_ORACLE_H
#include "lnsfaid.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lnsfaid_oracle lnsfaid_oracle;

int lnsfaid_oracle_create(lnsfaid_oracle** out, const lnsfaid_code* code, const lnsfaid_cfg* cfg);
void lnsfaid_oracle_destroy(lnsfaid_oracle* o);
/* same buffers and layout as lnsfaid_decode() */
int lnsfaid_oracle_decode(lnsfaid_oracle* o, const int8_t* fixInput, size_t n_groups, int8_t* decodedBits,
                          lnsfaid_group_stats* stats);
/* test hook: n_iter layered iterations of one group without early stop, En per lane (en_out[l * N + v]) */
int lnsfaid_oracle_layered_en(lnsfaid_oracle* o, const int8_t* fixInput, int n_iter, int8_t* en_out);
int lnsfaid_oracle_count_errors(const lnsfaid_code* code, const int8_t* decodedBits, const int8_t* inputBits,
                                size_t n_groups, uint64_t out[4]);

/* ---- vectorised CPU port (lnsfaid_cpu_avx2.c): bench.py's cpu_baseline; validated against the oracle ---- */
typedef struct lnsfaid_cpu lnsfaid_cpu;
int lnsfaid_cpu_create(lnsfaid_cpu** out, const lnsfaid_code* code, const lnsfaid_cfg* cfg);
void lnsfaid_cpu_destroy(lnsfaid_cpu* o);
int lnsfaid_cpu_decode(lnsfaid_cpu* o, const int8_t* fixInput, size_t n_groups, int8_t* decodedBits,
                       lnsfaid_group_stats* stats);

/* ---- front-end restatement (frontend_oracle.c): the reference's channel for one worker thread --- */
typedef struct lnsfaid_frontend {
    unsigned long IX, IY, IZ; /* Wichmann-Hill state, RandSeed (CChannel.h:15-20) */
} lnsfaid_frontend;

/* CChannel::Initial without CONTINUE_SEED (CChannel.cpp:121): IX = IY = IZ = seed */
void lnsfaid_frontend_seed(lnsfaid_frontend* fe, int seed);
/* sigma of CSimulate::Configure (CSimulate.cpp:69-74) for modulation order mod_type (1 BPSK, 2 QPSK) */
float lnsfaid_frontend_sigma(float eb_n0_db, int mod_type, double rate);
/* One pass of the loop body of CSimulate::Run (CSimulate.cpp:126-132) for QPSK, InterleaveModType 1,
 * transmitted codeword `codeword` ([n_var] bits 0/1, NULL = all-zero, same for all 32 frames as
 * FakeEncoder does): AWGNChannel(sigma/sqrt(2)) -> Demodulation -> AfterDeModulationDeInterleaver ->
 * float2LimitChar_4bit.  Writes one group of fixInput ([32][K] then [32][M]). */
void lnsfaid_frontend_qpsk_group(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* codeword, float sigma,
                                 float scale, int8_t* fixInput);

/* the same loop body for 16-QAM, modType 4 (table_16qam CModulate.cpp:5, demapper CModulate.cpp:283-293) */
/* same with 32 different frames, frames = [32][n_var] bits */
void lnsfaid_frontend_qpsk_frames(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* frames, float sigma,
                                  float scale, int8_t* fixInput);
/* General form: modulation order mod_type in {2, 4, 6, 8} (QPSK, 16-, 64-, 256-QAM, CModulate.cpp:4-7, :216-362), block
 * interleaver InterleaveModType = interleave >= 1 (CModulate.cpp:95-212), bits = sent code bits with `frame_stride`
 * between frames (0: one codeword for all 32 frames; NULL: all-zero).  Returns -1 for sizes the reference cannot map. */
int lnsfaid_frontend_group(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* bits, int frame_stride, int mod_type,
                           int interleave, float sigma, float scale, int8_t* fixInput);
void lnsfaid_frontend_qam16_group(lnsfaid_frontend* fe, int n_var, int n_check, const int8_t* codeword, float sigma,
                                  float scale, int8_t* fixInput);

#ifdef __cplusplus
}
#endif
#endif
