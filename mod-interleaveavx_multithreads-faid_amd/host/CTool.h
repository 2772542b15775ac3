/*
 * CTool.h — run-time configuration of the host driver.
 * Mirrors the reference's Parameter_Simulation / ReadProfile (reference CTool.h:23-39, CTool.cpp:588-621):
 * same Profile.txt, same keys in the same positional order, same field names.
 */
#ifndef CTOOL_H
#define CTOOL_H
#include <string>

#define REGULAR_COL_WEIGHT 3 /* reference CTool.h:6 */

struct Parameter_Simulation {
    float snr_start; /* StartSNR */
    float snr_pass;  /* SNRPass  */
    float snr_end;   /* EndSNR   */
    float scale;     /* scale: quantiser scale */
    int decode_method; /* DecodeMethod: 1 OMS, 2 FAID+DTBF, 3 OMS+BF, 4 OMS+DTBF, 5 FAID+2B1C (reference README.md:13) */
    int Max_Iteration; /* MaxIteration */
    int mod_type;      /* modType: 1 BPSK, 2 QPSK */
    int interleavemod_type; /* InterleaveModType */
    int Factor_1;
    int Factor_2;
    int nb_frames;     /* noFrames: 32 */
    std::string fileName;
    int Z;
    int ce;
};

/* Reads ./Profile.txt (or `path`).  Returns false instead of the reference's getchar()+exit() so that the
 * caller decides; main() keeps the reference behaviour of terminating. */
bool ReadProfile(Parameter_Simulation* p, const char* path = "Profile.txt");

extern int collectflag;
#endif
