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
    /* Profile.txt keys in file order: StartSNR, SNRPass, EndSNR, scale (quantiser), DecodeMethod (0 NMS, 1 OMS, 2 FAID + DTBF,
     * 3 OMS + BF, 4 OMS + DTBF, 5 FAID + 2B1C; reference README.md:13), MaxIteration, modType (1 BPSK, 2 QPSK, 4 16-QAM),
     * InterleaveModType, Factor_1, Factor_2, noFrames (32), code file name, Z (circulant size), ce (collect errors) */
    float snr_start = 0, snr_pass = 0, snr_end = 0, scale = 0;
    int decode_method = 0, Max_Iteration = 0, mod_type = 0, interleavemod_type = 0;
    int Factor_1 = 0, Factor_2 = 0, nb_frames = 0;
    std::string fileName;
    int Z = 0, ce = 0;
};

/* Reads ./Profile.txt (or `path`).  Returns false instead of the reference's getchar()+exit() so that the
 * caller decides; main() keeps the reference behaviour of terminating. */
bool ReadProfile(Parameter_Simulation* p, const char* path = "Profile.txt");

extern int collectflag;
#endif
