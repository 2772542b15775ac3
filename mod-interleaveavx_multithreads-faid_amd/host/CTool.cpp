#include "CTool.h"

#include <fstream>

int collectflag = 0;

bool ReadProfile(Parameter_Simulation* p, const char* path)
{
    std::ifstream fin(path);
    if (!fin.is_open()) return false;
    std::string rub;
    /* positional parser, reference CTool.cpp:598-616: every value is preceded by one key token, section
     * titles are two tokens */
    fin >> rub >> rub;                 /* "Simulation parameter"   */
    fin >> rub >> p->snr_start;        /* StartSNR:                */
    fin >> rub >> p->snr_pass;         /* SNRPass:                 */
    fin >> rub >> p->snr_end;          /* EndSNR:                  */
    fin >> rub >> p->decode_method;    /* DecodeMethod:            */
    fin >> rub >> p->Max_Iteration;    /* MaxIteration:            */
    fin >> rub >> rub;                 /* "Modulation Parameter:"  */
    fin >> rub >> p->mod_type;         /* modType:                 */
    fin >> rub >> p->interleavemod_type; /* InterleaveModType:     */
    fin >> rub >> rub;                 /* "NMS  Factor:"           */
    fin >> rub >> p->Factor_1;
    fin >> rub >> p->Factor_2;
    fin >> rub >> p->nb_frames;        /* noFrames:                */
    fin >> rub >> p->scale;
    fin >> rub >> rub;                 /* "Matrix Factor"          */
    fin >> rub >> p->fileName;         /* FileName: (the reference then overwrites it with MATRIX_FILE) */
    fin >> rub >> p->Z;
    p->ce = 0;
    return !fin.fail();
}
