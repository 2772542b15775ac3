/*
 * lnsfaid_tables.c — built-in code definition and the reference's shipped decoder constants.
 *
 * (1) The 50G-PON mother code as a 12 x 69 base matrix of circulant shifts (Z = 256).  The reference
 *     stores the same code expanded to 70400 VN indices in `PosNoeudsVariable`
 *     (Constants/50GPON-dc-original/Constants_SSE.h:29-3102); row br*256+i of that table is
 *     { cb*256 + ((shift + i) mod 256) } in ascending cb.  lnsfaid_code_50gpon()
 *     regenerates the table in that exact order; tests pin it by SHA-256 and, when the reference is
 *     mounted, by a text comparison with the header.
 * (2) lnsfaid_cfg_default(): the constants compiled into CDecoder_OMS.cpp / CDecoder_FAID.cpp /
 *     CDecoder_FAID_2B1C.cpp (file:line next to each value).
 */
#include "lnsfaid.h"
#include <string.h>

#define GPON_Z 256
#define GPON_BLOCK_ROWS 12
#define GPON_BLOCK_COLS 69
#define GPON_MAX_DEG 23

typedef struct { int16_t cb; int16_t shift; } gpon_circ;

static const int gpon_row_deg[GPON_BLOCK_ROWS] = { 23, 22, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23 };

static const gpon_circ gpon_base[GPON_BLOCK_ROWS][GPON_MAX_DEG] = {
    /* r0  */ { {0,80}, {3,60}, {4,169}, {6,11}, {8,143}, {11,222}, {13,59}, {15,218}, {18,178}, {24,105}, {27,19}, {30,126}, {34,211}, {40,247}, {42,255}, {45,85}, {52,246}, {53,94}, {59,242}, {64,129}, {66,19}, {67,58}, {68,27} },
    /* r1  */ { {1,0}, {3,0}, {5,0}, {7,0}, {9,0}, {11,0}, {13,0}, {15,0}, {17,0}, {18,0}, {20,0}, {24,0}, {27,0}, {32,0}, {36,0}, {39,0}, {43,0}, {47,0}, {52,0}, {56,0}, {60,0}, {67,0} },
    /* r2  */ { {1,91}, {3,74}, {5,237}, {6,202}, {9,201}, {10,136}, {12,178}, {14,239}, {16,183}, {19,217}, {21,232}, {25,169}, {32,129}, {33,60}, {39,19}, {40,76}, {46,77}, {50,2}, {54,101}, {57,217}, {61,48}, {67,172}, {68,42} },
    /* r3  */ { {0,105}, {3,87}, {5,43}, {7,165}, {9,180}, {11,80}, {12,227}, {14,221}, {16,77}, {19,0}, {24,16}, {29,252}, {31,96}, {33,0}, {38,17}, {44,219}, {47,198}, {48,165}, {53,36}, {58,171}, {63,228}, {67,39}, {68,234} },
    /* r4  */ { {1,170}, {2,250}, {5,195}, {6,139}, {9,135}, {11,92}, {13,147}, {15,1}, {20,13}, {23,98}, {26,142}, {30,225}, {36,23}, {37,108}, {44,0}, {46,0}, {51,135}, {56,121}, {57,0}, {63,0}, {66,46}, {67,242}, {68,228} },
    /* r5  */ { {1,46}, {3,37}, {5,49}, {6,150}, {8,65}, {11,177}, {12,144}, {14,70}, {16,95}, {19,221}, {23,192}, {25,128}, {28,214}, {34,51}, {38,100}, {41,19}, {44,235}, {52,4}, {55,251}, {58,109}, {64,140}, {67,193}, {68,241} },
    /* r6  */ { {0,137}, {2,104}, {4,238}, {7,228}, {9,225}, {10,247}, {13,191}, {15,177}, {17,255}, {22,192}, {27,51}, {32,195}, {34,0}, {37,172}, {43,219}, {46,236}, {49,136}, {53,0}, {57,159}, {60,10}, {65,5}, {67,25}, {68,94} },
    /* r7  */ { {1,118}, {2,15}, {4,93}, {7,228}, {9,78}, {11,16}, {12,0}, {14,48}, {16,0}, {20,62}, {22,0}, {25,0}, {30,0}, {36,112}, {38,0}, {45,0}, {49,0}, {50,0}, {55,22}, {61,0}, {62,0}, {67,120}, {68,192} },
    /* r8  */ { {1,208}, {2,0}, {4,0}, {6,0}, {8,0}, {10,0}, {13,251}, {14,0}, {17,44}, {18,123}, {23,0}, {26,0}, {31,0}, {35,0}, {40,0}, {43,153}, {48,0}, {51,0}, {55,0}, {65,0}, {66,0}, {67,16}, {68,0} },
    /* r9  */ { {0,0}, {3,123}, {5,41}, {6,191}, {8,211}, {10,217}, {12,243}, {14,97}, {16,252}, {21,0}, {28,0}, {31,41}, {35,29}, {37,0}, {42,0}, {47,193}, {49,145}, {54,0}, {61,140}, {62,46}, {65,58}, {67,202}, {68,215} },
    /* r10 */ { {0,209}, {2,252}, {4,39}, {7,159}, {8,69}, {10,37}, {12,134}, {15,201}, {16,49}, {21,104}, {26,129}, {29,157}, {33,222}, {41,139}, {42,39}, {48,203}, {50,94}, {56,194}, {59,3}, {62,43}, {63,153}, {67,207}, {68,109} },
    /* r11 */ { {0,53}, {2,93}, {4,216}, {7,57}, {8,9}, {10,130}, {13,130}, {15,238}, {22,144}, {28,162}, {29,0}, {35,175}, {39,145}, {41,0}, {45,36}, {51,91}, {54,22}, {58,0}, {59,0}, {60,212}, {64,0}, {67,69}, {68,88} },
};

int lnsfaid_code_50gpon(lnsfaid_code* code, uint16_t* pos_vn, int32_t* deg3, int32_t* deg_rows3)
{
    if (!code || !pos_vn || !deg3 || !deg_rows3) return LNSFAID_E_INVAL;
    size_t e = 0;
    for (int br = 0; br < GPON_BLOCK_ROWS; ++br)
        for (int i = 0; i < GPON_Z; ++i)
            for (int j = 0; j < gpon_row_deg[br]; ++j)
                pos_vn[e++] = (uint16_t)(gpon_base[br][j].cb * GPON_Z + ((gpon_base[br][j].shift + i) % GPON_Z));
    /* DEG_1 23 x256, DEG_2 22 x256, DEG_3 23 x2560 (Constants_SSE.h:14-19) */
    deg3[0] = 23; deg_rows3[0] = 256;
    deg3[1] = 22; deg_rows3[1] = 256;
    deg3[2] = 23; deg_rows3[2] = 2560;
    code->n_var = GPON_BLOCK_COLS * GPON_Z;   /* _NoVar   17664 */
    code->n_check = GPON_BLOCK_ROWS * GPON_Z; /* _NoCheck 3072  */
    code->n_edges = (int32_t)e;               /* _NoOnes  70400 */
    code->z = GPON_Z;
    code->puncture_tail = 384;                /* CDecoder_FAID.cpp:253-255 */
    code->nb_degres = 3;
    code->deg = deg3;
    code->deg_rows = deg_rows3;
    code->pos_vn = pos_vn;
    return LNSFAID_OK;
}

static void fill_map(int8_t dst[4][8], const int8_t row[8])
{
    for (int w = 0; w < 4; ++w) memcpy(dst[w], row, 8);
}

/* The reference's other compile-time table sets of Decode_FAID (#define FAID32 / FAID2 instead of FAID3,
 * CDecoder_FAID.cpp:8, :51-127); every weight class carries the same row in all of them. */
int lnsfaid_cfg_table_preset(lnsfaid_cfg* cfg, int32_t preset)
{
    static const int8_t faid3[6][8] = { /* CDecoder_FAID.cpp:13-48 */
        { 0, 1, 1, 2, 3, 3, 3, 3 }, { 0, 1, 1, 2, 3, 3, 3, 3 }, { 0, 1, 1, 2, 4, 4, 4, 4 },
        { 0, 1, 1, 3, 3, 4, 4, 4 }, { 0, 1, 1, 3, 3, 3, 6, 6 }, { 0, 1, 1, 3, 3, 3, 7, 7 },
    };
    static const int8_t faid32[6][8] = { /* CDecoder_FAID.cpp:52-87 */
        { 0, 1, 1, 2, 3, 3, 3, 3 }, { 0, 1, 1, 2, 3, 3, 3, 3 }, { 0, 1, 1, 2, 4, 4, 4, 4 },
        { 1, 1, 1, 1, 4, 4, 4, 4 }, { 1, 1, 1, 1, 5, 5, 5, 5 }, { 1, 1, 1, 1, 6, 6, 6, 6 },
    };
    static const int8_t faid2[6][8] = { /* CDecoder_FAID.cpp:91-126 */
        { 0, 0, 2, 2, 2, 2, 2, 2 }, { 0, 0, 2, 2, 2, 2, 2, 2 }, { 1, 1, 1, 3, 3, 3, 3, 3 },
        { 1, 1, 1, 4, 4, 4, 4, 4 }, { 1, 1, 1, 5, 5, 5, 5, 5 }, { 1, 1, 1, 6, 6, 6, 6, 6 },
    };
    const int8_t (*t)[8] = preset == LNSFAID_TABLES_FAID3 ? faid3 : preset == LNSFAID_TABLES_FAID32 ? faid32
                         : preset == LNSFAID_TABLES_FAID2 ? faid2 : 0;
    if (!cfg || !t) return LNSFAID_E_INVAL;
    for (int it = 0; it < 6; ++it) fill_map(cfg->v2c_map[it], t[it]);
    return LNSFAID_OK;
}

int lnsfaid_cfg_default(lnsfaid_cfg* cfg, int32_t decode_method, int32_t max_iteration)
{
    /* every weight class (3, 6, 11, other) carries the same row in the shipped tables */
    static const int8_t faid3[6][8] = {
        /* CDecoder_FAID.cpp:13-48 (#define FAID3, :8) */
        { 0, 1, 1, 2, 3, 3, 3, 3 }, { 0, 1, 1, 2, 3, 3, 3, 3 }, { 0, 1, 1, 2, 4, 4, 4, 4 },
        { 0, 1, 1, 3, 3, 4, 4, 4 }, { 0, 1, 1, 3, 3, 3, 6, 6 }, { 0, 1, 1, 3, 3, 3, 7, 7 },
    };
    static const int8_t faid_2b1c[6][8] = {
        /* CDecoder_FAID_2B1C.cpp:12-47 */
        { 0, 0, 1, 2, 3, 3, 3, 3 }, { 0, 1, 1, 2, 3, 3, 3, 3 }, { 0, 1, 1, 2, 3, 3, 3, 3 },
        { 0, 1, 1, 3, 3, 4, 4, 4 }, { 0, 1, 1, 3, 3, 3, 6, 6 }, { 0, 1, 1, 3, 3, 3, 7, 7 },
    };
    /* CDecoder_FAID.cpp:130-165, CDecoder_FAID_2B1C.cpp:49-84: identical for it1..it6 */
    static const int8_t ef[8] = { 2, 3, 3, 4, 5, 6, 6, 7 };
    static const int8_t ident[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };

    if (!cfg || max_iteration < 0) return LNSFAID_E_INVAL;
    memset(cfg, 0, sizeof(*cfg));
    cfg->decode_method = decode_method;
    cfg->max_iteration = max_iteration;
    cfg->factor_1 = 1; /* Profile.txt:11 */
    cfg->factor_2 = 6; /* Profile.txt:12 */
    cfg->bf_L1 = 0;    /* _L1    CDecoder_FAID.cpp:169 */
    cfg->bf_alpha = 1; /* _alpha CDecoder_FAID.cpp:170 */
    cfg->bf_delta = 1; /* _delta CDecoder_FAID.cpp:167 */
    cfg->regular_col_weight = 3; /* CTool.h:6 */
    cfg->hard2_threshold = 13;   /* CDecoder_FAID_2B1C.cpp:6130 */
    cfg->bf_vote_cap = 5;        /* CDecoder_OMSBF.cpp:3332 */
    for (int it = 0; it < 6; ++it) fill_map(cfg->v2c_map_ef[it], ef);
    switch (decode_method) {
    case 0: /* Decode: normalised min-sum, Factor_1 / Factor_2 are numerators over 32 (CLDPC.cpp:337-352) */
        cfg->max_bf_iter = 0;
        for (int it = 0; it < 6; ++it) fill_map(cfg->v2c_map[it], ident);
        break;
    case 1: /* Decode_OMS */
        cfg->floor_err_count = 100;  /* CDecoder_OMS.cpp:28 */
        cfg->floor_iter_thresh = 4;  /* CDecoder_OMS.cpp:29 */
        cfg->ef_elimination = 0;
        cfg->max_bf_iter = 0;        /* no bit flipping stage */
        cfg->bf_L0 = 0;
        for (int it = 0; it < 6; ++it) fill_map(cfg->v2c_map[it], ident); /* min(|t|,7), CDecoder_OMS.cpp:374 */
        break;
    case 3: /* Decode_OMSBF: the OMS layered loop followed by plain bit flipping */
        cfg->floor_err_count = 100;  /* CDecoder_OMSBF.cpp:28 */
        cfg->floor_iter_thresh = 4;  /* CDecoder_OMSBF.cpp:29 */
        cfg->ef_elimination = 0;
        cfg->max_bf_iter = 50;       /* CDecoder_OMSBF.cpp:30 */
        for (int it = 0; it < 6; ++it) fill_map(cfg->v2c_map[it], ident);
        break;
    case 4: /* Decode_OMS_DTBF: the OMS layered loop followed by the DTBF stage with its own constants */
        cfg->floor_err_count = 100;  /* CDecoder_OMS_DTBF.cpp:33 */
        cfg->floor_iter_thresh = 4;  /* CDecoder_OMS_DTBF.cpp:34 */
        cfg->ef_elimination = 0;
        cfg->max_bf_iter = 50;       /* CDecoder_OMS_DTBF.cpp:35 */
        cfg->bf_L0 = 0;              /* CDecoder_OMS_DTBF.cpp:7  */
        cfg->bf_L1 = 50;             /* CDecoder_OMS_DTBF.cpp:8  */
        for (int it = 0; it < 6; ++it) fill_map(cfg->v2c_map[it], ident);
        break;
    case 2: /* Decode_FAID */
        cfg->floor_err_count = 0;    /* CDecoder_FAID.cpp:193 */
        cfg->floor_iter_thresh = -1; /* CDecoder_FAID.cpp:194 */
        cfg->ef_elimination = 0;     /* CDecoder_FAID.cpp:6   */
        cfg->max_bf_iter = 10;       /* CDecoder_FAID.cpp:208 */
        cfg->bf_L0 = 50;             /* CDecoder_FAID.cpp:168 */
        for (int it = 0; it < 6; ++it) fill_map(cfg->v2c_map[it], faid3[it]);
        break;
    case 5: /* Decode_FAID_2B1C */
        cfg->floor_err_count = 50;   /* CDecoder_FAID_2B1C.cpp:117 */
        cfg->floor_iter_thresh = 6;  /* CDecoder_FAID_2B1C.cpp:118 */
        cfg->ef_elimination = 1;     /* CDecoder_FAID_2B1C.cpp:5   */
        cfg->max_bf_iter = 10;       /* CDecoder_FAID_2B1C.cpp:128 */
        cfg->bf_L0 = 100;            /* CDecoder_FAID_2B1C.cpp:88  */
        for (int it = 0; it < 6; ++it) fill_map(cfg->v2c_map[it], faid_2b1c[it]);
        break;
    default:
        return LNSFAID_E_INVAL;
    }
    return LNSFAID_OK;
}

/* EF_ELIMINATION of CDecoder_FAID.cpp (:6, :192-203): 0 off (the shipped build), 1 error-floor tables, 2 tables + erasure. */
int lnsfaid_cfg_ef_elimination(lnsfaid_cfg* cfg, int32_t mode)
{
    if (!cfg || cfg->decode_method != 2 || mode < 0 || mode > 2) return LNSFAID_E_INVAL;
    cfg->ef_elimination = mode;
    cfg->floor_err_count = mode == 0 ? 0 : (mode == 1 ? 100 : 20);
    cfg->floor_iter_thresh = mode == 0 ? -1 : 6;
    return LNSFAID_OK;
}
