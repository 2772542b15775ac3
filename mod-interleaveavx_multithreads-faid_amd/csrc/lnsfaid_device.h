/*
 * lnsfaid_device.h — device-side tables and launch arguments shared by the kernels (lnsfaid_kernels.hip)
 * and the host side of the C ABI (lnsfaid_capi.hip).  Internal: not part of the public boundary.
 */
#ifndef LNSFAID_DEVICE_H
#define LNSFAID_DEVICE_H

#include <stdint.h>

#include "lnsfaid.h"

#define LF_Z 256        /* circulant size = threads per workgroup: thread i owns check row i of every layer */
#define LF_MAX_BR 32    /* block rows (layers); 50G-PON: 12   */
#define LF_MAX_DEG 24   /* check degree;        50G-PON: 23   */
#define LF_MAX_BC 256   /* block columns;       50G-PON: 69   */
#define LF_MAX_COLW 16  /* column weight;       50G-PON: 12   */

#define LF_DONE 0x40000000 /* status flag: codeword finished; low bits keep its last decision point */
#define LF_PROG_MASK 0x0fffffff

/* Quasi-cyclic view of the reference's PosNoeudsVariable table, read with scalar loads. */
struct LfDevCode {
    int32_t n_var, n_check, k_info, nbr, nbc, puncture_tail, n_words /* n_var / 32 */, p_words /* n_check / 32 */;
    int32_t deg[LF_MAX_BR];
    uint32_t circ[LF_MAX_BR][LF_MAX_DEG];     /* block column | shift << 8 | weight class << 16 */
    int32_t col_weight[LF_MAX_BC];
    uint32_t colcirc[LF_MAX_BC][LF_MAX_COLW]; /* block row | shift << 8 for every circulant of the column */
};

struct LfDevCfg {
    int32_t method, max_iter, factor_1, factor_2, floor_err_count, floor_iter_thresh, ef, max_bf;
    int32_t L0, L1, alpha, delta, W, hard2_thr;
    uint32_t lut[6][4];    /* V2C_map_it{1..6}_[class], 8 nibbles: entry a in bits 4a..4a+3 */
    uint32_t lut_ef[6][4]; /* V2C_map_it{1..6}_ef                                            */
};

/* Per-codeword scalars of the bit-flipping stage that survive a pause. */
struct LfLaneState {
    int32_t Th, l0, l1, t;
};

struct LfKernelArgs {
    const LfDevCode* code;
    const LfDevCfg* cfg;
    const int8_t* fix_input;      /* reference fixInput layout, per group [32][K] then [32][M]           */
    int8_t* decoded;              /* reference decodedBits layout, per group [32][N]                      */
    int8_t* st_en;                /* [n_cw][n_var]   a-posteriori LLRs En of paused codewords              */
    uint2* st_rows;               /* [n_cw][nbr][256] compressed check-to-variable messages, see kernels   */
    uint32_t* st_bits;            /* [n_cw][3][n_words] hard / hard_ch / hard2 bit planes (BF stage)       */
    LfLaneState* st_lane;         /* [n_cw]                                                               */
    const int32_t* status_cur;    /* [n_cw] decision point each codeword is parked at (snapshot)          */
    int32_t* status_next;         /* [n_cw] written by this launch                                        */
    uint32_t* remaining;          /* number of codewords not finished after this launch                   */
    lnsfaid_group_stats* stats;   /* [n_groups] or null                                                   */
    int32_t n_cw;
};

#endif
