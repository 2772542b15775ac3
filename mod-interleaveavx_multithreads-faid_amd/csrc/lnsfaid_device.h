/*
 * lnsfaid_device.h — device-side tables and launch arguments shared by the kernels (lnsfaid_kernels.hip)
 * and the host side of the C ABI (lnsfaid_capi.hip).  Internal: not part of the public boundary.
 */
#ifndef LNSFAID_DEVICE_H
#define LNSFAID_DEVICE_H

#include <stdint.h>

#include "lnsfaid.h"

#define LF_Z 256        /* circulant size                                                                  */
#define LF_T 128        /* threads per workgroup: thread i owns check rows i and i + 128 of every layer     */
#define LF_MAX_BR 32    /* block rows (layers); 50G-PON: 12   */
#define LF_MAX_DEG 24   /* check degree;        50G-PON: 23   */
#define LF_MAX_BC 256   /* block columns;       50G-PON: 69   */
#define LF_MAX_COLW 16  /* column weight;       50G-PON: 12   */

#define LF_DONE 0x40000000 /* status flag: codeword finished; low bits keep its last decision point */
#define LF_PROG_MASK 0x0fffffff

/* Edge j of a row pair inside the 16-bit min-search key: the bit index of its sign in the 64-bit pair
 * {.y, .x} of the compressed messages (row A: low halves, row B: high halves); also its lane in sbtab. */
#define LF_JCODE_A(j) ((j) < 16 ? (j) : (j) + 16)
#define LF_JCODE_B(j) ((j) < 16 ? (j) + 16 : (j) + 32)

struct LfCirc {
    uint32_t sb;     /* block column * 256 + circulant shift: LDS byte offset of row 0's variable node */
    uint32_t wclass; /* weight class 0..3 of the block column (V2C_map row)                            */
};

/* Quasi-cyclic view of the reference's PosNoeudsVariable table.  Uniformly indexed fields are read with
 * scalar loads (constant address space), lane-indexed ones (sbtab, synw, colcirc, wcol) with vector loads. */
struct LfDevCode {
    int32_t n_var, n_check, k_info, nbr, nbc, puncture_tail, n_words /* n_var / 32 */, p_words /* n_check / 32 */;
    int32_t n_wcols;                          /* block columns whose weight equals REGULAR_COL_WEIGHT           */
    int32_t deg[LF_MAX_BR];
    LfCirc circ[LF_MAX_BR][LF_MAX_DEG];
    uint32_t sbtab[LF_MAX_BR][64];            /* lane LF_JCODE_A(j) and lane LF_JCODE_B(j): circ[br][j].sb              */
    uint32_t sbplain[LF_MAX_BR][32];          /* lane j: (block column * 256) << 16 | 4 * shift of edge j (four-rows kernel)  */
    uint32_t s4tab[LF_MAX_BR][LF_MAX_DEG];    /* 4 * shift of edge j: what lane i adds to 4 i to get its dword (scalar loads)    */
    uint32_t cbtab[LF_MAX_BR][LF_MAX_DEG];    /* block column * 256 of edge j: LDS base of the column                            */
    uint2 synw[LF_MAX_BR][LF_MAX_DEG][8];     /* syndrome walk, per (layer, circulant, 32-row word k): .x = LDS byte addresses of
                                               * the two hard-plane words holding bits (32k + shift) mod 256 ... + 31 of the block
                                               * column (low / high 16 bits), .y = bit offset; unused slots: the zero word, 0     */
    int32_t col_weight[LF_MAX_BC];
    int32_t wcol[LF_MAX_BC];                  /* the n_wcols block columns of weight W                           */
    uint32_t era_edges[LF_MAX_BR];            /* bit j: edge j of the layer is the FIRST edge, in row order, of a block column of
                                               * weight W (EF_ELIMINATION 2 erases a variable node's V2C once per iteration)      */
    uint32_t colcirc[LF_MAX_BC][LF_MAX_COLW]; /* block row | shift << 8 for every circulant of the column         */
};

struct LfDevCfg {
    int32_t method, max_iter, factor_1, factor_2, floor_err_count, floor_iter_thresh, ef, max_bf;
    int32_t L0, L1, alpha, delta, W, hard2_thr, vote_cap;
    int32_t uniform_w; /* all four weight classes carry the same table rows and every row is non-decreasing (true for
                        * every shipped set): the table is then applied to the two minima instead of to every edge   */
    int32_t bf_fast;   /* W == 3 and alpha in {0, 1}: bit-sliced flip decision                                */
    int32_t nms_fits;  /* DecodeMethod 0: one factor whose cste() has 16 levels at most (lnsfaid_swar.h sw_nms_fits): runs on the
                        * four-rows-per-lane kernel with the table below                                                */
    uint32_t nms_t[4]; /* thermometer codes of cste(0..15), 16 bytes                                                   */
    /* V2C_map_it{1..6}_[class] as 8 bytes for v_perm_b32: lo = entries 0..3, hi = entries 4..7 */
    uint32_t lut_lo[6][4], lut_hi[6][4];
    uint32_t lut_ef_lo[6][4], lut_ef_hi[6][4];
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
    int8_t* st_en;                /* [n_cw][n_var]   a-posteriori LLRs En of parked codewords              */
    uint4* st_rows;               /* [n_cw][nbr][128] compressed check-to-variable messages of a row pair  */
    uint32_t* st_bits;            /* [n_cw][3][n_words] hard / hard_ch / hard2 bit planes (BF stage)       */
    LfLaneState* st_lane;         /* [n_cw]                                                               */
    const int32_t* status_cur;    /* [n_cw] decision point each codeword is parked at (snapshot); null = all 0 (first launch) */
    int32_t* status_next;         /* [n_cw] written by this launch                                        */
    uint32_t* remaining;          /* += number of codewords not finished after this launch (a running counter: never reset) */
    int32_t* live;                /* [n_cw] highest decision point each codeword has PASSED in this decode call, published
                                   * while the launch runs (four-rows-per-lane kernel); proof for group mates that the
                                   * group does not stop there */
    lnsfaid_group_stats* stats;   /* [n_groups] or null                                                   */
    int32_t n_cw;
};

/* dynamic LDS carve-up, shared by host (size) and device (offsets):
 *   [0, N)        En (int8).  In the bit-flipping stage En is dead and the same bytes hold
 *                 hard_ch  at [0, 4*n_words) and hard2 at [4*n_words, 8*n_words)
 *   off_hard      hard-decision bit plane, n_words words
 *   off_p         parity plane l_checksum_, p_words (+2) words
 *   off_stat      32 status words of the group + reduction scratch
 * 50G-PON: 17664 + 2208 + 392 + 160 = 20424 B <= 20480 B, i.e. 8 workgroups per CU. */
static inline __host__ __device__ uint32_t lf_lds_off_hard(int n_var) { return ((uint32_t)n_var + 15u) & ~15u; }
static inline __host__ __device__ uint32_t lf_lds_off_p(int n_var, int n_words) { return lf_lds_off_hard(n_var) + (uint32_t)n_words * 4u; }
static inline __host__ __device__ uint32_t lf_lds_off_stat(int n_var, int n_words, int p_words)
{
    return (lf_lds_off_p(n_var, n_words) + ((uint32_t)p_words + 2u) * 4u + 7u) & ~7u;
}
#define LF_ZERO_SLOT 7 /* reduction-scratch word that is zeroed at kernel entry and never written again */
static inline __host__ __device__ uint32_t lf_lds_off_zero(int n_var, int n_words, int p_words)
{
    return lf_lds_off_stat(n_var, n_words, p_words) + (LNSFAID_GROUP + LF_ZERO_SLOT) * 4u;
}
static inline __host__ __device__ uint32_t lf_lds_bytes(int n_var, int n_words, int p_words)
{
    return lf_lds_off_stat(n_var, n_words, p_words) + (LNSFAID_GROUP + 8) * 4u;
}

#endif
