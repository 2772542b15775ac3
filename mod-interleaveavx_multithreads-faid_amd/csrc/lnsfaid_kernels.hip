/*
 * lnsfaid_kernels.hip — CDNA4 (gfx950) kernels of the batched LDPC decode hot path.
 *
 * Mapping.  The reference interleaves 32 codewords in the int8 lanes of one AVX
 * register and walks the 3072 check rows serially (CDecoder_FAID.cpp:631-1527).  Here one 256-thread
 * workgroup owns ONE codeword and thread i owns check row i of every layer (block row) of the
 * quasi-cyclic H: the 256 rows of a layer touch disjoint variable nodes (no block row repeats a block
 * column), so a layer is one parallel step and the reference's row-serial schedule is reproduced
 * exactly by running the 12 layers in order with a barrier in between.
 *   - a-posteriori LLRs En (int8, 17664 B) live in LDS for the whole launch; a circulant with shift s
 *     makes lane i touch byte (s + i) mod 256 of its block column: conflict-free, wrap included;
 *   - check-to-variable messages are never stored per edge (the reference keeps 70400 bytes per
 *     codeword, CLDPC.h:123).  A row's outgoing messages are +-c1 on the edge holding the first
 *     minimum and +-c2 elsewhere, so 8 bytes per row {sign bits | argmin, c1, c2} reproduce every Lmn
 *     bit for bit (proof of the tie case in DESIGN.md §3.2); they stream through global memory one
 *     layer ahead of use (coalesced 8 B per lane);
 *   - per-row reductions (min1 / min2 / argmin / sign parity) are register-serial over the <= 24
 *     edges of the row; the per-codeword syndrome weight is a wave ballot + popcount.
 *
 * Group-of-32 semantics.  The reference stops a group of 32 codewords only when all 32 are clean
 * (CDecoder_FAID.cpp:616, :6782) and keeps iterating / flipping already-clean lanes meanwhile, which
 * changes their output.  Decision points are numbered on one time line per group:
 *   t = 1..max_iter           syndrome check in front of layered iteration t
 *   t = max_iter+1+b          syndrome check in front of bit-flipping iteration b (b < max_bf)
 *   t = max_iter+1+max_bf     loops exhausted
 * A codeword may pass point t once it knows the group does not stop there: it is dirty itself, or some
 * lane of its group is already parked beyond t.  So a workgroup runs ahead while its codeword is dirty
 * and parks (state to HBM, status = t) when it is clean; the host relaunches until no codeword is left.
 * Every launch reads one consistent snapshot of the previous launch's status words (double buffer), so
 * there is no inter-workgroup communication inside a launch and nothing to deadlock.
 */
#include <hip/hip_runtime.h>

#include "lnsfaid_device.h"

#define SAT_POS_VAR 31 /* Constants_SSE.h:22 */
#define SAT_NEG_VAR (-31)
#define SAT_POS_MSG 7  /* Constants_SSE.h:24 */

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int iclamp(int x, int lo, int hi) { return imin(imax(x, lo), hi); }

/* LDS byte address of the variable node that lane `tid` touches through circulant `ci` */
__device__ __forceinline__ int vn_addr(uint32_t ci, int tid)
{
    return (int)((ci & 0xffu) << 8) + (int)((((ci >> 8) & 0xffu) + (uint32_t)tid) & 0xffu);
}

/* sum of a wave-uniform per-wave value over the 4 waves of the workgroup */
__device__ __forceinline__ int block_sum4(int wave_value, int tid, int* sRed)
{
    if ((tid & 63) == 0) sRed[tid >> 6] = wave_value;
    __syncthreads();
    int total = sRed[0] + sRed[1] + sRed[2] + sRed[3];
    __syncthreads();
    return total;
}

/* ---- syndrome in front of a layered iteration (CDecoder_FAID.cpp:291-343, CDecoder_OMS.cpp:102-323) -----
 * returns the number of unsatisfied checks of the codeword; pbits bit br = parity of this lane's row in
 * layer br (l_checksum_[br*256 + tid]). */
__device__ int eval_main(const LfDevCode* __restrict__ c, const int8_t* sEn, int tid, uint32_t& pbits, int* sRed)
{
    uint32_t pb = 0;
    int cnt = 0;
    const int nbr = c->nbr;
    for (int br = 0; br < nbr; ++br) {
        const int deg = c->deg[br];
        int p = 0;
        for (int j = 0; j < deg; ++j) p ^= (sEn[vn_addr(c->circ[br][j], tid)] > 0) ? 1 : 0;
        pb |= (uint32_t)p << br;
        cnt += __popcll(__ballot(p));
    }
    pbits = pb;
    return block_sum4(cnt, tid, sRed);
}

/* selective offset of OMS_MODE 1 on one minimum (CDecoder_OMS.cpp:388-425); all operands are int8 in the
 * reference and small enough that its saturating adds never saturate */
__device__ __forceinline__ int oms_offset(int x, bool window, bool F, int f1, int f2)
{
    if (window && F) {
        if (x < f2) x += 1;
        if (x <= f1) x += 1;
    } else {
        if (x > f1) x -= 1;
        if (x >= f2) x -= 1;
    }
    return x;
}

/* ---- one layered iteration (FAID / 2B1C: CDecoder_FAID.cpp:631-1527; OMS: CDecoder_OMS.cpp:334-743) -----
 * rows: this codeword's compressed messages, [nbr][256] uint2:
 *   .x = bit j: sign of Lmn on edge j (1 = negative) | argmin edge << 24
 *   .y = c1 | c2 << 4      Lmn(edge j) = (j == argmin ? c1 : c2) with that sign */
template <int METHOD>
__device__ void main_step(const LfDevCode* __restrict__ c, const LfDevCfg* __restrict__ f, int8_t* sEn,
                          uint2* __restrict__ rows, int tid, int it, uint32_t pbits, bool lme)
{
    const bool fresh = (it == 1); /* no iteration has run yet: every Lmn is still 0, nothing in HBM */
    const int rem = f->max_iter - it; /* nombre_iterations inside the loop body */
    const int itx = (it >= 1 && it <= 5) ? it - 1 : 5; /* switch at CDecoder_FAID.cpp:760-779 */
    const bool window = rem <= f->floor_iter_thresh;
    const int f1 = f->factor_1, f2 = f->factor_2;
    const int nbr = c->nbr;

    uint2 cur = make_uint2(0u, 0u); /* Lmn = 0 before the first iteration (CDecoder_FAID.cpp:211-214) */
    if (!fresh) cur = rows[tid];
    for (int br = 0; br < nbr; ++br) {
        uint2 nxt = make_uint2(0u, 0u);
        if (!fresh && br + 1 < nbr) nxt = rows[(br + 1) * LF_Z + tid]; /* one layer ahead of use */
        const int deg = c->deg[br];
        const uint32_t negw = cur.x;
        const int idx_old = (int)(cur.x >> 24);
        const int c1o = (int)(cur.y & 15u), c2o = (int)((cur.y >> 4) & 15u);
        const bool pr = (pbits >> br) & 1u;
        const bool efsel = (METHOD == 5) && window && lme && pr; /* mask_eef, CDecoder_FAID.cpp:713-720 */

        uint32_t tw[LF_MAX_DEG / 4] = { 0u, 0u, 0u, 0u, 0u, 0u }; /* V2C values t of the row, one byte each */
        uint32_t sgn = 0;
        int min1 = SAT_POS_VAR, min2 = SAT_POS_VAR, jmin = 0;

#pragma unroll
        for (int j = 0; j < LF_MAX_DEG; ++j) {
            if (j < deg) {
                const uint32_t ci = c->circ[br][j];
                const int En = sEn[vn_addr(ci, tid)];
                const int mag = (j == idx_old) ? c1o : c2o;
                const int Lmn = ((negw >> j) & 1u) ? -mag : mag;
                int t = imax(En - Lmn, SAT_NEG_VAR); /* VECTOR_SUB_AND_SATURATE_VAR_8bits */
                int s, m;
                if (METHOD == 1) {
                    s = t < 0;                                        /* CDecoder_OMS.cpp:372 */
                    m = imin(t < 0 ? -t : t, SAT_POS_MSG);            /* :374 */
                } else {
                    t = imin(t, SAT_POS_VAR);                         /* CDecoder_FAID.cpp:672 */
                    s = ((t != 0) ? t : En) < 0;                      /* sign back-track, :682 */
                    const int a = imin(t < 0 ? -t : t, SAT_POS_MSG);  /* |t| >= 8 maps through column 7, :783 */
                    const uint32_t w = (ci >> 16) & 3u;
                    uint32_t lut = f->lut[itx][w];
                    if (METHOD == 5) { const uint32_t le = f->lut_ef[itx][w]; lut = efsel ? le : lut; }
                    m = (int)((lut >> (4 * a)) & 15u);
                }
                sgn |= (uint32_t)s << j;
                min2 = imin(min2, imax(min1, m)); /* VECTOR_MIN_2 with the old min1 */
                jmin = (m < min1) ? j : jmin;
                min1 = imin(min1, m);
                tw[j >> 2] |= ((uint32_t)t & 0xffu) << ((j & 3) * 8);
            }
        }

        int c1, c2;
        if (METHOD == 1) {
            const bool F = pr && lme;
            c1 = imin(oms_offset(min2, window, F, f1, f2), SAT_POS_MSG); /* cste_1, CDecoder_OMS.cpp:431 */
            c2 = imin(oms_offset(min1, window, F, f1, f2), SAT_POS_MSG); /* cste_2 */
        } else {
            c1 = imin(min2, SAT_POS_MSG); /* CDecoder_FAID.cpp:865-866, offset 0 */
            c2 = imin(min1, SAT_POS_MSG);
        }
        /* sign of the new message on edge j: XOR of all signs ^ (deg odd) ^ own sign
         * (the 0xC0 / 0x40 constants of CDecoder_FAID.cpp:902-906 fed to _mm256_sign_epi8) */
        const uint32_t flip = ((__popc(sgn) ^ deg) & 1) ? 0xffffffffu : 0u;
        const uint32_t negn = (sgn ^ flip) & ((1u << deg) - 1u);

#pragma unroll
        for (int j = 0; j < LF_MAX_DEG; ++j) {
            if (j < deg) {
                const uint32_t ci = c->circ[br][j];
                const int t = (int)(int8_t)(tw[j >> 2] >> ((j & 3) * 8));
                const int mag = (j == jmin) ? c1 : c2;
                const int Lmn = ((negn >> j) & 1u) ? -mag : mag;
                sEn[vn_addr(ci, tid)] = (int8_t)iclamp(t + Lmn, SAT_NEG_VAR, SAT_POS_VAR); /* :919-920 */
            }
        }
        rows[br * LF_Z + tid] = make_uint2(negn | ((uint32_t)jmin << 24), (uint32_t)c1 | ((uint32_t)c2 << 4));
        __syncthreads(); /* the next layer reads what this one wrote */
        cur = nxt;
    }
}

/* ---- bit-flipping stage: layout of the LDS overlay -------------------------------------------------- */
/* [0, nw) hard_llr   [nw, 2nw) hard_ch   [2nw, 3nw) hard2_llr   [3nw, 3nw + pw) l_checksum_ bits */

/* hard decisions from En, straight to the codeword's HBM bit planes (CDecoder_FAID.cpp:6416-6419,
 * CDecoder_FAID_2B1C.cpp:6132-6136) */
__device__ void bf_init_planes(const LfDevCode* __restrict__ c, const LfDevCfg* __restrict__ f, const int8_t* sEn,
                               uint32_t* __restrict__ gbits, int tid)
{
    const int nw = c->n_words;
    const int thr = f->hard2_thr;
    for (int base = 0; base < c->n_var; base += LF_Z) {
        const int v = base + tid;
        int e = (v < c->n_var) ? (int)sEn[v] : 0;
        const unsigned long long h = __ballot(e > 0);
        const unsigned long long h2 = __ballot(e >= thr || e <= -thr);
        if ((tid & 63) == 0 && v < c->n_var) {
            const int w = v >> 5;
            gbits[w] = (uint32_t)h;
            gbits[nw + w] = (uint32_t)h;
            gbits[2 * nw + w] = (uint32_t)h2;
            if (w + 1 < nw) {
                gbits[w + 1] = (uint32_t)(h >> 32);
                gbits[nw + w + 1] = (uint32_t)(h >> 32);
                gbits[2 * nw + w + 1] = (uint32_t)(h2 >> 32);
            }
        }
    }
}

__device__ __forceinline__ int bit_of(const uint32_t* words, int v) { return (int)((words[v >> 5] >> (v & 31)) & 1u); }

/* syndrome on the hard decisions (CDecoder_FAID.cpp:6443-6491): parity bits into the LDS plane, returns
 * the number of unsatisfied checks */
__device__ int eval_bf(const LfDevCode* __restrict__ c, uint32_t* sBits, int tid, int* sRed)
{
    const int nw = c->n_words;
    uint32_t* sP = sBits + 3 * nw;
    int cnt = 0;
    const int nbr = c->nbr;
    for (int br = 0; br < nbr; ++br) {
        const int deg = c->deg[br];
        int p = 0;
        for (int j = 0; j < deg; ++j) p ^= bit_of(sBits, vn_addr(c->circ[br][j], tid));
        const unsigned long long b = __ballot(p);
        cnt += __popcll(b);
        if ((tid & 63) == 0) {
            const int w = (br * LF_Z + tid) >> 5;
            sP[w] = (uint32_t)b;
            sP[w + 1] = (uint32_t)(b >> 32);
        }
    }
    return block_sum4(cnt, tid, sRed); /* its barriers also publish sP */
}

/* one bit-flipping iteration after a dirty syndrome (CDecoder_FAID.cpp:6787-6845, :7084-7086;
 * CDecoder_FAID_2B1C.cpp:6801-6814) */
template <int METHOD>
__device__ void bf_step(const LfDevCode* __restrict__ c, const LfDevCfg* __restrict__ f, uint32_t* sBits, int tid,
                        LfLaneState& ls, int* sRed)
{
    const int nw = c->n_words;
    uint32_t* sHard = sBits;
    const uint32_t* sHard0 = sBits + nw;
    uint32_t* sHard2 = sBits + 2 * nw;
    const uint32_t* sP = sBits + 3 * nw;
    const int W = f->W;

    /* threshold state machine on int8 lanes (CDecoder_FAID.cpp:6787-6799) */
    int Th = ls.Th, l0 = ls.l0, l1 = ls.l1;
    if (!ls.t) Th = imax(Th - f->delta, -128);
    const bool max_Th = ls.t && (l0 < (int)(int8_t)f->L0);
    if (max_Th) { Th = (int8_t)(W + f->alpha); l0 = imin(l0 + 1, 127); }
    const bool submax_Th = ls.t && !max_Th && (l1 < (int)(int8_t)f->L1);
    if (submax_Th) { Th = (int8_t)(W + f->alpha - f->delta); l1 = imin(l1 + 1, 127); }
    if (ls.t && !max_Th && !submax_Th) Th = (int8_t)(W + f->alpha - 2 * f->delta);
    Th = imax(Th, 1);
    const bool big = Th >= (int)(int8_t)W; /* mask_big_jump, 2B1C only */
    const int alpha = (int8_t)f->alpha;

    int any = 0;
    const int nbc = c->nbc;
    for (int cb = 0; cb < nbc; ++cb) {
        if (c->col_weight[cb] != W) continue; /* VN_weight_[v] == REGULAR_COL_WEIGHT, :6806 */
        const int v = cb * LF_Z + tid;
        int vote = 0;
        for (int k = 0; k < W; ++k) {
            const uint32_t cc = c->colcirc[cb][k];
            const int r = (int)((cc & 0xffu) << 8) + (int)(((uint32_t)tid - ((cc >> 8) & 0xffu)) & 0xffu);
            vote += bit_of(sP, r);
        }
        const int flipped = bit_of(sHard, v) ^ bit_of(sHard0, v);
        const int fl = (imin(vote + (flipped ? alpha : 0), 127) >= Th) ? 1 : 0;
        const unsigned long long fm = __ballot(fl);
        any |= (fm != 0ull);
        if ((tid & 63) == 0) {
            const int w = v >> 5;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t m = (uint32_t)(fm >> (32 * h));
                if (METHOD == 5) {
                    const uint32_t h2 = sHard2[w + h];
                    if (big) { sHard[w + h] ^= m; sHard2[w + h] = h2 ^ m; }
                    else { sHard[w + h] ^= m & ~h2; sHard2[w + h] = h2 & ~m; }
                } else {
                    sHard[w + h] ^= m;
                }
            }
        }
    }
    ls.Th = Th; ls.l0 = l0; ls.l1 = l1;
    ls.t = block_sum4(any, tid, sRed) != 0; /* barriers also order the plane updates before the next syndrome */
}

/* ---- the decode kernel: one workgroup per codeword ---------------------------------------------------- */
template <int METHOD>
__global__ __launch_bounds__(LF_Z) void lnsfaid_decode_kernel(LfKernelArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const LfDevCode* __restrict__ c = a.code;
    const LfDevCfg* __restrict__ f = a.cfg;
    const int tid = (int)threadIdx.x;
    const int cw = (int)blockIdx.x;
    const int N = c->n_var, M = c->n_check, K = c->k_info, nw = c->n_words;
    const int lds_main = (N + 15) & ~15;
    const int lds_bf = ((3 * nw + c->p_words + 2) * 4 + 15) & ~15;
    const int lds_state = lds_main > lds_bf ? lds_main : lds_bf;
    int8_t* sEn = (int8_t*)smem;
    uint32_t* sBits = (uint32_t*)smem;
    int* sStat = (int*)(smem + lds_state);
    int* sRed = sStat + LNSFAID_GROUP;

    const int max_iter = f->max_iter, max_bf = f->max_bf;
    const int t_bf0 = max_iter + 1;      /* first bit-flipping decision point */
    const int t_end = t_bf0 + max_bf;    /* both loops exhausted               */

    const int my_status = a.status_cur[cw];
    if (my_status & LF_DONE) { /* uniform exit */
        if (tid == 0) a.status_next[cw] = my_status;
        return;
    }
    /* snapshot of the 32 lanes of this group */
    const int g = cw >> 5, lane = cw & 31;
    if (tid < LNSFAID_GROUP) sStat[tid] = a.status_cur[g * LNSFAID_GROUP + tid];
    __syncthreads();
    int kmax = 0, all_same = 1;
    for (int l = 0; l < LNSFAID_GROUP; ++l) {
        const int s = sStat[l];
        kmax = imax(kmax, s & LF_PROG_MASK);
        all_same &= (s == my_status);
    }
    int prog = my_status & LF_PROG_MASK;

    int8_t* g_en = a.st_en + (size_t)cw * (size_t)N;
    uint2* g_rows = a.st_rows + (size_t)cw * (size_t)(c->nbr * LF_Z);
    uint32_t* g_bits = a.st_bits + (size_t)cw * (size_t)(3 * nw);
    int8_t* g_out = a.decoded + (size_t)cw * (size_t)N;

    /* parked on the group's front, not everybody there yet: nothing to do in this launch */
    if (prog != 0 && prog == kmax && !all_same) {
        if (tid == 0) { a.status_next[cw] = my_status; atomicAdd(a.remaining, 1u); }
        return;
    }

    bool in_bf = max_bf > 0 && prog >= t_bf0 && prog != 0;
    LfLaneState ls = { 0, 0, 0, 0 };

    /* ---- bring the codeword's state on chip ---- */
    if (prog == 0) {
        /* input staging (CDecoder_FAID.cpp:217-255): lane l of group g is information row l of the
         * [32][K] block followed by parity row l of the [32][M] block; erase the punctured tail */
        const int8_t* gi = a.fix_input + (size_t)g * (size_t)LNSFAID_GROUP * (size_t)N;
        const uint32_t* src_i = (const uint32_t*)(gi + (size_t)lane * (size_t)K);
        const uint32_t* src_p = (const uint32_t*)(gi + (size_t)LNSFAID_GROUP * (size_t)K + (size_t)lane * (size_t)M);
        uint32_t* dst = (uint32_t*)sEn;
        for (int i = tid; i < (K >> 2); i += LF_Z) dst[i] = src_i[i];
        for (int i = tid; i < (M >> 2); i += LF_Z) dst[(K >> 2) + i] = src_p[i];
        __syncthreads();
        for (int i = tid; i < c->puncture_tail; i += LF_Z) sEn[N - 1 - i] = 0;
        __syncthreads();
        prog = 1;
        in_bf = max_bf > 0 && prog >= t_bf0;
        if (in_bf) { /* max_iter == 0: straight to the bit-flipping stage */
            bf_init_planes(c, f, sEn, g_bits, tid);
            __threadfence_block();
            __syncthreads();
            for (int i = tid; i < 3 * nw; i += LF_Z) sBits[i] = g_bits[i];
            ls.Th = (int8_t)f->W; ls.l0 = 0; ls.l1 = 0; ls.t = 1;
            __syncthreads();
        }
    } else if (!in_bf) {
        const uint32_t* src = (const uint32_t*)g_en;
        uint32_t* dst = (uint32_t*)sEn;
        for (int i = tid; i < (N >> 2); i += LF_Z) dst[i] = src[i];
        __syncthreads();
    } else {
        for (int i = tid; i < 3 * nw; i += LF_Z) sBits[i] = g_bits[i];
        ls = a.st_lane[cw];
        __syncthreads();
    }

    /* ---- all 32 lanes parked clean at the same decision point: the group stops here ---- */
    const bool group_stop = (my_status != 0) && all_same;

    if (!group_stop) {
        for (;;) {
            if (prog >= t_end) break; /* loops exhausted (also OMS after max_iter iterations) */
            if (!in_bf) {
                uint32_t pbits;
                const int unsat = eval_main(c, sEn, tid, pbits, sRed);
                if (unsat == 0 && prog >= kmax) break; /* clean on the group's front: park */
                bool lme;
                if (METHOD == 1) lme = imin(unsat, 255) < (int)(uint8_t)f->floor_err_count; /* CDecoder_OMS.cpp:328 */
                else lme = imin(unsat, 127) < (int)(int8_t)f->floor_err_count;              /* CDecoder_FAID.cpp:619 */
                main_step<METHOD>(c, f, sEn, g_rows, tid, prog, pbits, lme);
                prog++;
                if (prog == t_bf0 && max_bf > 0) {
                    /* the layered loop ran out: enter the bit-flipping stage */
                    bf_init_planes(c, f, sEn, g_bits, tid);
                    __threadfence_block();
                    __syncthreads();
                    for (int i = tid; i < 3 * nw; i += LF_Z) sBits[i] = g_bits[i];
                    ls.Th = (int8_t)f->W; ls.l0 = 0; ls.l1 = 0; ls.t = 1;
                    in_bf = true;
                    __syncthreads();
                }
            } else {
                const int unsat = eval_bf(c, sBits, tid, sRed);
                if (unsat == 0 && prog >= kmax) break;
                bf_step<METHOD>(c, f, sBits, tid, ls, sRed);
                prog++;
            }
        }
    }

    const bool finished = group_stop || prog >= t_end;
    if (finished) {
        /* decodedBits[l][v] = hard decision (CDecoder_FAID.cpp:7091-7102, CDecoder_OMS.cpp:2966-2967) */
        uint32_t* out32 = (uint32_t*)g_out;
        if (!in_bf) {
            for (int i = tid; i < (N >> 2); i += LF_Z) {
                const uint32_t e4 = ((const uint32_t*)sEn)[i];
                uint32_t o = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) o |= (uint32_t)(((int8_t)(e4 >> (8 * b))) > 0) << (8 * b);
                out32[i] = o;
            }
        } else {
            for (int i = tid; i < (N >> 2); i += LF_Z) {
                const uint32_t bits = (sBits[i >> 3] >> ((i & 7) * 4)) & 15u;
                out32[i] = (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21);
            }
        }
        if (tid == 0) {
            a.status_next[cw] = prog | LF_DONE;
            if (a.stats && lane == 0) {
                lnsfaid_group_stats st;
                st.iterations = prog <= max_iter ? prog - 1 : max_iter;
                st.bf_iterations = prog <= max_iter ? 0 : prog - t_bf0;
                a.stats[g] = st;
            }
        }
    } else {
        /* park: state back to HBM, status = the decision point the codeword is clean at */
        if (!in_bf) {
            const uint32_t* src = (const uint32_t*)sEn;
            uint32_t* dst = (uint32_t*)g_en;
            for (int i = tid; i < (N >> 2); i += LF_Z) dst[i] = src[i];
        } else {
            for (int i = tid; i < 3 * nw; i += LF_Z) g_bits[i] = sBits[i];
            if (tid == 0) a.st_lane[cw] = ls;
        }
        if (tid == 0) { a.status_next[cw] = prog; atomicAdd(a.remaining, 1u); }
    }
}

/* ---- CalculateErrors (CLDPC.cpp:4842-4876): one workgroup per frame ----------------------------------- */
__global__ __launch_bounds__(LF_Z) void lnsfaid_count_errors_kernel(const int8_t* __restrict__ decoded,
                                                                    const int8_t* __restrict__ input_bits, int n_var,
                                                                    int k_info, unsigned long long* __restrict__ out)
{
    __shared__ int sRed[4];
    const int tid = (int)threadIdx.x;
    const size_t cw = blockIdx.x;
    const int8_t* d = decoded + cw * (size_t)n_var;
    const int8_t* r = input_bits ? input_bits + cw * (size_t)k_info : nullptr;
    int cnt = 0;
    for (int j = tid; j < k_info; j += LF_Z) cnt += (d[j] != (r ? r[j] : (int8_t)0)) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((tid & 63) == 0) sRed[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        const int errorBits = sRed[0] + sRed[1] + sRed[2] + sRed[3];
        atomicAdd(&out[0], 1ull);
        if (errorBits > 0) {
            atomicAdd(&out[1], 1ull);
            atomicAdd(&out[2], (unsigned long long)errorBits);
            if (errorBits < 3) atomicAdd(&out[3], 1ull);
        }
    }
}

/* ---- launchers (called from lnsfaid_capi.hip) ---------------------------------------------------------- */
extern "C" hipError_t lf_launch_decode(int method, const LfKernelArgs* args, size_t lds_bytes, hipStream_t stream)
{
    const dim3 grid((unsigned)args->n_cw), block(LF_Z);
    switch (method) {
    case 1: hipLaunchKernelGGL(lnsfaid_decode_kernel<1>, grid, block, lds_bytes, stream, *args); break;
    case 2: hipLaunchKernelGGL(lnsfaid_decode_kernel<2>, grid, block, lds_bytes, stream, *args); break;
    case 5: hipLaunchKernelGGL(lnsfaid_decode_kernel<5>, grid, block, lds_bytes, stream, *args); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" hipError_t lf_launch_count_errors(const int8_t* decoded, const int8_t* input_bits, int n_var, int k_info,
                                             size_t n_cw, unsigned long long* out, hipStream_t stream)
{
    hipLaunchKernelGGL(lnsfaid_count_errors_kernel, dim3((unsigned)n_cw), dim3(LF_Z), 0, stream, decoded, input_bits,
                       n_var, k_info, out);
    return hipGetLastError();
}
