/*
 * lnsfaid_kernel4.hip — the decode kernel with ONE wavefront per codeword and four check rows per lane (gfx950).
 *
 * Same decoder, same group-of-32 protocol, same HBM state as lnsfaid_kernels.hip (see its header for the decision-point time
 * line and the park / relaunch rules); what differs is how a layer is computed:
 *   - lane i owns rows i, i + 64, i + 128, i + 192 of every layer; byte k of a working register belongs to row i + 64 k and
 *     the layer step is carry-free byte-parallel arithmetic on plain 32-bit operations (lnsfaid_swar.h) instead of two rows on
 *     packed 16-bit operations: 928 VALU instructions per layer of 256 rows, 70 % of them of the full-rate class, against
 *     2 x 615 with three quarters of the half-rate class;
 *   - En is kept in LDS interleaved (variable node v of a block column in dword v mod 64, byte v div 64, biased by 120), so an
 *     edge is one ds_read_b32 + one byte rotation for four rows, and a workgroup is a single wave: no barrier between layers;
 *   - the compressed messages of a lane's four rows are 24 bytes per layer (SwRow).
 * One wave per codeword means two waves per SIMD (the LDS image of a codeword allows 8 per CU), and with two waves nothing hides a
 * stall: every loop that loads keeps all its loads in flight before the first use, the walk tables of the bit-flipping stage live in
 * registers, and everything on the hot path is inlined (tests/test_kernel_isa.py holds these properties).
 * Used for DecodeMethods 1..5 whenever the FAID tables are uniform over the weight classes and non-decreasing (every shipped
 * set) and for DecodeMethod 0 with one normalisation factor >= 15; other tables / factors run on the two-rows-per-lane kernel.
 */
#include <hip/hip_runtime.h>

#include "lnsfaid_rows4.h"

/* ---- the compressed messages of the codeword on chip (RM instances) ------------------------------------------------
 * A lane's four rows are 6 dwords per layer (SwRow), 72 per codeword for the 12 layers of the 50G-PON code: they stay in
 * registers for the whole launch, field f of layer br in element br of vector f.  The layer number is wave-uniform, so an
 * access is one v_mov_b32 under s_set_gpr_idx_on (no scratch, no waterfall).  Codes with more than LF4_RM_LAYERS layers
 * stream the messages through HBM one layer ahead of use (the !RM instances). */
#define LF4_RM_LAYERS 12
static_assert(LF4_RM_LAYERS * 16 <= LF_SYN_ROUNDS * 64, "the RM instances assume that the syndrome walk tables fit the register cache");
typedef uint32_t lf4_vec __attribute__((ext_vector_type(LF4_RM_LAYERS)));
struct SwRegs {
    lf4_vec x0, x1, x2, cw, pa0, pa1;
};
__device__ __forceinline__ SwRow regs_get(const SwRegs& R, int br)
{
    SwRow r;
    r.x[0] = R.x0[br]; r.x[1] = R.x1[br]; r.x[2] = R.x2[br]; r.cw = R.cw[br]; r.pa[0] = R.pa0[br]; r.pa[1] = R.pa1[br];
    return r;
}
__device__ __forceinline__ void regs_put(SwRegs& R, int br, const SwRow& r)
{
    R.x0[br] = r.x[0]; R.x1[br] = r.x[1]; R.x2[br] = r.x[2]; R.cw[br] = r.cw; R.pa0[br] = r.pa[0]; R.pa1[br] = r.pa[1];
}

/* ---- one layered iteration (lnsfaid_swar.h does the rows) ---- */
template <int METHOD, bool ERA, bool RM>
__device__ __forceinline__ void main_step4(CCode c, CCfg f, const LfDevCode* gc, SwRow* __restrict__ rows, SwRegs& R, int lane, int it, const uint32_t* sP,
                           bool have_par, bool lme, uint32_t era_plane)
{
    /* register constants of the layer step: built per iteration (17 moves), outside the layer loop and the per-degree instances,
     * and dead again before the syndrome stage - kept alive across it they are spilled (they come from asm statements, which
     * the compiler cannot rematerialise) */
    it = __builtin_amdgcn_readfirstlane(it); /* uniform, and the compiler must know it: a divergent iteration number turns the
                                              * scalar branches and table loads of every layer into masked / per-lane ones */
    const SwK K = sw_consts((uint32_t)it);
    const bool fresh = (it == 1); /* no iteration has run yet: every Lmn is still 0, nothing in HBM */
    const int rem = f->max_iter - it;
    const int itx = (it >= 1 && it <= 5) ? it - 1 : 5; /* switch at CDecoder_FAID.cpp:760-779 */
    SwParams p;
    p.lut_lo = f->lut_lo[itx][0]; p.lut_hi = f->lut_hi[itx][0];
    p.ef_lo = f->lut_ef_lo[itx][0]; p.ef_hi = f->lut_ef_hi[itx][0];
    p.f1 = f->factor_1; p.f2 = f->factor_2;
    p.window = rem <= f->floor_iter_thresh;
    p.ef_tables = f->ef >= 1;
    if (LF4_OMS(METHOD)) sw_oms_tables(p); /* uniform: scalar work, once per iteration */
    if (METHOD == 0) { p.nms_t[0] = f->nms_t[0]; p.nms_t[1] = f->nms_t[1]; p.nms_t[2] = f->nms_t[2]; p.nms_t[3] = f->nms_t[3]; }
    const int nbr = c->nbr;
    const SwLds lds = SwLds();
    const SwRow zero = { { 0u, 0u, 0u }, 0u, { 0u, 0u } }; /* Lmn = 0 before the first iteration (CDecoder_FAID.cpp:211-214) */
    if (RM) {
        /* messages in registers: no vector memory operation inside the layer loop (the registers hold zeros before the first
         * iteration: the kernel clears them when it stages a fresh codeword) */
        uint32_t tabv = gc->sbplain[0][lane & 31];
#pragma nounroll
        for (int br = 0; br < nbr; ++br) {
            const int brn = br + 1 < nbr ? br + 1 : 0;
            const uint32_t tabn = gc->sbplain[brn][lane & 31]; /* next layer's edge table, a layer ahead of its use */
            const int deg = c->deg[br];
            uint32_t rowpar = 0;
            if (have_par) { /* syndrome bits of rows lane + 64 k of this layer as byte masks */
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t wv = sP[br * 8 + 2 * k + (lane >> 5)];
                    rowpar |= ((wv >> (lane & 31)) & 1u) ? (0xffu << (8 * k)) : 0u;
                }
            }
            DevTab4 tab;
            tab.c = c; tab.br = br; tab.sbv = tabv;
            const SwRow cur = regs_get(R, br);
            SwRow st;
            const uint32_t era_edges = ERA ? c->era_edges[br] : 0u;
            if (ERA) st = sw_layer_step<METHOD, 0, ERA>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme, era_edges, era_plane); /* rare: one instance */
            else if (deg == 23) st = sw_layer_step<METHOD, 23>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme);
            else if (deg == 22) st = sw_layer_step<METHOD, 22>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme);
            else st = sw_layer_step<METHOD, 0>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme);
            regs_put(R, br, st);
            tabv = tabn;
        }
        return;
    }
    SwRow cur = zero;
    if (!fresh) cur = rows[lane];
    uint32_t tabv = gc->sbplain[0][lane & 31];
    /* Nothing may be in flight when the layer loop is entered: the compiler merges the counter state of this path into the
     * loop header, and with loads pending here it waits in front of every layer as if they still were - in steady state that
     * is a wait for the row store issued a few instructions earlier (a memory round trip per layer). */
    __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0) */
    for (int br = 0; br < nbr; ++br) {
        /* next layer's messages and edge table: issued a whole layer ahead of their use; always a valid address (the last
         * layer re-reads layer 0, the first iteration reads what it is about to overwrite and ignores it) */
        const int brn = br + 1 < nbr ? br + 1 : 0;
        const SwRow nxt = rows[brn * LF_T4 + lane];
        const uint32_t tabn = gc->sbplain[brn][lane & 31];
        const int deg = c->deg[br]; /* (a bit mask over the layers instead of this scalar load was measured: 1 % slower) */
        uint32_t rowpar = 0;
        if (have_par) { /* syndrome bits of rows lane + 64 k of this layer as byte masks */
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t wv = sP[br * 8 + 2 * k + (lane >> 5)];
                rowpar |= ((wv >> (lane & 31)) & 1u) ? (0xffu << (8 * k)) : 0u;
            }
        }
        DevTab4 tab;
        tab.c = c; tab.br = br; tab.sbv = tabv;
        SwRow st;
        const uint32_t era_edges = ERA ? c->era_edges[br] : 0u;
        if (ERA) st = sw_layer_step<METHOD, 0, ERA>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme, era_edges, era_plane); /* rare: one instance */
        else if (deg == 23) st = sw_layer_step<METHOD, 23>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme);
        else if (deg == 22) st = sw_layer_step<METHOD, 22>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme);
        else st = sw_layer_step<METHOD, 0>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme);
        /* take the prefetched data BEFORE the store is issued: vector-memory operations retire in order, so a wait for these
         * loads placed after the store would also wait for the store's round trip, once per layer */
        cur = fresh ? zero : nxt;
        tabv = tabn;
        asm volatile("" : "+v"(cur.x[0]), "+v"(cur.x[1]), "+v"(cur.x[2]), "+v"(cur.cw), "+v"(cur.pa[0]), "+v"(cur.pa[1]), "+v"(tabv));
        __builtin_amdgcn_sched_barrier(0);
        if (rem > 0) rows[br * LF_T4 + lane] = st; /* the last layered iteration's messages are never read again */
    }
}

/* messages of a parking / resuming codeword between the registers and its slot in HBM (RM instances): every layer's transfer
 * in flight together (layers beyond the last one repeat it: no branches between the loads) */
__device__ __forceinline__ void regs_store(const SwRegs& R, SwRow* __restrict__ rows, int nbr, int lane)
{
#pragma unroll
    for (int br = 0; br < LF4_RM_LAYERS; ++br)
        if (br < nbr) rows[br * LF_T4 + lane] = regs_get(R, br);
}
__device__ __forceinline__ void regs_load(SwRegs& R, const SwRow* __restrict__ rows, int nbr, int lane)
{
    SwRow r[LF4_RM_LAYERS];
#pragma unroll
    for (int br = 0; br < LF4_RM_LAYERS; ++br) r[br] = rows[(br < nbr ? br : nbr - 1) * LF_T4 + lane];
#pragma unroll
    for (int br = 0; br < LF4_RM_LAYERS; ++br) regs_put(R, br, r[br]);
}
__device__ __forceinline__ void regs_clear(SwRegs& R)
{
    const lf4_vec z = (lf4_vec)(0u);
    R.x0 = z; R.x1 = z; R.x2 = z; R.cw = z; R.pa0 = z; R.pa1 = z;
}

/* ---- the decode kernel: one wave per codeword ---------------------------------------------------------- */
template <int METHOD, bool RM, bool EF2>
__global__ __launch_bounds__(LF_T4, 2) void lnsfaid_decode4_kernel(LfKernelArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    CCode c = (CCode)a.code;
    CCfg f = (CCfg)a.cfg;
    const int tid = (int)threadIdx.x;
    const int cw = (int)blockIdx.x;
    const int N = c->n_var, M = c->n_check, K = c->k_info, nw = c->n_words, pw = c->p_words;
    /* (the layer step addresses En by its LDS offset: the dynamic segment must start at 0, i.e. the kernel must have no static
     * LDS - checked on the host when a context picks its kernel, lnsfaid_capi.hip kernel_check) */
    uint32_t* sHard0 = (uint32_t*)smem;      /* bit-flipping stage: hard_ch and hard2 overlay the dead En */
    uint32_t* sHard2 = (uint32_t*)smem + nw;
    uint32_t* sHard = (uint32_t*)(smem + lf_lds_off_hard(N));
    uint32_t* sP = (uint32_t*)(smem + lf_lds_off_p(N, nw));
    int* sStat = (int*)(smem + lf_lds_off_stat(N, nw, pw));
    int* sRed = sStat + LNSFAID_GROUP;

    const int max_iter = f->max_iter, max_bf = f->max_bf;
    const int t_bf0 = max_iter + 1;   /* first bit-flipping decision point */
    const int t_end = t_bf0 + max_bf; /* both loops exhausted               */

    /* snapshot of the 32 lanes of this group: one load per lane (both halves of the wave hold the same 32 words), everything
     * else in registers - no LDS round trips in front of the early exits, which most workgroups of a relaunch take */
    const int g = cw >> 5, lane_in_group = cw & 31;
    const int sv = a.status_cur ? a.status_cur[g * LNSFAID_GROUP + (tid & 31)] : 0; /* null: first launch of a batch, every codeword fresh */
    const int my_status = __builtin_amdgcn_readlane(sv, lane_in_group);
    if (my_status & LF_DONE) { /* uniform exit */
        if (tid == 0) a.status_next[cw] = my_status;
        return;
    }
    if (tid == LNSFAID_GROUP) sRed[LF_ZERO_SLOT] = 0; /* the word unused synw slots point at */
    int kmax;
    {
        int v = sv & LF_PROG_MASK; /* maximum over lanes 0..31, same DPP pattern as add_reduce32 (values are not negative) */
        v = imax(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
        v = imax(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
        v = imax(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
        v = imax(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
        v = imax(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
        kmax = __builtin_amdgcn_readlane(v, 31);
    }
    const int all_same = __ballot(sv != my_status) == 0ull;
    LF_WG_SYNC();
    int prog = my_status & LF_PROG_MASK;

    uint32_t* g_en = (uint32_t*)(a.st_en + (size_t)cw * (size_t)N);
    SwRow* g_rows = (SwRow*)(a.st_rows + (size_t)cw * (size_t)(c->nbr * LF_T)); /* the 2-row kernel's slot: 16 B x 128 >= 24 B x 64 */
    uint32_t* g_bits = a.st_bits + (size_t)cw * (size_t)(3 * nw);
    int8_t* g_out = a.decoded + (size_t)cw * (size_t)N;

    /* parked on the group's front, not everybody there yet: nothing to do in this launch */
    if (prog != 0 && prog == kmax && !all_same) {
        if (tid == 0) { a.status_next[cw] = my_status; atomicAdd(a.remaining, 1u); }
        return;
    }

    /* all 32 lanes parked clean at the same decision point: the group stops there (the reference's break).  Every lane
     * wrote its hard decisions when it parked, so nothing is left to do but to say so. */
    if (my_status != 0 && all_same) {
        if (tid == 0) {
            a.status_next[cw] = my_status | LF_DONE;
            if (a.stats && lane_in_group == 0) {
                lnsfaid_group_stats st;
                st.iterations = prog <= max_iter ? prog - 1 : max_iter;
                st.bf_iterations = prog <= max_iter ? 0 : prog - t_bf0;
                a.stats[g] = st;
            }
        }
        return;
    }

    bool in_bf = max_bf > 0 && prog >= t_bf0 && prog != 0;
    LfLaneState ls = { 0, 0, 0, 0 };
    SwRegs R; /* RM: the codeword's compressed messages (dead in the bit-flipping stage) */
    if (RM) regs_clear(R);

    /* ---- bring the codeword's state on chip ---- */
    if (prog == 0) {
        /* input staging (CDecoder_FAID.cpp:217-255): lane l of group g is information row l of the [32][K] block followed by
         * parity row l of the [32][M] block; punctured tail erased; interleaved and biased for the layer step */
        const int8_t* gi = a.fix_input + (size_t)g * (size_t)LNSFAID_GROUP * (size_t)N;
        const int8_t* src_i = gi + (size_t)lane_in_group * (size_t)K;
        const int8_t* src_p = gi + (size_t)LNSFAID_GROUP * (size_t)K + (size_t)lane_in_group * (size_t)M;
        const int first_erased = N - c->puncture_tail;
        const int nbc = c->nbc;
        if ((((size_t)a.fix_input) & 3u) == 0u) {
            /* K, M and N are multiples of Z = 256: a block column is 64 aligned dwords of one of the two rows.  Lane d loads
             * dword d (variable nodes 4 d .. 4 d + 3) of LF_STAGE_COLS columns at a time, all loads in flight together, and
             * scatters the four bytes to their places in the interleaved image (node n: dword n mod 64, byte n div 64). */
            constexpr int SB = 23;
            const uint32_t base_d = ((16u * (uint32_t)tid) & 0xffu) + ((uint32_t)tid >> 4);
            const SwLds lds = SwLds();
            for (int cb0 = 0; cb0 < nbc; cb0 += SB) {
                uint32_t w[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int cb = cb0 + u;
                    if (cb < nbc) { /* uniform */
                        const int8_t* col = cb * LF_Z < K ? src_i + cb * LF_Z : src_p + (cb * LF_Z - K);
                        w[u] = ((const uint32_t*)col)[tid];
                    }
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int cb = cb0 + u;
                    if (cb < nbc) {
                        uint32_t x = w[u];
                        const int lim = first_erased - cb * LF_Z; /* nodes of this column from lim on are erased */
                        if (lim < LF_Z) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) if (4 * tid + k >= lim) x &= ~(0xffu << (8 * k));
                        }
                        x = ((x & 0x7f7f7f7fu) + (uint32_t)SW_BIAS_EN * 0x01010101u) ^ (x & 0x80808080u); /* + SW_BIAS_EN (< 128) per byte, no carries */
                        const uint32_t ad = (uint32_t)cb * 256u + base_d;
                        lds.wr8(ad, x); lds.wr8(ad + 4u, x >> 8); lds.wr8(ad + 8u, x >> 16); lds.wr8(ad + 12u, x >> 24);
                    }
                }
            }
        } else {
            for (int cb = 0; cb < nbc; ++cb) { /* caller's buffer not dword aligned: byte loads */
                uint32_t w = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int v = cb * LF_Z + tid + 64 * k;
                    int x = v < K ? src_i[v] : src_p[v - K];
                    if (v >= first_erased) x = 0;
                    w |= (uint32_t)((x + SW_BIAS_EN) & 0xff) << (8 * k);
                }
                lds4_wr((uint32_t)cb * 256u + 4u * (uint32_t)tid, w);
            }
        }
        LF_WG_SYNC();
        prog = 1;
    } else if (!in_bf) {
        copy_in<23>((uint32_t*)smem, g_en, N >> 2, tid);
        if (RM && prog >= 2) regs_load(R, g_rows, c->nbr, tid); /* parked in front of iteration 1: every Lmn is still 0 */
        LF_WG_SYNC();
    } else {
        copy_in<9>(sHard, g_bits, nw, tid);
        copy_in<9>(sHard0, g_bits + nw, nw, tid);
        copy_in<9>(sHard2, g_bits + 2 * nw, nw, tid);
        ls = a.st_lane[cw];
        LF_WG_SYNC();
    }

    bool parked = false;
    uint32_t pA = 0, pB = 0;
    /* ---- layered iterations (the syndrome stage in front of iteration prog is decision point prog) ---- */
    if (!in_bf) {
        while (prog < t_end && !(max_bf > 0 && prog >= t_bf0)) {
            /* the lane number as this iteration sees it: opaque, so that the per-lane addresses and masks of the syndrome stage and
             * the plane build are recomputed per iteration (a few dozen operations) instead of being hoisted out of the loop and
             * kept alive - spilled, with the messages in registers - through every layer */
            int tid_i = tid;
            asm volatile("" : "+v"(tid_i));
            if (METHOD == 0) { /* CLDPC::Decode has no syndrome stage and no early stop (CLDPC.cpp:287-2283) */
                main_step4<METHOD, false, RM>(c, f, a.code, g_rows, R, tid_i, prog, sP, false, false, 0u);
                prog++;
                continue;
            }
            bool lme = false, have_par = false;
            /* l_checksum_ and the unsatisfied count are consumed only inside the error-floor window
             * (nombre_iterations <= floor_iter_thresh: OMS selective offset CDecoder_OMS.cpp:388, 2B1C tables
             * CDecoder_FAID.cpp:714) and never by DecodeMethod 2; elsewhere only unsat != 0 matters */
            const bool needs_checksums = max_iter - prog <= f->floor_iter_thresh; /* never for the shipped DecodeMethod 2: -1 */
            /* behind the group's front (the snapshot shows a lane parked beyond this point) the group is known to go on, and
             * outside the window nothing else reads the syndrome: a catching-up codeword skips the stage altogether */
            const bool must_know = needs_checksums || prog >= kmax;
            if (must_know && (needs_checksums || !layer0_dirty4(c, tid_i))) {
                build_plane4<false>(c, sHard, 0, tid_i);
                int unsat;
                if (RM || syn_cache_fits(c->nbr)) { /* (RM: a code of up to LF4_RM_LAYERS layers always fits) all table entries of the walk loaded together: one memory round trip, not one per round */
                    SynCache sc;
                    syn_cache_load(a.code, c->nbr, tid_i, sc);
                    unsat = syndrome<LF_T4, false, true>(c, a.code, sP, tid_i, pA, pB, sRed, &sc);
                } else {
                    unsat = syndrome<LF_T4, false>(c, a.code, sP, tid_i, pA, pB, sRed);
                }
                /* clean on the group's front: park, unless a group mate is known to have passed this point */
                if (unsat == 0 && prog >= kmax && !group_passed(a.live, g, prog, tid_i)) {
                    /* the messages leave the registers here, not in the common epilogue: there the compiler would have to keep
                     * them alive through the whole bit-flipping stage */
                    if (RM && prog >= 2) regs_store(R, g_rows, c->nbr, tid_i);
                    parked = true;
                    break;
                }
                if (LF4_OMS(METHOD)) lme = imin(unsat, 255) < (int)(uint8_t)f->floor_err_count; /* CDecoder_OMS.cpp:328 */
                else lme = imin(unsat, 127) < (int)(int8_t)f->floor_err_count;              /* CDecoder_FAID.cpp:619 */
                have_par = true;
            }
            publish_pass(a.live, cw, prog, tid_i);
            if (EF2 && f->ef == 2 && needs_checksums && have_par && lme) {
                /* EF_ELIMINATION 2 inside the window, few unsatisfied checks: this iteration erases (CDecoder_FAID.cpp:673-680) */
                build_erasure_plane4(c, a.code, sHard, sP, f->W, tid_i);
                main_step4<METHOD, EF2, RM>(c, f, a.code, g_rows, R, tid_i, prog, sP, true, lme, lf_lds_off_hard(N));
            } else {
                main_step4<METHOD, false, RM>(c, f, a.code, g_rows, R, tid_i, prog, sP, have_par && needs_checksums, lme, 0u);
            }
            prog++;
        }
        if (!parked && prog < t_end) {
            /* the layered loop ran out: enter the bit-flipping stage (CDecoder_FAID.cpp:6411-6428) */
            uint32_t conf[LF_MAX_BC * 8 / LF_T4]; /* this lane's share of the 2B1C confidence plane */
            if (METHOD == 5) {
                build_plane4<true>(c, sHard, f->hard2_thr, tid); /* staged where the hard plane will go */
#pragma unroll
                for (int k = 0; k < LF_MAX_BC * 8 / LF_T4; ++k) conf[k] = (tid + k * LF_T4 < nw) ? sHard[tid + k * LF_T4] : 0u;
                LF_WG_SYNC();
            }
            build_plane4<false>(c, sHard, 0, tid);
            /* En is dead from here on: its bytes take hard_ch (= hard) and hard2 */
            copy_out<9>(sHard0, sHard, nw, tid);
            if (METHOD == 5) {
#pragma unroll
                for (int k = 0; k < LF_MAX_BC * 8 / LF_T4; ++k) if (tid + k * LF_T4 < nw) sHard2[tid + k * LF_T4] = conf[k];
            }
            ls.Th = (int8_t)f->W; ls.l0 = 0; ls.l1 = 0; ls.t = 1;
            in_bf = true;
            LF_WG_SYNC();
        }
    }
    /* ---- bit-flipping iterations.  Nothing of the layer step is alive here, so the lanes keep their entries of the walk
     * tables in registers for the whole stage (no table load, hence no exposed memory latency, per iteration) ---- */
    if (in_bf && !parked) {
        if ((RM || syn_cache_fits(c->nbr)) && (METHOD == 3 || bf_cache_fits(c, f))) {
            SynCache sc;
            BfCache bc;
            syn_cache_load(a.code, c->nbr, tid, sc);
            if (METHOD != 3) bf_cache_load(c, a.code, tid, bc);
            while (prog < t_end) {
                const int unsat = syndrome<LF_T4, false, true>(c, a.code, sP, tid, pA, pB, sRed, &sc);
                if (unsat == 0 && prog >= kmax && !group_passed(a.live, g, prog, tid)) { parked = true; break; }
                publish_pass(a.live, cw, prog, tid);
                if (METHOD == 3) bf_step_plain<LF_T4>(c, f, a.code, sHard, sHard2 + nw /* 4 count planes in the dead En */, sP, tid, sRed);
                else bf_step<LF_T4, METHOD, true>(c, f, a.code, sHard, sHard0, sHard2, sP, tid, ls, sRed, &bc);
                prog++;
            }
        } else {
            while (prog < t_end) {
                const int unsat = syndrome<LF_T4, false>(c, a.code, sP, tid, pA, pB, sRed);
                if (unsat == 0 && prog >= kmax && !group_passed(a.live, g, prog, tid)) { parked = true; break; }
                publish_pass(a.live, cw, prog, tid);
                if (METHOD == 3) bf_step_plain<LF_T4>(c, f, a.code, sHard, sHard2 + nw /* 4 count planes in the dead En */, sP, tid, sRed);
                else bf_step<LF_T4, METHOD>(c, f, a.code, sHard, sHard0, sHard2, sP, tid, ls, sRed);
                prog++;
            }
        }
    }

    const bool finished = prog >= t_end;
    if (finished) {
        if (!in_bf) build_plane4<false>(c, sHard, 0, tid);
        write_decoded(sHard, g_out, N, tid);
        if (tid == 0) {
            a.status_next[cw] = prog | LF_DONE;
            if (a.stats && lane_in_group == 0) {
                lnsfaid_group_stats st;
                st.iterations = prog <= max_iter ? prog - 1 : max_iter;
                st.bf_iterations = prog <= max_iter ? 0 : prog - t_bf0;
                a.stats[g] = st;
            }
        }
    } else {
        /* park clean at decision point prog: state back to HBM for the case that the group goes on, and the hard decisions
         * (the syndrome stage has just built the plane from this En; in the bit-flipping stage the plane is the state) as the
         * output for the case that it stops here */
        if (!in_bf) {
            copy_out<23>(g_en, (const uint32_t*)smem, N >> 2, tid); /* (RM: the messages were stored where the codeword parked) */
        } else {
            copy_out<9>(g_bits, sHard, nw, tid);
            copy_out<9>(g_bits + nw, sHard0, nw, tid);
            copy_out<9>(g_bits + 2 * nw, sHard2, nw, tid);
            if (tid == 0) a.st_lane[cw] = ls;
        }
        write_decoded(sHard, g_out, N, tid);
        if (tid == 0) { a.status_next[cw] = prog; atomicAdd(a.remaining, 1u); }
    }
}

/* Instances: messages in registers (RM) for codes of up to LF4_RM_LAYERS layers, streamed through HBM otherwise; the erasing
 * layer step of EF_ELIMINATION 2 (Decode_FAID only) lives in an instance of its own, so that the common ones do not carry its
 * registers. */
extern "C" int lf_decode4_rm_layers(void) { return LF4_RM_LAYERS; }

/* the instance a configuration runs on (for hipFuncGetAttributes / the occupancy query, and for the launch) */
extern "C" const void* lf_decode4_func(int method, int ef, int rm)
{
    if (method == 2 && ef == 2) return (const void*)lnsfaid_decode4_kernel<2, false, true>;
#define LF4_FUNC(M) case M: return rm ? (const void*)lnsfaid_decode4_kernel<M, true, false> : (const void*)lnsfaid_decode4_kernel<M, false, false>;
    switch (method) {
    case 0: return (const void*)lnsfaid_decode4_kernel<0, false, false>; /* (16-level search: with the messages in registers too the layer step spills) */
        LF4_FUNC(1) LF4_FUNC(2) LF4_FUNC(3) LF4_FUNC(4) LF4_FUNC(5)
    default: return nullptr;
    }
#undef LF4_FUNC
}
extern "C" int lf_decode4_threads(void) { return LF_T4; }

extern "C" hipError_t lf_launch_decode4(int method, int ef, int rm, const LfKernelArgs* args, size_t lds_bytes, hipStream_t stream)
{
    const void* fn = lf_decode4_func(method, ef, rm);
    if (!fn) return hipErrorInvalidValue;
    void* kargs[] = { (void*)args };
    return hipLaunchKernel(fn, dim3((unsigned)args->n_cw), dim3(LF_T4), kargs, lds_bytes, stream);
}
