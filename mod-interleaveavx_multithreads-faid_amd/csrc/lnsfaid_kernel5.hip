/*
 * lnsfaid_kernel5.hip - the four-rows-per-lane decode kernel with TWO wavefronts per codeword (gfx950).  EXPERIMENTAL and opt-in
 * (lnsfaid_select_waves(ctx, 2) / LNSFAID_WAVES_PER_CODEWORD=2); bit-exact with lnsfaid_kernel4.hip, and SLOWER: 14.85 ms against
 * 13.37 ms per launch of the headline batch (profiles/r03_two_waves/).  Kept because it is the measurement that
 * closes the "more waves per SIMD" question for this decoder, with the split layer step it needs (lnsfaid_swar.h, WAVES = 2).
 *
 * Same decoder, protocol, HBM state and LDS image as lnsfaid_kernel4.hip.  The LDS image of a codeword allows 8 codewords per CU;
 * with one wave each that is two waves per SIMD.  Here a workgroup is 128 threads: wave 0 is the kernel of lnsfaid_kernel4.hip
 * (staging, syndromes, bit flipping, park / resume, output - everything that is not a layered iteration), and inside a layered
 * iteration the layer's edges are dealt to the two waves by the parity of their index (sw_layer_step<..., WAVES = 2>): four
 * waves per SIMD, 128 registers per wave, the compressed messages streamed through HBM.  Wave 1 sleeps at a barrier whenever
 * wave 0 is outside the layer loop.  DecodeMethods 1..5 without the erasing EF_ELIMINATION 2.
 */
#include <hip/hip_runtime.h>

/* Wave 0 runs the single-wave phases of lnsfaid_kernel4.hip / lnsfaid_phases.h on its own: their LF_WG_SYNC() (__syncthreads():
 * a no-op barrier plus an LDS fence in a one-wave workgroup) must not become an s_barrier that waits for wave 1. */
__device__ __forceinline__ void lf5_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
#define LF_WG_SYNC() lf5_wave_sync()
/* both waves of the workgroup */
__device__ __forceinline__ void lf5_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#include "lnsfaid_rows4.h"

/* exchange area of the two waves: the hard-decision plane's LDS, dead between the syndrome stage and the next one; slot k of
 * lane i at word k * 64 + i */
struct Xch5 {
    uint32_t base; /* LDS byte offset + 4 * lane */
    __device__ __forceinline__ void put(int k, uint32_t v) const { lds4_wr(base + 256u * (uint32_t)k, v); }
    __device__ __forceinline__ uint32_t get(int k) const { return lds4_rd(base + 256u * (uint32_t)k); }
    __device__ __forceinline__ void barrier() const { lf5_barrier(); }
};
#define LF5_CMD_SLOT 5 /* reduction-scratch word through which role 0 tells role 1 what to do next: 0 = the kernel ends, else the
                        * number of the layered iteration to run */

/* ---- one layered iteration, both waves (lnsfaid_swar.h does the rows) ---- */
template <int METHOD, int WAVE>
__device__ __forceinline__ void main_step5(CCode c, CCfg f, const LfDevCode* gc, SwRow* __restrict__ rows, int lane, int it, const uint32_t* sP,
                           bool have_par, bool lme, uint32_t xch_base)
{
    it = __builtin_amdgcn_readfirstlane(it);
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
    if (LF4_OMS(METHOD)) sw_oms_tables(p);
    const int nbr = c->nbr;
    const SwLds lds = SwLds();
    Xch5 xch;
    xch.base = xch_base + 4u * (uint32_t)lane;
    const SwRow zero = { { 0u, 0u, 0u }, 0u, { 0u, 0u } };
    SwRow cur = zero;
    if (!fresh) cur = rows[lane];
    uint32_t tabv = WAVE == 0 ? gc->sbplain[0][lane & 31] : 0u;
    __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0): nothing in flight when the layer loop is entered (lnsfaid_kernel4.hip) */
    for (int br = 0; br < nbr; ++br) {
        const int brn = br + 1 < nbr ? br + 1 : 0;
        const SwRow nxt = rows[brn * LF_T4 + lane]; /* (both waves read the record: wave 1 needs the sign words and the magnitudes) */
        const uint32_t tabn = WAVE == 0 ? gc->sbplain[brn][lane & 31] : 0u;
        const int deg = c->deg[br];
        uint32_t rowpar = 0;
        if (have_par && WAVE == 0) { /* syndrome bits of rows lane + 64 k of this layer as byte masks (wave 0 forms the magnitudes) */
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t wv = sP[br * 8 + 2 * k + (lane >> 5)];
                rowpar |= ((wv >> (lane & 31)) & 1u) ? (0xffu << (8 * k)) : 0u;
            }
        }
        DevTab4 tab;
        tab.c = c; tab.br = br; tab.sbv = tabv;
        SwRow st;
        if (deg == 23) st = sw_layer_step<METHOD, 23, false, 2, WAVE>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme, 0u, 0u, xch);
        else if (deg == 22) st = sw_layer_step<METHOD, 22, false, 2, WAVE>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme, 0u, 0u, xch);
        else st = sw_layer_step<METHOD, 0, false, 2, WAVE>(lds, tab, p, K, (uint32_t)lane, deg, cur, fresh, rowpar, lme, 0u, 0u, xch);
        cur = fresh ? zero : nxt;
        tabv = tabn;
        asm volatile("" : "+v"(cur.x[0]), "+v"(cur.x[1]), "+v"(cur.x[2]), "+v"(cur.cw), "+v"(cur.pa[0]), "+v"(cur.pa[1]), "+v"(tabv));
        __builtin_amdgcn_sched_barrier(0);
        if (WAVE == 0 && rem > 0) rows[br * LF_T4 + lane] = st; /* the last layered iteration's messages are never read again */
    }
    /* the record stores of the last layers must be visible to wave 1's loads of the next iteration: the barrier in front of the
     * next iteration (command hand-over) follows a release fence, and both waves of a workgroup share the CU's caches */
}

/* ---- wave 0: the decode kernel of lnsfaid_kernel4.hip (messages streamed through HBM); every layered iteration is announced to
 * wave 1 through the command word and run by both ---------------------------------------------------------- */
template <int METHOD>
__device__ __forceinline__ void decode5_wave0(const LfKernelArgs& a, unsigned char* smem, const int tid)
{
    CCode c = (CCode)a.code;
    CCfg f = (CCfg)a.cfg;
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
    const uint32_t xch_base = lf_lds_off_hard(N);

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
                if (syn_cache_fits(c->nbr)) { /* all table entries of the walk loaded together: one memory round trip, not one per round */
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
                    parked = true;
                    break;
                }
                if (LF4_OMS(METHOD)) lme = imin(unsat, 255) < (int)(uint8_t)f->floor_err_count; /* CDecoder_OMS.cpp:328 */
                else lme = imin(unsat, 127) < (int)(int8_t)f->floor_err_count;              /* CDecoder_FAID.cpp:619 */
                have_par = true;
            }
            publish_pass(a.live, cw, prog, tid_i);
            /* hand the iteration to wave 1 (it sleeps at this barrier), then run wave 0's share of it */
            if (tid_i == 0) sRed[LF5_CMD_SLOT] = prog;
            lf5_barrier();
            main_step5<METHOD, 0>(c, f, a.code, g_rows, tid_i, prog, sP, have_par && needs_checksums, lme, xch_base);
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
        if (syn_cache_fits(c->nbr) && (METHOD == 3 || bf_cache_fits(c, f))) {
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
            copy_out<23>(g_en, (const uint32_t*)smem, N >> 2, tid);
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

/* ---- the kernel: 128 threads per codeword ---- */
template <int METHOD>
__global__ __launch_bounds__(2 * LF_T4, 4) void lnsfaid_decode5_kernel(LfKernelArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    CCode c = (CCode)a.code;
    const int N = c->n_var, nw = c->n_words, pw = c->p_words;
    int* sRed = (int*)(smem + lf_lds_off_stat(N, nw, pw)) + LNSFAID_GROUP;
    /* Fixed roles: the dispatcher already spreads the first and the second waves of the workgroups evenly over the four SIMDs of a CU
     * (tools/ubench/hwid.hip: pairs (0,2) (1,3) (2,1) (3,0) in turn), so every SIMD holds two waves of either role.  (Swapping the
     * roles by the parity of the wave slot skews that to 452 : 572 and costs 10 %: profiles/r03_two_waves/.) */
    const int role = __builtin_amdgcn_readfirstlane(wave);
    if (role == 0) {
        decode5_wave0<METHOD>(a, smem, lane);
        if (lane == 0) sRed[LF5_CMD_SLOT] = 0; /* every path of wave 0 ends here: wave 1 may go */
        lf5_barrier();
    } else {
        CCfg f = (CCfg)a.cfg;
        const int cw = (int)blockIdx.x;
        SwRow* g_rows = (SwRow*)(a.st_rows + (size_t)cw * (size_t)(c->nbr * LF_T));
        const uint32_t* sP = (const uint32_t*)(smem + lf_lds_off_p(N, nw));
        for (;;) {
            lf5_barrier();
            const int it = __builtin_amdgcn_readfirstlane(sRed[LF5_CMD_SLOT]);
            if (it == 0) break;
            main_step5<METHOD, 1>(c, f, a.code, g_rows, lane, it, sP, false, false, lf_lds_off_hard(N));
        }
    }
}

extern "C" const void* lf_decode5_func(int method)
{
    switch (method) {
    case 1: return (const void*)lnsfaid_decode5_kernel<1>;
    case 2: return (const void*)lnsfaid_decode5_kernel<2>;
    case 3: return (const void*)lnsfaid_decode5_kernel<3>;
    case 4: return (const void*)lnsfaid_decode5_kernel<4>;
    case 5: return (const void*)lnsfaid_decode5_kernel<5>;
    default: return nullptr;
    }
}
extern "C" int lf_decode5_threads(void) { return 2 * LF_T4; }

extern "C" hipError_t lf_launch_decode5(int method, const LfKernelArgs* args, size_t lds_bytes, hipStream_t stream)
{
    const void* fn = lf_decode5_func(method);
    if (!fn) return hipErrorInvalidValue;
    void* kargs[] = { (void*)args };
    return hipLaunchKernel(fn, dim3((unsigned)args->n_cw), dim3(2 * LF_T4), kargs, lds_bytes, stream);
}
