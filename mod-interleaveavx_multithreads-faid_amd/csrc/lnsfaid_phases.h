/*
 * lnsfaid_phases.h — the phases of a decode that do not depend on how a layer's rows are spread over lanes: the bit-parallel
 * syndrome on the hard-decision plane, the bit-sliced DTBF / 2B1C flip and the plain bit flipping of Decode_OMSBF.  Shared by
 * the two decode kernels: T = 128 threads per codeword (two rows per lane on packed 16-bit operations, lnsfaid_kernels.hip)
 * and T = 64 (four rows per lane, byte-parallel, lnsfaid_kernel4.hip).
 */
#ifndef LNSFAID_PHASES_H
#define LNSFAID_PHASES_H

/* Workgroup-wide synchronisation of the phases below and of the kernels that include them.  A kernel whose workgroup holds a wave
 * that does not take part in these phases (lnsfaid_kernel5.hip: its second wave sleeps at a barrier meanwhile) defines it
 * before including this header as the fence its working wave needs instead. */
#ifndef LF_WG_SYNC
#define LF_WG_SYNC() __syncthreads()
#endif

#include <hip/hip_runtime.h>

#include "lnsfaid_device.h"

/* tables are read through the constant address space so that uniform accesses become scalar loads */
typedef const __attribute__((address_space(4))) LfDevCode* CCode;
typedef const __attribute__((address_space(4))) LfDevCfg* CCfg;

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

/* sum of a wave-uniform per-wave value over the T / 64 waves of the workgroup; its barriers also order LDS traffic */
template <int T>
__device__ __forceinline__ int block_sum(int wave_value, int tid, int* sRed)
{
    if (T == 64) {
        LF_WG_SYNC(); /* one wave: nothing to add up, LDS operations of a wave execute in order */
        return wave_value;
    }
    if ((tid & 63) == 0) sRed[tid >> 6] = wave_value;
    LF_WG_SYNC();
    int total = 0;
    for (int w = 0; w < T / 64; ++w) total += sRed[w];
    LF_WG_SYNC();
    return total;
}

/* 64 bits starting at bit `o` (mod 256) of a 256-bit block kept as 8 LDS words */
__device__ __forceinline__ void window64(const uint32_t* blk, uint32_t o, uint32_t& lo, uint32_t& hi)
{
    const uint32_t q = o >> 5, r = o & 31u;
    const uint32_t w0 = blk[q & 7u], w1 = blk[(q + 1u) & 7u], w2 = blk[(q + 2u) & 7u];
    lo = __builtin_amdgcn_alignbit(w1, w0, r);
    hi = __builtin_amdgcn_alignbit(w2, w1, r);
}

/* sum over lanes 0..31 / 32..63 of a wave, results in lanes 31 / 63 (same DPP pattern as xor_reduce32) */
__device__ __forceinline__ int add_reduce32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); /* row_shr:8 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1, 3 */
    return v;
}

/* ---- bit-parallel syndrome of the hard-decision plane sHard (CDecoder_FAID.cpp:291-343, :6443-6491) -----------
 * The parity of the 32 rows [32k, 32k+32) of layer br is the XOR over the row's circulants of 32 consecutive (mod 256)
 * bits of the circulant's block column.  One lane per (layer, k): it walks the layer's circulants with a host-built
 * table of {LDS addresses of the two plane words, bit offset} (LfDevCode::synw), two ds_read_b32 + one v_alignbit_b32
 * + one v_xor per circulant and no cross-lane reduction; slots beyond the row degree point at a zero word.
 * Writes the parity plane sP (bit r = l_checksum_[r]), returns the number of unsatisfied checks; with ROWBITS pA / pB
 * get bit br = parity of this thread's rows tid / tid + 128 in layer br (the error-floor tables need them). */
/* The bit-flipping stage runs the same walk up to _maxBFiter times with nothing else alive in the registers (the layer step's
 * state is dead), and with two waves per SIMD nothing hides the latency of the table loads: the one-wave kernel keeps its
 * lanes' table entries in registers for the whole stage (codes of up to LF_SYN_ROUNDS * 32 / 8 layers). */
#define LF_SYN_ROUNDS 3
#define LF_SYN_JP (LF_MAX_DEG / 2)
struct SynCache {
    uint2 e[LF_SYN_ROUNDS][LF_SYN_JP];
};
__device__ __forceinline__ bool syn_cache_fits(int nbr) { return nbr * 16 <= LF_SYN_ROUNDS * 64; }
__device__ __forceinline__ void syn_cache_load(const LfDevCode* gc, int nbr, int tid, SynCache& sc)
{
#pragma unroll
    for (int r = 0; r < LF_SYN_ROUNDS; ++r) {
        const int ht = tid + 64 * r;
        const int task = ht < nbr * 16 ? ht >> 1 : 0, part = ht & 1;
#pragma unroll
        for (int j = 0; j < LF_SYN_JP; ++j) sc.e[r][j] = gc->synw[task >> 3][part * LF_SYN_JP + j][task & 7];
    }
}

template <int T, bool ROWBITS, bool CACHED = false>
__device__ __forceinline__ int syndrome(CCode c, const LfDevCode* gc, uint32_t* sP, int tid, uint32_t& pA, uint32_t& pB, int* sRed,
                        const SynCache* sc = nullptr)
{
    typedef const __attribute__((address_space(3))) uint32_t lds_u32;
    const int nbr = c->nbr;
    int cnt = 0;
    if (CACHED) {
        static_assert(!CACHED || T == 64, "register tables: one-wave kernel only");
#pragma unroll
        for (int r = 0; r < LF_SYN_ROUNDS; ++r) {
            const int ht = tid + 64 * r;
            if (ht < nbr * 16) {
                uint32_t acc = 0;
                /* all reads of the round in flight before the first use (left alone, a compiler short of registers pairs
                 * "two reads, wait, combine": one LDS round trip per circulant) */
                uint32_t w0[LF_SYN_JP], w1[LF_SYN_JP];
#pragma unroll
                for (int j = 0; j < LF_SYN_JP; ++j) {
                    const uint2 e = sc->e[r][j];
                    w0[j] = *(lds_u32*)(size_t)(e.x & 0xffffu); w1[j] = *(lds_u32*)(size_t)(e.x >> 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < LF_SYN_JP; ++j) acc ^= __builtin_amdgcn_alignbit(w1[j], w0[j], sc->e[r][j].y);
                acc ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)acc, 0xb1, 0xf, 0xf, false); /* the other half of the circulant list */
                if (!(ht & 1)) {
                    sP[ht >> 1] = acc;
                    cnt += __popc(acc);
                }
            }
        }
    } else
    {
    /* T = 64: 96 tasks would leave half the wave idle in a second round, so every task is cut in two halves of the circulant
     * list held by neighbouring lanes (192 half tasks = three full rounds of 12 circulants instead of two rounds of 24) */
    constexpr int PARTS = T == 64 ? 2 : 1, JP = LF_MAX_DEG / PARTS;
    static_assert(LF_MAX_DEG % PARTS == 0, "circulant slots must split evenly");
    for (int ht = tid; ht < nbr * 8 * PARTS; ht += T) {
        const int task = ht / PARTS, part = ht % PARTS;
        const uint2* tab = &gc->synw[task >> 3][part * JP][task & 7];
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < JP; ++j) {
            const uint2 e = tab[j * 8];
            const uint32_t w0 = *(lds_u32*)(size_t)(e.x & 0xffffu), w1 = *(lds_u32*)(size_t)(e.x >> 16);
            acc ^= __builtin_amdgcn_alignbit(w1, w0, e.y);
        }
        if (PARTS == 2) acc ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)acc, 0xb1, 0xf, 0xf, false); /* quad_perm [1,0,3,2]: the other half */
        if (part == 0) {
            sP[task] = acc;
            cnt += __popc(acc);
        }
    }
    }
    cnt = add_reduce32(cnt);
    const int wave_cnt = __builtin_amdgcn_readlane(cnt, 31) + __builtin_amdgcn_readlane(cnt, 63);
    const int total = block_sum<T>(wave_cnt, tid, sRed); /* its barriers also publish sP */
    if (ROWBITS) {
        uint32_t a = 0, b = 0;
        const uint32_t* p = sP + (tid >> 5);
        const uint32_t sh = (uint32_t)tid & 31u;
        for (int br = 0; br < nbr; ++br) {
            a |= ((p[br * 8] >> sh) & 1u) << br;
            b |= ((p[br * 8 + 4] >> sh) & 1u) << br;
        }
        pA = a; pB = b;
    }
    return total;
}

/* ---- bit-flipping iteration after a dirty syndrome (CDecoder_FAID.cpp:6787-6845, :7084-7086;
 *      CDecoder_FAID_2B1C.cpp:6801-6814) --------------------------------------------------------------- */
__device__ __forceinline__ int bit_of(const uint32_t* words, int v) { return (int)((words[v >> 5] >> (v & 31)) & 1u); }

/* register copy of this lane's units of the bit-sliced flip (one-wave kernel, see SynCache) */
#define LF_BF_ROUNDS 4
struct BfCache {
    uint32_t cc[LF_BF_ROUNDS][3];
    int w0[LF_BF_ROUNDS];
};
__device__ __forceinline__ bool bf_cache_fits(CCode c, CCfg f) { return f->bf_fast && c->n_wcols * 4 <= LF_BF_ROUNDS * 64; }
__device__ __forceinline__ void bf_cache_load(CCode c, const LfDevCode* gc, int tid, BfCache& bc)
{
    const int units = c->n_wcols * 4;
#pragma unroll
    for (int r = 0; r < LF_BF_ROUNDS; ++r) {
        const int u = tid + 64 * r;
        const int cb = gc->wcol[u < units ? u >> 2 : 0];
        bc.w0[r] = cb * 8 + 2 * (u & 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) bc.cc[r][k] = gc->colcirc[cb][k];
    }
}

template <int T, int METHOD, bool CACHED = false>
__device__ __forceinline__ void bf_step(CCode c, CCfg f, const LfDevCode* gc, uint32_t* sHard, const uint32_t* sHard0, uint32_t* sHard2,
                        const uint32_t* sP, int tid, LfLaneState& ls, int* sRed, const BfCache* bc = nullptr)
{
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

    if (f->bf_fast) {
        /* bit-sliced: one (weight-3 block column, 64-VN window) per lane */
        const int units = c->n_wcols * 4;
        static_assert(!CACHED || T == 64, "register tables: one-wave kernel only");
        auto unit = [&](int u, uint32_t cc0, uint32_t cc1, uint32_t cc2, int w0) {
            const uint32_t win = (uint32_t)(u & 3);
            const uint32_t cc[3] = { cc0, cc1, cc2 };
            uint32_t plo[3], phi[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                window64(sP + (cc[k] & 0xffu) * 8u, (64u * win - ((cc[k] >> 8) & 0xffu)) & 255u, plo[k], phi[k]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t p1 = h ? phi[0] : plo[0], p2 = h ? phi[1] : plo[1], p3 = h ? phi[2] : plo[2];
                const uint32_t hd = sHard[w0 + h];
                const uint32_t fl = alpha ? (hd ^ sHard0[w0 + h]) : 0u; /* already flipped once: +alpha */
                const uint32_t s0 = p1 ^ p2 ^ p3, cy = (p1 & p2) | (p1 & p3) | (p2 & p3);
                uint32_t m;
                if (Th <= 1) m = s0 | cy | fl;          /* votes + fl >= 1 */
                else if (Th == 2) m = cy | (s0 & fl);
                else if (Th == 3) m = cy & (s0 | fl);
                else if (Th == 4) m = cy & s0 & fl;
                else m = 0u;
                any |= (m != 0u);
                if (METHOD == 5) {
                    const uint32_t h2 = sHard2[w0 + h];
                    if (big) { sHard[w0 + h] = hd ^ m; sHard2[w0 + h] = h2 ^ m; }
                    else { sHard[w0 + h] = hd ^ (m & ~h2); sHard2[w0 + h] = h2 & ~m; }
                } else {
                    sHard[w0 + h] = hd ^ m;
                }
            }
        };
        if (CACHED) {
#pragma unroll
            for (int r = 0; r < LF_BF_ROUNDS; ++r) {
                const int u = tid + 64 * r;
                if (u < units) unit(u, bc->cc[r][0], bc->cc[r][1], bc->cc[r][2], bc->w0[r]);
            }
        } else {
            for (int u = tid; u < units; u += T) {
                const int cb = gc->wcol[u >> 2];
                unit(u, gc->colcirc[cb][0], gc->colcirc[cb][1], gc->colcirc[cb][2], cb * 8 + 2 * (u & 3));
            }
        }
    } else {
        /* generic column weight / alpha: one variable node per lane and step */
        const int nbc = c->nbc;
        for (int cb = 0; cb < nbc; ++cb) {
            if (c->col_weight[cb] != W) continue; /* VN_weight_[v] == REGULAR_COL_WEIGHT, :6806 */
            for (int half = 0; half < LF_Z / T; ++half) { /* T lanes cover the 256 variable nodes of a block column in LF_Z / T steps */
                const int x = tid + half * T;
                const int v = cb * LF_Z + x;
                int vote = 0;
                for (int k = 0; k < W; ++k) {
                    const uint32_t cc = gc->colcirc[cb][k];
                    vote += bit_of(sP, (int)((cc & 0xffu) << 8) + (int)(((uint32_t)x - ((cc >> 8) & 0xffu)) & 0xffu));
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
        }
    }
    ls.Th = Th; ls.l0 = l0; ls.l1 = l1;
    const unsigned long long anyw = __ballot(any);
    ls.t = block_sum<T>(anyw != 0ull ? 1 : 0, tid, sRed) != 0; /* barriers also order the plane updates */
}

/* ---- plain bit flipping of Decode_OMSBF (CDecoder_OMSBF.cpp:2969-3514): flip every variable node whose vote count
 * reaches min(max vote of the frame, cap).  Votes of all block columns are counted bit-sliced (4 planes, column
 * weight <= 15), the frame's maximum is found from "some count >= k" flags, then the planes are compared with the
 * threshold. */
__device__ __forceinline__ uint32_t votes_ge(uint32_t c3, uint32_t c2, uint32_t c1, uint32_t c0, int k)
{
    /* bitwise 4-bit comparator: count < k, scanned from the most significant plane */
    uint32_t lt = 0u, eqm = 0xffffffffu;
    const uint32_t pl[4] = { c0, c1, c2, c3 };
#pragma unroll
    for (int b = 3; b >= 0; --b) {
        if ((k >> b) & 1) { lt |= eqm & ~pl[b]; eqm &= pl[b]; }
        else eqm &= ~pl[b];
    }
    return ~lt;
}

template <int T>
__device__ __forceinline__ void bf_step_plain(CCode c, CCfg f, const LfDevCode* gc, uint32_t* sHard, uint32_t* sCnt, const uint32_t* sP, int tid,
                              int* sRed)
{
    const int nw = c->n_words;
    const int units = c->nbc * 4;
    uint32_t seen = 0; /* bit k: some variable node of this lane's units has >= k votes (k = 1..15) */
    for (int u = tid; u < units; u += T) {
        const int cb = u >> 2;
        const uint32_t win = (uint32_t)(u & 3);
        const int wgt = gc->col_weight[cb];
        uint32_t c0[2] = { 0u, 0u }, c1[2] = { 0u, 0u }, c2[2] = { 0u, 0u }, c3[2] = { 0u, 0u };
        for (int k = 0; k < wgt; ++k) {
            const uint32_t cc = gc->colcirc[cb][k];
            uint32_t x[2];
            window64(sP + (cc & 0xffu) * 8u, (64u * win - ((cc >> 8) & 0xffu)) & 255u, x[0], x[1]);
#pragma unroll
            for (int h = 0; h < 2; ++h) { /* ripple-carry increment of the 4-bit counters where x is set */
                uint32_t carry = x[h], t;
                t = c0[h] & carry; c0[h] ^= carry; carry = t;
                t = c1[h] & carry; c1[h] ^= carry; carry = t;
                t = c2[h] & carry; c2[h] ^= carry; carry = t;
                c3[h] ^= carry;
            }
        }
        const int w0 = cb * 8 + 2 * (int)win;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            sCnt[w0 + h] = c0[h]; sCnt[nw + w0 + h] = c1[h]; sCnt[2 * nw + w0 + h] = c2[h]; sCnt[3 * nw + w0 + h] = c3[h];
#pragma unroll
            for (int k = 1; k < 16; ++k) seen |= (votes_ge(c3[h], c2[h], c1[h], c0[h], k) != 0u) ? (1u << k) : 0u;
        }
    }
    /* OR of `seen` over the workgroup */
    for (int o = 32; o > 0; o >>= 1) seen |= (uint32_t)__shfl_xor((int)seen, o);
    if (T > 64) {
        if ((tid & 63) == 0) sRed[tid >> 6] = (int)seen;
        LF_WG_SYNC();
        seen = 0;
        for (int w = 0; w < T / 64; ++w) seen |= (uint32_t)sRed[w];
    }
    LF_WG_SYNC();
    const int max_vote = seen ? 31 - __clz((int)seen) : 1; /* max_vote starts at 1 (CDecoder_OMSBF.cpp:2975) */
    const int thr = imin(imax(max_vote, 1), (int)(int8_t)f->vote_cap);
    for (int u = tid; u < units; u += T) {
        const int w0 = (u >> 2) * 8 + 2 * (u & 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t m = 0xffffffffu; /* thr <= 0: every vote count qualifies */
            if (thr >= 16) m = 0u;
            else if (thr > 0) m = votes_ge(sCnt[3 * nw + w0 + h], sCnt[2 * nw + w0 + h], sCnt[nw + w0 + h], sCnt[w0 + h], thr);
            sHard[w0 + h] ^= m;
        }
    }
    LF_WG_SYNC();
}

#endif /* LNSFAID_PHASES_H */
