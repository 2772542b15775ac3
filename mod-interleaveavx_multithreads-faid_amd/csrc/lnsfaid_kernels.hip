/*
 * lnsfaid_kernels.hip — the two-rows-per-lane decode kernel and the error-counter kernel (CDNA4, gfx950).
 * Since round 2 the default decode kernel is lnsfaid_kernel4.hip (four rows per lane, byte-parallel); this one runs DecodeMethod 0
 * (NMS) and FAID tables that differ between weight classes or are not monotone.  Syndrome and bit flipping
 * are shared with it through lnsfaid_phases.h.
 *
 * Mapping.  The reference interleaves 32 codewords in the int8 lanes of one AVX register
 * and walks the 3072 check rows serially (CDecoder_FAID.cpp:631-1527).  Here one 128-thread workgroup
 * (two wavefronts) owns ONE codeword and thread i owns check rows i and i+128 of every layer (block row) of
 * the quasi-cyclic H.  The 256 rows of a layer touch disjoint variable nodes (no block row repeats a block
 * column), so a layer is one parallel step and the reference's row-serial, in-place schedule is reproduced
 * exactly by running the layers in order with a workgroup barrier in between.
 *   - a-posteriori LLRs En (int8, 17664 B) live in LDS for the whole launch.  Through a circulant with
 *     shift s, lane i touches byte (s+i) mod 256 of its block column; its second row touches that byte ^ 128;
 *   - the two rows of a thread are processed as the two 16-bit halves of packed VALU operations
 *     (v_pk_sub_i16 / v_pk_min/max / v_perm_b32 as an 8-entry FAID look-up table), which halves the vector
 *     instruction count per edge;
 *   - check-to-variable messages are never stored per edge (the reference keeps 70400 bytes per codeword,
 *     CLDPC.h:123).  A row's outgoing messages are +-c1 on the edge of the first minimum and +-c2 elsewhere:
 *     8 bytes per row {sign bits, argmin, c1, c2} reproduce every Lmn bit for bit; they
 *     stream through global memory one layer ahead of use (16 B per lane, coalesced);
 *   - syndromes are bit-parallel: hard decisions are kept as a bit plane and the parity of 32 rows of a
 *     layer is the XOR of <= 24 rotated 32-bit windows of it, one lane per (layer, 32 rows) walking a host-built
 *     table of plane addresses (lnsfaid_phases.h: no cross-lane reduction);
 *   - the bit-flipping stage is bit-sliced: votes of 64 variable nodes are three rotated windows of the
 *     parity plane, added and compared with the threshold by boolean word operations.
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
#include "lnsfaid_phases.h"

/* DecodeMethods whose layered loop is Decode_OMS's: 1 (alone), 3 (+ plain bit flipping), 4 (+ DTBF) */
#define LF_OMS(M) ((M) == 1 || (M) == 3 || (M) == 4)
/* DecodeMethod 0 (CLDPC::Decode, normalised min-sum) shares the un-clamped V2C and the plain sign with the OMS loop */
#define LF_MINSUM(M) ((M) == 0 || LF_OMS(M))

#define SAT_POS_VAR 31 /* Constants_SSE.h:22 */
#define SAT_NEG_VAR (-31)
#define SAT_POS_MSG 7  /* Constants_SSE.h:24 */


typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));
union Pk { uint32_t u; s2 s; u2 us; };
__device__ __forceinline__ s2 S(uint32_t x) { Pk v; v.u = x; return v.s; }
__device__ __forceinline__ u2 US(uint32_t x) { Pk v; v.u = x; return v.us; }
__device__ __forceinline__ uint32_t U(s2 x) { Pk v; v.s = x; return v.u; }
__device__ __forceinline__ uint32_t U(u2 x) { Pk v; v.us = x; return v.u; }
__device__ __forceinline__ s2 pk_max(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 pk_min(s2 a, s2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ u2 pk_maxu(u2 a, u2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ u2 pk_minu(u2 a, u2 b) { return __builtin_elementwise_min(a, b); }

/* a * b + c on both 16-bit halves in ONE instruction (the compiler turns the 0/1 multiply into two
 * compare + select pairs otherwise) */
__device__ __forceinline__ u2 pk_mad(u2 a, u2 b, u2 c)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(U(a)), "v"(U(b)), "v"(U(c)));
    return US(r);
}

/* 1 where the 16-bit half is non-zero, else 0 (the compiler expands min(x, 1) into compares and selects) */
__device__ __forceinline__ u2 pk_nonzero(uint32_t x)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(x));
    return US(r);
}
/* signed variant, and the two constant forms used for sign multipliers: q = 2*b - 1 and q = 1 - 2*b (b in {0,1}) */
__device__ __forceinline__ s2 pk_mad_i(s2 a, s2 b, s2 c)
{
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(U(a)), "v"(U(b)), "v"(U(c)));
    return S(r);
}
__device__ __forceinline__ s2 pk_2b_minus_1(uint32_t b)
{
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, 2, -1 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(b));
    return S(r);
}
__device__ __forceinline__ s2 pk_1_minus_2b(uint32_t b)
{
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, -2, 1 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(b));
    return S(r);
}

/* En lives at LDS byte offset 0 of the workgroup (the decode kernel has no static LDS; checked at kernel entry), so
 * a byte offset IS the ds_read / ds_write address: no base add per access. */
typedef __attribute__((address_space(3))) int8_t lds_i8;
__device__ __forceinline__ int en_ld(uint32_t off) { return *(const lds_i8*)(size_t)off; }
__device__ __forceinline__ void en_st(uint32_t off, int v) { *(lds_i8*)(size_t)off = (int8_t)v; }
/* LDS byte offset of row A's variable node on a circulant: ((tid + shift) mod 256) inside the block column, in
 * two instructions (the compiler otherwise re-derives the row B address from scratch) */
__device__ __forceinline__ uint32_t vn_offset(uint32_t tid, uint32_t sb, uint32_t vff)
{
    uint32_t ad;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(tid + sb), "v"(vff), "s"(sb & ~255u));
    return ad;
}

/* (a & mask) | (b & ~mask) */
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}

/* XOR over lanes 0..31 of a wave half, result in lane 31 (DPP row shifts + row broadcast, no LDS) */
__device__ __forceinline__ uint32_t xor_reduce32(uint32_t v)
{
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); /* row_shr:8 */
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1, 3 */
    return v;
}

/* ---- bit plane from En: hard decision En > 0 (CDecoder_FAID.cpp:299, :6416-6419), or with CONF the 2B1C
 * confidence bit |En| >= thr (CDecoder_FAID_2B1C.cpp:6132-6136) ----------------------------------------- */
/* Four variable nodes per lane: one ds_read_b32, a byte-parallel sign test, the four flags gathered into a nibble
 * (v_dot4 with weights 1, 2, 4, 8) and the nibbles of eight neighbouring lanes OR-ed into one plane word with three DPP
 * row shifts.  N must be a multiple of 4; thr <= 31 (|En| <= 31). */
template <bool CONF>
__device__ void build_plane(CCode c, const int8_t* sEn, uint32_t* plane, int thr, int tid)
{
    typedef const __attribute__((address_space(3))) uint32_t lds_u32;
    const int N = c->n_var;
    const uint32_t lane = (uint32_t)tid & 63u, sh = 4u * (lane & 7u);
    const int th = thr < 1 ? 1 : (thr > 32 ? 32 : thr); /* |En| <= 31: every threshold above 31 means "never" */
    const uint32_t thr4 = (uint32_t)th * 0x01010101u, thm4 = (uint32_t)(th - 1) * 0x01010101u;
    constexpr int U = 4; /* LDS reads in flight per thread */
    for (int base = 0; base < N; base += U * 4 * LF_T) {
        uint32_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = base + u * 4 * LF_T + 4 * tid;
            x[u] = v < N ? *(lds_u32*)(size_t)(uint32_t)v : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = base + u * 4 * LF_T + 4 * tid;
            uint32_t p;
            if (CONF) { /* |En| >= thr on the biased bytes En + 128: no carries between bytes for |En|, thr <= 31 */
                const uint32_t y = x[u] ^ 0x80808080u;
                p = thr < 1 ? 0x80808080u : ((y - thr4) | ~(y + thm4)) & 0x80808080u;
            } else {    /* En > 0: sign clear and low seven bits not all zero */
                p = ((x[u] & 0x7f7f7f7fu) + 0x7f7f7f7fu) & ~x[u] & 0x80808080u;
            }
            uint32_t w = __builtin_amdgcn_udot4(p >> 7, 0x08040201u, 0u, false) << sh;
            w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x111, 0xf, 0xf, true); /* row_shr:1 */
            w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x112, 0xf, 0xf, true); /* row_shr:2 */
            w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x114, 0xf, 0xf, true); /* row_shr:4 */
            if ((lane & 7u) == 7u && v < N) plane[v >> 5] = w; /* v = 4 * (lane of the word's last nibble): v >> 5 is the word */
        }
    }
    __syncthreads();
}

/* ---- cheap "certainly dirty" test for decoders that only need unsat != 0 (DecodeMethod 2: no EF tables, no
 * selective offset): parity of this thread's two rows of layer 0 straight from En.  The hard decision En > 0
 * is the sign bit of -En, and the XOR of the sign bits is the sign bit of the XOR.  A non-zero parity anywhere
 * proves unsat > 0 (the group cannot stop here and the codeword cannot park), so the bit plane and the full
 * syndrome are only built when layer 0 is clean.  Returns a workgroup-uniform flag. */
__device__ bool layer0_dirty(CCode c, const int8_t* sEn, int tid, uint32_t vff, int* sRed)
{
    /* Half of the layer's rows (those of wave 0: 0..63 and 128..191) prove "dirty" just as well in all but a vanishing
     * share of the cases, where the full syndrome decides as it would have anyway; the other wave only joins the barrier. */
    int dirty = 0;
    if (tid < 64) {
        const int deg = c->deg[0];
        int accA = 0, accB = 0;
#pragma unroll
        for (int j = 0; j < LF_MAX_DEG; ++j) {
            if (j < deg) {
                const uint32_t sb = c->circ[0][j].sb;
                const uint32_t ad = vn_offset((uint32_t)tid, sb, vff);
                accA ^= -en_ld(ad);
                accB ^= -en_ld(ad ^ 128u);
            }
        }
        dirty = __ballot((accA | accB) < 0) != 0ull ? 1 : 0;
    }
    return block_sum<LF_T>(dirty, tid, sRed) != 0;
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

#ifndef LF_WAVES_PER_SIMD
#define LF_WAVES_PER_SIMD 4 /* 128 VGPRs per wave; LDS allows 8 workgroups = 16 waves per CU */
#endif
#define JJ(j) ((uint32_t)(j) | ((uint32_t)(j) << 16))
/* edge j inside the min-search key: LF_JCODE_A / LF_JCODE_B per half (bit index of the edge's sign, lane of sbtab) */
#define JC(j) ((uint32_t)LF_JCODE_A(j) | ((uint32_t)LF_JCODE_B(j) << 16))

/* ---- one layered iteration (FAID / 2B1C: CDecoder_FAID.cpp:631-1527; OMS: CDecoder_OMS.cpp:334-743; NMS:
 * CLDPC.cpp:287-2283) -----------------------------------------------------------------------------------------
 * rows: this codeword's compressed messages, [nbr][128] uint4 for the row pair (tid, tid+128); 16-bit halves:
 *   .x bit j (j < 16), .y bit j-16: Lmn on edge j is negative (= s_j ^ F: s_j the sign of the V2C, F = XOR_all(s) ^
 *      (deg odd)); 0 for a zero message (FAID / 2B1C); low half row A, high half row B
 *   .z per half: c1 << 5 | c2 << 8 | n << 11, n = the argmin edge's message is negative
 *   .w per half: LDS offset of the argmin edge's variable node
 *   Lmn(edge j) = (j == argmin ? c1 : c2) with that sign
 * DecodeMethod 0 keeps a by-value mask instead (its two constants use different factors):
 *   .w bit j / .y bit 8 + j - 16: |t_j| == min1, no argmin fields.
 *
 * Only ONE edge per row carries c1.  Both passes therefore treat every edge as if it carried c2 and the argmin edge is
 * patched through LDS instead of selecting per edge:
 *   before pass 1   En[argmin_old] += q (c1 - c2): (En + q (c1 - c2)) + q c2 = En + q c1, the exact V2C
 *   after pass 1    the new argmin edge's V2C is recomputed from En (still the old value), its exact new En replaces
 *                   the as-if value pass 2 writes (same thread, LDS operations of a wave execute in order)
 * No other row of the layer touches these variable nodes, so the detour is invisible outside this function.
 *
 * DEG > 0: compile-time degree (no per-edge branches, so the scheduler can issue all table loads and LDS reads of
 * the row up front and interleave the edges); DEG == 0: run-time degree `deg` with a guard per edge. */
template <int METHOD, bool UNIW, int DEG>
__device__ __forceinline__ uint4 layer_step(CCode c, CCfg f, int8_t* sEn, int tid, int br, int deg, int itx, bool window,
                                            bool lme, int f1, int f2, uint4 cur, bool prA, bool prB, uint32_t vff,
                                            uint32_t sbtab, bool fresh, const uint4* nxt_rows, uint4& nxt,
                                            const uint32_t* nxt_tab, uint32_t& tab)
{
    constexpr int NJ = DEG > 0 ? DEG : LF_MAX_DEG;
    /* DecodeMethod 0 with Factor_1 == Factor_2 (one normalisation factor, the usual NMS): a tie between the two minima gives
     * cste_1 == cste_2 like everywhere else, so the row carries c1 on one edge only and takes the patch path; with two
     * different factors (NMSV) every edge with |t| == min1 carries c1 and the by-value mask is kept per edge. */
    constexpr bool NMSV = (METHOD == 0 && !UNIW);
    constexpr bool PATCH = !NMSV;
    uint32_t llo = f->lut_lo[itx][0], lhi = f->lut_hi[itx][0];
    uint32_t elo = f->lut_ef_lo[itx][0], ehi = f->lut_ef_hi[itx][0];
    /* UNIW: one non-decreasing table for every edge of the row.  min(LUT[a]) = LUT[min a] and likewise for the second
     * minimum, so the search runs on |t| and the table (and the clamp to 7 in front of it) is applied to the two
     * results; which of several tied edges is called the argmin does not matter. */
    constexpr bool LATE_LUT = !LF_MINSUM(METHOD) && UNIW;
    uint32_t selk = 0;
    if (!LF_MINSUM(METHOD) && !UNIW) asm volatile("v_mov_b32 %0, 0x06020400" : "=v"(selk)); /* key = {m.b2, code.b2, m.b0, code.b0} */
    const uint32_t XL = cur.x, XH = cur.y; /* bit j: Lmn on edge j is negative */
    const u2 C1o = US((cur.z >> 5) & 0x00070007u), C2o = US((cur.z >> 8) & 0x00070007u);
    uint32_t c64;
    asm volatile("v_mov_b32 %0, 0x400040" : "=v"(c64)); /* packed 64 kept in a VGPR for v_pk_mad_i16 */
    uint32_t efmask = 0; /* mask_eef per row, CDecoder_FAID.cpp:713-720 */
    if (METHOD == 5 && window && lme) efmask = (prA ? 0x0000ffffu : 0u) | (prB ? 0xffff0000u : 0u);

    if (PATCH && !fresh) {
        const uint32_t pa = cur.w & 0xffffu, pb = cur.w >> 16;
        const int eA = en_ld(pa), eB = en_ld(pb);
        const s2 E = S(__builtin_amdgcn_perm((uint32_t)eB, (uint32_t)eA, 0x05040100u));
        const s2 Ep = pk_mad_i(pk_2b_minus_1((cur.z >> 11) & 0x00010001u), S(U(C1o - C2o)), E);
        en_st(pa, Ep.x);
        en_st(pb, Ep.y);
    }

    const s2 nC2o = (s2)(0) - S(U(C2o));
    uint32_t y[NJ];
    uint32_t adr[NJ];
    uint32_t sx = 0;
    /* NMS starts its minima from 31 like the reference's registers (CLDPC.cpp:296); elsewhere any value above the keys */
    u2 k1 = US(METHOD == 0 ? 0x1fff1fffu : 0x7fff7fffu), k2 = k1;

#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (DEG > 0 || j < deg) {
            /* LDS byte address of row A's variable node: (tid + shift) mod 256 inside the block column */
            const uint32_t ad = vn_offset((uint32_t)tid, c->circ[br][j].sb, vff);
            adr[j] = ad;
            const int eA = en_ld(ad), eB = en_ld(ad ^ 128u);
            const s2 E = S(__builtin_amdgcn_perm((uint32_t)eB, (uint32_t)eA, 0x05040100u));
            u2 mag = C2o;
            if (NMSV) { /* stored per edge: was |t| == min1 */
                const uint32_t imb = ((j < 16 ? cur.w : cur.y) >> (j < 16 ? j : j - 16 + 8)) & 0x00010001u;
                mag = pk_mad(US(imb), C1o - C2o, C2o);
            }
            /* Lmn = neg ? -mag : mag with neg = bit j of the sign words, so En - Lmn = En + qn * (-mag), qn = 1 - 2 * neg */
            const uint32_t nb = ((j < 16 ? XL : XH) >> (j & 15)) & 0x00010001u;
            const s2 qn = pk_1_minus_2b(nb);
            s2 t = pk_max(pk_mad_i(qn, NMSV ? (s2)(0) - S(U(mag)) : nC2o, E), (s2)(SAT_NEG_VAR)); /* VECTOR_SUB_AND_SATURATE_VAR_8bits */
            s2 yy;
            if (LF_MINSUM(METHOD)) {
                yy = pk_mad_i(t, S(c64), S(0x00200020u)); /* 64 t + 32: sign(yy) = (t < 0), CDecoder_OMS.cpp:372 */
            } else {
                t = pk_min(t, (s2)(SAT_POS_VAR)); /* CDecoder_FAID.cpp:672 */
                /* 64 t + qn.  The reference takes the sign of a zero V2C from En (back-track, CDecoder_FAID.cpp:682); t = 0
                 * means En = Lmn, so that sign is Lmn < 0 = neg (sign bits of zero messages are stored as 0, see the end
                 * of this function), i.e. the sign of qn.  Reading it from the LDS value would be wrong on the argmin edge,
                 * whose En is the patched one. */
                yy = pk_mad_i(t, S(c64), qn);
            }
            y[j] = U(yy);
            /* XOR of the signs, two edges per instruction (v_bitop3_b32, truth table 0x96 = a ^ b ^ c) */
            if (DEG > 0) { if (j & 1) sx = __builtin_amdgcn_bitop3_b32(sx, y[j - 1], U(yy), 0x96); else if (j == DEG - 1) sx ^= U(yy); }
            else sx ^= U(yy);
            s2 a = pk_max(t, (s2)(0) - t);
            u2 key;
            if (LF_MINSUM(METHOD) || LATE_LUT) {
                /* |t| << 8 | code in one instruction, the code from an SGPR (VOP3 takes no literal); |t| is clamped to 7
                 * (and mapped) after the search, CDecoder_OMS.cpp:374 */
                uint32_t kk;
                asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(kk) : "v"(U(a)), "s"(JC(j)));
                key = US(kk);
            } else {
                a = pk_min(a, (s2)(SAT_POS_MSG)); /* |t| >= 8 maps through column 7 */
                if (!UNIW) {
                    const uint32_t wc = c->circ[br][j].wclass;
                    llo = f->lut_lo[itx][wc]; lhi = f->lut_hi[itx][wc];
                    if (METHOD == 5) { elo = f->lut_ef_lo[itx][wc]; ehi = f->lut_ef_hi[itx][wc]; }
                }
                /* bytes 1 and 3 of the selector are 0 and pick table entry 0: dropped again when the key is assembled */
                uint32_t m = __builtin_amdgcn_perm(lhi, llo, U(a));
                if (METHOD == 5) {
                    const uint32_t me = __builtin_amdgcn_perm(ehi, elo, U(a));
                    m = (m & ~efmask) | (me & efmask);
                }
                key = US(__builtin_amdgcn_perm(m, JC(j), selk));
            }
            k2 = pk_minu(k2, pk_maxu(k1, key)); /* VECTOR_MIN_2 with the old min1 */
            k1 = pk_minu(k1, key);
        }
    }

    /* the next layer's messages and argmin table are fetched from here on: after pass 1, where the register pressure
     * peaks, and still most of a layer ahead of their use */
    nxt = *nxt_rows;
    tab = *nxt_tab;

    u2 min1 = k1 >> (u2)(8), min2 = k2 >> (u2)(8);
    if (METHOD != 0 && (LF_MINSUM(METHOD) || LATE_LUT)) {
        min1 = pk_minu(min1, (u2)(SAT_POS_MSG));
        min2 = pk_minu(min2, (u2)(SAT_POS_MSG));
    }
    if (LATE_LUT) {
        uint32_t m1 = __builtin_amdgcn_perm(lhi, llo, U(min1) | 0x0c000c00u), m2 = __builtin_amdgcn_perm(lhi, llo, U(min2) | 0x0c000c00u);
        if (METHOD == 5) {
            m1 = (m1 & ~efmask) | (__builtin_amdgcn_perm(ehi, elo, U(min1) | 0x0c000c00u) & efmask);
            m2 = (m2 & ~efmask) | (__builtin_amdgcn_perm(ehi, elo, U(min2) | 0x0c000c00u) & efmask);
        }
        min1 = US(m1);
        min2 = US(m2);
    }
    u2 C1n, C2n;
    if (METHOD == 0) {
        /* cste_2 = min(((min1 * Factor_1) & 0xffff) >> 5, 7), cste_1 likewise from min2 and Factor_2 (CLDPC.cpp:337-352;
         * the signed-saturating pack never triggers below the limit of 7) */
        const uint32_t g1 = (uint32_t)(uint16_t)(int16_t)f1, g2 = (uint32_t)(uint16_t)(int16_t)f2;
        const uint32_t a2 = imin((int)(((min1.x * g1) & 0xffffu) >> 5), SAT_POS_MSG), b2 = imin((int)(((min1.y * g1) & 0xffffu) >> 5), SAT_POS_MSG);
        const uint32_t a1 = imin((int)(((min2.x * g2) & 0xffffu) >> 5), SAT_POS_MSG), b1 = imin((int)(((min2.y * g2) & 0xffffu) >> 5), SAT_POS_MSG);
        C1n = US(a1 | (b1 << 16));
        C2n = US(a2 | (b2 << 16));
    } else if (LF_OMS(METHOD)) {
        const bool FA = prA && lme, FB = prB && lme;
        const int a1 = imin(oms_offset(min2.x, window, FA, f1, f2), SAT_POS_MSG); /* cste_1, CDecoder_OMS.cpp:431 */
        const int a2 = imin(oms_offset(min1.x, window, FA, f1, f2), SAT_POS_MSG); /* cste_2 */
        const int b1 = imin(oms_offset(min2.y, window, FB, f1, f2), SAT_POS_MSG);
        const int b2 = imin(oms_offset(min1.y, window, FB, f1, f2), SAT_POS_MSG);
        C1n = US((uint32_t)(a1 & 0xffff) | ((uint32_t)b1 << 16));
        C2n = US((uint32_t)(a2 & 0xffff) | ((uint32_t)b2 << 16));
    } else {
        /* CDecoder_FAID.cpp:865-866, offset 0.  Table entries are validated to lie in 0..7 (lnsfaid_set_cfg), so the
         * values mapped after the search need no further clamp. */
        C1n = LATE_LUT ? min2 : pk_minu(min2, (u2)(SAT_POS_MSG));
        C2n = LATE_LUT ? min1 : pk_minu(min1, (u2)(SAT_POS_MSG));
    }
    /* sign of the new message on edge j: XOR of all signs ^ (deg odd) ^ own sign
     * (the 0xC0 / 0x40 constants of CDecoder_FAID.cpp:902-906 fed to _mm256_sign_epi8) */
    const uint32_t Fn = U(S(sx) >> (s2)(15)) ^ (((DEG > 0 ? DEG : deg) & 1) ? 0xffffffffu : 0u);
    const uint32_t Fn01 = Fn & 0x00010001u;

    /* ---- the new argmin edge, exactly (its En is still the old value: pass 2 has not started) ---- */
    uint32_t pa = 0, pb = 0, nq = 0;
    s2 en_arg = (s2)(0);
    uint32_t ca = U(k1) & 0xffu, cb = (U(k1) >> 16) & 0xffu; /* LF_JCODE_A / _B of the argmin edges */
    if (METHOD == 0) { /* minima start from 31: with every |t| above it no edge is the argmin, c1 == c2, any edge will do */
        ca = ca == 0xffu ? (uint32_t)LF_JCODE_A(0) : ca;
        cb = cb == 0xffu ? (uint32_t)LF_JCODE_B(0) : cb;
    }
    if (PATCH) {
        const uint32_t sba = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ca << 2), (int)sbtab);
        const uint32_t sbb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(cb << 2), (int)sbtab);
        pa = bfi(vff, (uint32_t)tid + sba, sba);          /* as vn_offset, operands in VGPRs */
        pb = bfi(vff, (uint32_t)tid + sbb, sbb) ^ 128u;
        const int eA = en_ld(pa), eB = en_ld(pb);
        const s2 E = S(__builtin_amdgcn_perm((uint32_t)eB, (uint32_t)eA, 0x05040100u));
        const unsigned long long X = ((unsigned long long)XH << 32) | XL;  /* bit LF_JCODE: old message negative */
        const uint32_t nb = ((uint32_t)(X >> ca) & 1u) | (((uint32_t)(X >> cb) & 1u) << 16);
        const s2 qn = pk_1_minus_2b(nb);
        s2 t = pk_max(pk_mad_i(qn, nC2o, E), (s2)(SAT_NEG_VAR));
        uint32_t sj;
        if (LF_MINSUM(METHOD)) {
            sj = U(US(U(t)) >> (u2)(15));
        } else {
            t = pk_min(t, (s2)(SAT_POS_VAR));
            sj = U(US(U(pk_mad_i(t, S(c64), qn))) >> (u2)(15));
        }
        nq = sj ^ Fn01;
        en_arg = pk_min(pk_max(pk_mad_i(pk_1_minus_2b(nq), S(U(C1n)), t), (s2)(SAT_NEG_VAR)), (s2)(SAT_POS_VAR));
    }

    uint32_t nXL = 0, nXH = 0, nIL = 0, nIH = 0;
    uint32_t one2;
    asm volatile("s_mov_b32 %0, 0x10001" : "=s"(one2));
    uint32_t KA = 0, KB = 0; /* 64 * Lmn (+ 32 where pass 1 left the rounding to pass 2) for s_j = 1 / s_j = 0 */
    if (!NMSV) {
        const s2 kp = pk_mad_i(S(U(C2n)), S(c64), LF_MINSUM(METHOD) ? (s2)(0) : (s2)(32));
        const s2 kn = pk_mad_i(S(U(C2n)), (s2)(0) - S(c64), LF_MINSUM(METHOD) ? (s2)(0) : (s2)(32));
        KA = __builtin_amdgcn_bitop3_b32(Fn, U(kp), U(kn), 0xca); /* s_j = 1: negative unless F */
        KB = __builtin_amdgcn_bitop3_b32(Fn, U(kn), U(kp), 0xca);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (DEG > 0 || j < deg) {
            const s2 yy = S(y[j]);
            const uint32_t sm = U(yy >> (s2)(15)); /* raw sign s_j per half as a mask: 0 / 0xffff */
            s2 en;
            if (NMSV) { /* by value: every edge with |t| == min1 takes cste_1 (CLDPC.cpp:371-375) */
                const s2 t = yy >> (s2)(6);
                const u2 ne = pk_nonzero(U(pk_max(t, (s2)(0) - t)) ^ U(min1));
                const uint32_t im = U(ne) ^ 0x00010001u;
                if (j < 16) nIL |= im << j; else nIH |= im << (j - 16);
                const u2 mag = pk_mad(ne, C2n - C1n, C1n);
                /* new Lmn = (s_j ^ F) ? -mag : mag, so t + Lmn = t + q * mag with q = (s_j ^ F) ? -1 : 1 = (mask ^ F) | 1 */
                const s2 q = S(__builtin_amdgcn_bitop3_b32(sm, Fn, one2, 0xbe)); /* truth table of (a ^ b) | c */
                en = pk_mad_i(q, S(U(mag)), t);
            } else {
                /* y = 64 t + r with 0 < r + 32 < 64 (FAID) or r = 32 (min-sum): (y + 64 Lmn [+ 32]) >> 6 = t + Lmn, and every
                 * edge carries +-c2 (the argmin edge is patched below), so the signed, scaled, rounded addend is one of two
                 * row constants chosen by the edge's own sign */
                en = (yy + S(__builtin_amdgcn_bitop3_b32(sm, KA, KB, 0xca))) >> (s2)(6); /* truth table of a ? b : c */
            }
            en = pk_min(pk_max(en, (s2)(SAT_NEG_VAR)), (s2)(SAT_POS_VAR)); /* :919-920 */
            if (j < 16) asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(nXL) : "v"(sm), "s"(0x00010001u << (j & 15)));
            else asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(nXH) : "v"(sm), "s"(0x00010001u << (j & 15)));
            uint32_t ad = adr[j];
            asm("" : "+v"(ad)); /* row B's address is recomputed here rather than kept in a register since pass 1 */
            en_st(ad, en.x);
            en_st(ad ^ 128u, en.y);
        }
    }
    if (NMSV) return make_uint4(nXL ^ Fn, ((nXH ^ Fn) & 0x00ff00ffu) | (nIH << 8), (U(C1n) << 5) | (U(C2n) << 8), nIL);
    en_st(pa, en_arg.x);
    en_st(pb, en_arg.y);
    /* the new Lmn on edge j is negative iff s_j ^ F */
    if (!LF_MINSUM(METHOD)) {
        /* A zero message has no sign: store "not negative" for it, so that the back-track of the next iteration (pass 1)
         * can read "Lmn < 0" straight from the sign bit.  c2 = 0 zeroes every message of the row but the argmin's. */
        const uint32_t N2 = U((s2)(0) - S(U(pk_nonzero(U(C2n))))), Z = ~N2; /* halves with c2 != 0 / == 0 as 0xffff masks */
        const uint32_t N1 = U((s2)(0) - S(U(pk_nonzero(U(C1n)))));          /* halves with c1 != 0 */
        const unsigned long long oh = (1ull << ca) | (1ull << cb);
        const uint32_t clrL = Z & ~((uint32_t)oh & N1), clrH = Z & ~((uint32_t)(oh >> 32) & N1);
        nXL = __builtin_amdgcn_bitop3_b32(nXL, Fn, clrL, 0x14); /* (a ^ b) & ~c */
        nXH = __builtin_amdgcn_bitop3_b32(nXH, Fn, clrH, 0x14);
    } else {
        nXL ^= Fn;
        nXH ^= Fn;
    }
    return make_uint4(nXL, nXH, (U(C1n) << 5) | (U(C2n) << 8) | (nq << 11), pa | (pb << 16));
}

template <int METHOD, bool UNIW>
__device__ void main_step(CCode c, CCfg f, const LfDevCode* gc, int8_t* sEn, uint4* __restrict__ rows, int tid, int it,
                          uint32_t pA, uint32_t pB, bool lme)
{
    const bool fresh = (it == 1); /* no iteration has run yet: every Lmn is still 0, nothing in HBM */
    const int rem = f->max_iter - it; /* nombre_iterations inside the loop body */
    const int itx = (it >= 1 && it <= 5) ? it - 1 : 5; /* switch at CDecoder_FAID.cpp:760-779 */
    const bool window = rem <= f->floor_iter_thresh;
    const int f1 = f->factor_1, f2 = f->factor_2;
    const int nbr = c->nbr;
    uint32_t vff;
    asm volatile("v_mov_b32 %0, 0xff" : "=v"(vff)); /* bit-field-insert mask kept in a VGPR (constant bus) */

    uint4 cur = make_uint4(0u, 0u, 0u, 0u); /* Lmn = 0 before the first iteration (CDecoder_FAID.cpp:211-214) */
    if (!fresh) cur = rows[tid];
    uint32_t tab = gc->sbtab[0][tid & 63]; /* lane-indexed: vector load, one layer ahead like the messages */
    for (int br = 0; br < nbr; ++br) {
        /* always a valid address (the last layer re-reads layer 0, the first iteration reads what it is about to
         * overwrite): the loads stay unconditional inside the layer's single basic block */
        const int brn = br + 1 < nbr ? br + 1 : 0;
        uint4 nxt;
        const uint4* nxt_rows = rows + brn * LF_T + tid;
        const uint32_t* nxt_tab = &gc->sbtab[brn][tid & 63];
        const uint32_t tab_cur = tab;
        const int deg = c->deg[br];
        const bool prA = (pA >> br) & 1u, prB = (pB >> br) & 1u;
        uint4 st;
        if (deg == 23) st = layer_step<METHOD, UNIW, 23>(c, f, sEn, tid, br, deg, itx, window, lme, f1, f2, cur, prA, prB, vff, tab_cur, fresh, nxt_rows, nxt, nxt_tab, tab);
        else if (deg == 22) st = layer_step<METHOD, UNIW, 22>(c, f, sEn, tid, br, deg, itx, window, lme, f1, f2, cur, prA, prB, vff, tab_cur, fresh, nxt_rows, nxt, nxt_tab, tab);
        else st = layer_step<METHOD, UNIW, 0>(c, f, sEn, tid, br, deg, itx, window, lme, f1, f2, cur, prA, prB, vff, tab_cur, fresh, nxt_rows, nxt, nxt_tab, tab);
        if (rem > 0) rows[br * LF_T + tid] = st; /* the last layered iteration's messages are never read again */
        __syncthreads(); /* the next layer reads what this one wrote */
        if (!fresh) cur = nxt;
    }
}

/* ---- the decode kernel: one workgroup per codeword ---------------------------------------------------- */
template <int METHOD, bool UNIW>
__global__ __launch_bounds__(LF_T, LF_WAVES_PER_SIMD) void lnsfaid_decode_kernel(LfKernelArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    CCode c = (CCode)a.code;
    CCfg f = (CCfg)a.cfg;
    const int tid = (int)threadIdx.x;
    const int cw = (int)blockIdx.x;
    const int N = c->n_var, M = c->n_check, K = c->k_info, nw = c->n_words, pw = c->p_words;
    int8_t* sEn = (int8_t*)smem;
    /* (en_ld / en_st address En by its LDS offset: the dynamic segment must start at 0, i.e. no static LDS in this kernel -
     * checked on the host when a context picks its kernel, lnsfaid_capi.hip kernel_check) */
    uint32_t* sHard0 = (uint32_t*)smem;      /* bit-flipping stage: hard_ch and hard2 overlay the dead En */
    uint32_t* sHard2 = (uint32_t*)smem + nw;
    uint32_t* sHard = (uint32_t*)(smem + lf_lds_off_hard(N));
    uint32_t* sP = (uint32_t*)(smem + lf_lds_off_p(N, nw));
    int* sStat = (int*)(smem + lf_lds_off_stat(N, nw, pw));
    int* sRed = sStat + LNSFAID_GROUP;

    const int max_iter = f->max_iter, max_bf = f->max_bf;
    const int t_bf0 = max_iter + 1;   /* first bit-flipping decision point */
    const int t_end = t_bf0 + max_bf; /* both loops exhausted               */

    const int my_status = a.status_cur ? a.status_cur[cw] : 0; /* null: first launch of a batch, every codeword fresh */
    if (my_status & LF_DONE) { /* uniform exit */
        if (tid == 0) a.status_next[cw] = my_status;
        return;
    }
    /* snapshot of the 32 lanes of this group */
    const int g = cw >> 5, lane_in_group = cw & 31;
    if (tid < LNSFAID_GROUP) sStat[tid] = a.status_cur ? a.status_cur[g * LNSFAID_GROUP + tid] : 0;
    if (tid == LNSFAID_GROUP) sRed[LF_ZERO_SLOT] = 0; /* the word unused synw slots point at */
    __syncthreads();
    int kmax = 0, all_same = 1;
    for (int l = 0; l < LNSFAID_GROUP; ++l) {
        const int s = sStat[l];
        kmax = imax(kmax, s & LF_PROG_MASK);
        all_same &= (s == my_status);
    }
    int prog = my_status & LF_PROG_MASK;

    int8_t* g_en = a.st_en + (size_t)cw * (size_t)N;
    uint4* g_rows = a.st_rows + (size_t)cw * (size_t)(c->nbr * LF_T);
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
        const uint32_t* src_i = (const uint32_t*)(gi + (size_t)lane_in_group * (size_t)K);
        const uint32_t* src_p = (const uint32_t*)(gi + (size_t)LNSFAID_GROUP * (size_t)K + (size_t)lane_in_group * (size_t)M);
        uint32_t* dst = (uint32_t*)sEn;
        for (int i = tid; i < (K >> 2); i += LF_T) dst[i] = src_i[i];
        for (int i = tid; i < (M >> 2); i += LF_T) dst[(K >> 2) + i] = src_p[i];
        __syncthreads();
        for (int i = tid; i < c->puncture_tail; i += LF_T) sEn[N - 1 - i] = 0;
        __syncthreads();
        prog = 1;
    } else if (!in_bf) {
        const uint32_t* src = (const uint32_t*)g_en;
        uint32_t* dst = (uint32_t*)sEn;
        for (int i = tid; i < (N >> 2); i += LF_T) dst[i] = src[i];
        __syncthreads();
    } else {
        for (int i = tid; i < nw; i += LF_T) { sHard[i] = g_bits[i]; sHard0[i] = g_bits[nw + i]; sHard2[i] = g_bits[2 * nw + i]; }
        ls = a.st_lane[cw];
        __syncthreads();
    }

    uint32_t vff0;
    asm volatile("v_mov_b32 %0, 0xff" : "=v"(vff0));
    /* ---- all 32 lanes parked clean at the same decision point: the group stops here ---- */
    const bool group_stop = (my_status != 0) && all_same;

    if (!group_stop) {
        for (;;) {
            if (prog >= t_end) break; /* loops exhausted (also OMS after max_iter iterations) */
            if (max_bf > 0 && prog >= t_bf0 && !in_bf) {
                /* the layered loop ran out: enter the bit-flipping stage (CDecoder_FAID.cpp:6411-6428) */
                uint32_t conf[LF_MAX_BC * 8 / LF_T]; /* this thread's share of the 2B1C confidence plane */
                if (METHOD == 5) {
                    build_plane<true>(c, sEn, sHard, f->hard2_thr, tid); /* staged where the hard plane will go */
#pragma unroll
                    for (int k = 0; k < LF_MAX_BC * 8 / LF_T; ++k) conf[k] = (tid + k * LF_T < nw) ? sHard[tid + k * LF_T] : 0u;
                    __syncthreads();
                }
                build_plane<false>(c, sEn, sHard, 0, tid);
                /* En is dead from here on: its bytes take hard_ch (= hard) and hard2 */
                for (int i = tid; i < nw; i += LF_T) sHard0[i] = sHard[i];
                if (METHOD == 5) {
#pragma unroll
                    for (int k = 0; k < LF_MAX_BC * 8 / LF_T; ++k) if (tid + k * LF_T < nw) sHard2[tid + k * LF_T] = conf[k];
                }
                ls.Th = (int8_t)f->W; ls.l0 = 0; ls.l1 = 0; ls.t = 1;
                in_bf = true;
                __syncthreads();
            }
            uint32_t pA = 0, pB = 0;
            if (METHOD == 0) { /* CLDPC::Decode has no syndrome stage and no early stop */
                main_step<METHOD, UNIW>(c, f, a.code, sEn, g_rows, tid, prog, 0u, 0u, false);
                prog++;
            } else if (!in_bf) {
                bool lme = false;
                /* l_checksum_ and the unsatisfied count are consumed only inside the error-floor window
                 * (nombre_iterations <= floor_iter_thresh: OMS selective offset CDecoder_OMS.cpp:388, 2B1C tables
                 * CDecoder_FAID.cpp:714) and never by DecodeMethod 2; elsewhere only unsat != 0 matters */
                const bool needs_checksums = (METHOD != 2) && (max_iter - prog <= f->floor_iter_thresh);
                if (needs_checksums || !layer0_dirty(c, sEn, tid, vff0, sRed)) {
                    build_plane<false>(c, sEn, sHard, 0, tid);
                    const int unsat = syndrome<LF_T, true>(c, a.code, sP, tid, pA, pB, sRed);
                    if (unsat == 0 && prog >= kmax) break; /* clean on the group's front: park */
                    if (LF_OMS(METHOD)) lme = imin(unsat, 255) < (int)(uint8_t)f->floor_err_count; /* CDecoder_OMS.cpp:328 */
                    else lme = imin(unsat, 127) < (int)(int8_t)f->floor_err_count;              /* CDecoder_FAID.cpp:619 */
                }
                main_step<METHOD, UNIW>(c, f, a.code, sEn, g_rows, tid, prog, pA, pB, lme);
                prog++;
            } else {
                const int unsat = syndrome<LF_T, false>(c, a.code, sP, tid, pA, pB, sRed);
                if (unsat == 0 && prog >= kmax) break;
                if (METHOD == 3) bf_step_plain<LF_T>(c, f, a.code, sHard, sHard2 + nw /* 4 count planes in the dead En */, sP, tid, sRed);
                else bf_step<LF_T, METHOD>(c, f, a.code, sHard, sHard0, sHard2, sP, tid, ls, sRed);
                prog++;
            }
        }
    }

    const bool finished = group_stop || prog >= t_end;
    if (finished) {
        /* decodedBits[l][v] = hard decision (CDecoder_FAID.cpp:7091-7102, CDecoder_OMS.cpp:2966-2967) */
        uint32_t* out32 = (uint32_t*)g_out;
        if (!in_bf) {
            for (int i = tid; i < (N >> 2); i += LF_T) {
                const uint32_t e4 = ((const uint32_t*)sEn)[i];
                uint32_t o = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) o |= (uint32_t)(((int8_t)(e4 >> (8 * b))) > 0) << (8 * b);
                out32[i] = o;
            }
        } else {
            for (int i = tid; i < (N >> 2); i += LF_T) {
                const uint32_t bits = (sHard[i >> 3] >> ((i & 7) * 4)) & 15u;
                out32[i] = (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21);
            }
        }
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
        /* park: state back to HBM, status = the decision point the codeword is clean at */
        if (!in_bf) {
            const uint32_t* src = (const uint32_t*)sEn;
            uint32_t* dst = (uint32_t*)g_en;
            for (int i = tid; i < (N >> 2); i += LF_T) dst[i] = src[i];
        } else {
            for (int i = tid; i < nw; i += LF_T) { g_bits[i] = sHard[i]; g_bits[nw + i] = sHard0[i]; g_bits[2 * nw + i] = sHard2[i]; }
            if (tid == 0) a.st_lane[cw] = ls;
        }
        if (tid == 0) { a.status_next[cw] = prog; atomicAdd(a.remaining, 1u); }
    }
}

/* ---- CalculateErrors (CLDPC.cpp:4842-4876): one workgroup per group of 32 frames, one frame per wave pass;
 * 16 B per lane loads when K and N allow; four global atomics per workgroup ----------------------------- */
__device__ __forceinline__ int nonzero_bytes(uint32_t x)
{
    return __popc((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void lnsfaid_count_errors_kernel(const int8_t* __restrict__ decoded,
                                                                   const int8_t* __restrict__ input_bits, int n_var,
                                                                   int k_info, unsigned long long* __restrict__ out)
{
    __shared__ unsigned int sAcc[3];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3) sAcc[tid] = 0u;
    __syncthreads();
    unsigned int frames_err = 0, bits_err = 0, lt3 = 0;
    for (int fr = wave; fr < LNSFAID_GROUP; fr += 4) {
        const size_t cw = (size_t)blockIdx.x * LNSFAID_GROUP + (size_t)fr;
        const int8_t* d = decoded + cw * (size_t)n_var;
        const int8_t* r = input_bits ? input_bits + cw * (size_t)k_info : nullptr;
        int cnt = 0;
        if (WIDE) {
            const uint4* d4 = (const uint4*)d;
            const uint4* r4 = (const uint4*)r;
            /* eight 16-byte loads per lane in flight before the first use (a plain loop waits for every load in turn) */
            const int n16 = k_info >> 4;
            for (int j0 = 0; j0 < n16; j0 += 8 * 64) {
                uint4 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * 64 + lane;
                    x[u] = j < n16 ? d4[j] : make_uint4(0u, 0u, 0u, 0u);
                    if (r && j < n16) { const uint4 y = r4[j]; x[u].x ^= y.x; x[u].y ^= y.y; x[u].z ^= y.z; x[u].w ^= y.w; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) cnt += nonzero_bytes(x[u].x) + nonzero_bytes(x[u].y) + nonzero_bytes(x[u].z) + nonzero_bytes(x[u].w);
            }
        } else {
            const uint32_t* d1 = (const uint32_t*)d;
            const uint32_t* r1 = (const uint32_t*)r;
            for (int j = lane; j < (k_info >> 2); j += 64) cnt += nonzero_bytes(d1[j] ^ (r ? r1[j] : 0u));
        }
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
        if (lane == 0 && cnt > 0) { frames_err += 1; bits_err += (unsigned)cnt; lt3 += (cnt < 3) ? 1u : 0u; }
    }
    if (lane == 0) { atomicAdd(&sAcc[0], frames_err); atomicAdd(&sAcc[1], bits_err); atomicAdd(&sAcc[2], lt3); }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&out[0], (unsigned long long)LNSFAID_GROUP);
        if (sAcc[0]) {
            atomicAdd(&out[1], (unsigned long long)sAcc[0]);
            atomicAdd(&out[2], (unsigned long long)sAcc[1]);
            if (sAcc[2]) atomicAdd(&out[3], (unsigned long long)sAcc[2]);
        }
    }
}

/* ---- launchers (called from lnsfaid_capi.hip) ---------------------------------------------------------- */
template <int METHOD>
static const void* func_method(bool uniw)
{
    return uniw ? (const void*)lnsfaid_decode_kernel<METHOD, true> : (const void*)lnsfaid_decode_kernel<METHOD, false>;
}

/* the instance a configuration runs on (for hipFuncGetAttributes / the occupancy query, and for the launch) */
extern "C" const void* lf_decode_func(int method, int uniform_w)
{
    switch (method) {
    case 0: return func_method<0>(uniform_w != 0); /* normalised min-sum; uniform_w: Factor_1 == Factor_2 */
    case 1: return func_method<1>(true); /* OMS has no look-up table */
    case 2: return func_method<2>(uniform_w != 0);
    case 3: return func_method<3>(true); /* OMS arithmetic + plain bit flipping */
    case 4: return func_method<4>(true); /* OMS arithmetic + DTBF */
    case 5: return func_method<5>(uniform_w != 0);
    default: return nullptr;
    }
}
extern "C" int lf_decode_threads(void) { return LF_T; }

extern "C" hipError_t lf_launch_decode(int method, int uniform_w, const LfKernelArgs* args, size_t lds_bytes,
                                       hipStream_t stream)
{
    const void* fn = lf_decode_func(method, uniform_w);
    if (!fn) return hipErrorInvalidValue;
    void* kargs[] = { (void*)args };
    return hipLaunchKernel(fn, dim3((unsigned)args->n_cw), dim3(LF_T), kargs, lds_bytes, stream);
}

extern "C" hipError_t lf_launch_count_errors(const int8_t* decoded, const int8_t* input_bits, int n_var, int k_info,
                                             size_t n_cw, unsigned long long* out, hipStream_t stream)
{
    const dim3 grid((unsigned)(n_cw / LNSFAID_GROUP)), block(256);
    const bool wide = (k_info % 16 == 0) && (n_var % 16 == 0) && ((uintptr_t)decoded % 16 == 0)
        && ((uintptr_t)input_bits % 16 == 0);
    if (wide) hipLaunchKernelGGL(lnsfaid_count_errors_kernel<true>, grid, block, 0, stream, decoded, input_bits, n_var, k_info, out);
    else hipLaunchKernelGGL(lnsfaid_count_errors_kernel<false>, grid, block, 0, stream, decoded, input_bits, n_var, k_info, out);
    return hipGetLastError();
}
