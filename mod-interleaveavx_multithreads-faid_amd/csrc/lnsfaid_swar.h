/*
 * lnsfaid_swar.h — one layer of the layered decoder with FOUR check rows per lane, byte-parallel in 32-bit registers.
 *
 * Mapping.  One wavefront owns one codeword; lane i owns check rows i, i+64, i+128, i+192 of every
 * layer (block row) of the quasi-cyclic H.  Byte k of every working register belongs to row i + 64 k.  Inside a block
 * column the a-posteriori LLRs are stored so that the four variable nodes these rows meet on one circulant share ONE
 * LDS dword: variable node v (0..255) of the column lives in dword v mod 64, byte v div 64.  Through a circulant with
 * shift s lane i reads dword (i + s) mod 64 and rotates it right by ((i + s) div 64) mod 4 bytes: one ds_read_b32 and
 * one v_alignbyte_b32 instead of four byte reads and three packs, and all 64 lanes hit 64 consecutive dwords.
 *
 * Arithmetic is carry-free SWAR on biased bytes (every intermediate stays inside its byte, so plain 32-bit adds never
 * carry across rows) built from the instructions gfx950 issues at the full rate (two-source 32-bit ALU operations and
 * v_bitop3_b32) plus v_perm_b32 used three ways: as an 8-entry byte table (FAID look-up table, thermometer code, +-c
 * constants), as a "bit 7 of every byte -> byte mask" expander (its sign-replicating selectors 8..11), and as a packer.
 *   LDS byte        Eb = En + 120                      (En in [-31, 31])
 *   V2C             tb = t + 128,  t = En - Lold       (one add of a per-row, per-sign constant picked by v_perm)
 *   sign            bit 7 of ts = tb - b, b = "old message negative": the reference's back-tracked sign of a zero V2C
 *                   (CDecoder_FAID.cpp:682: t == 0 means En == Lold, so sign(En) is the stored sign bit)
 *   minima          thermometer code U(a), a = min(|t|, 7): bit i set iff a > i, so min = AND and
 *                   second-min' = second-min & (min | U): two boolean operations per edge for four rows; the binary
 *                   index of an edge attaining the minimum comes from five more AND accumulators (index bits 0..2: over
 *                   the edges with the bit clear, bits 3 and 4: over the fewer edges with it set) compared with the minimum
 *                   afterwards — no per-edge compare / select.
 *   update          En' = clamp(z, -31, 31 - c) + (L < 0 ? 0 : c) with z = t - (L < 0 ? c : 0): limits independent of the sign
 *                   of the new message L = +-c (sw_update)
 * Reference statements restated here: CDecoder_FAID.cpp:662-929 (FAID / 2B1C rows), CDecoder_OMS.cpp:363-471 (OMS rows),
 * CLDPC.cpp:296-375 (NMS rows).  As in the 2-rows-per-lane kernel the check-to-variable messages are kept compressed
 * (sign bit per edge, the two magnitudes of the row, the edge that carries the larger one) and only ONE edge per row
 * carries c1: every edge is processed as if it carried c2 and the arg-min edge is patched through LDS before pass 1 and
 * recomputed exactly after pass 2 (DESIGN.md 3.2 explains why this is bit-exact).
 *
 * The file compiles for the device (hipcc) and for the host (g++, -DSW_HOST): tests/ run the very same statements on the
 * CPU against the oracle, lane by lane (lanes of a layer are independent: they touch disjoint variable nodes).
 */
#ifndef LNSFAID_SWAR_H
#define LNSFAID_SWAR_H

#include <stdint.h>

#if defined(__HIPCC__) /* both passes of hipcc see the device flavour; a plain C++ compiler gets the restatements */
#define SW_DEV 1
#define SW_FN __device__ __forceinline__
#define SW_HD __host__ __device__ __forceinline__ /* table builders: also run by the host side of the C ABI */
#else
#define SW_DEV 0
#define SW_FN static inline
#define SW_HD static inline
#endif

#define SW_MAX_DEG 24
#ifndef SW_ILP
#define SW_ILP 4 /* edges whose dependency chains are interleaved in the two passes */
#endif

/* Keeps the compiler's scheduler from moving instructions across this point.  With two waves per SIMD nothing hides an LDS
 * round trip, so the layer step issues all loads of a phase back to back and only then starts consuming them; left alone the
 * scheduler interleaves "load, wait, use" per edge (one read in flight, every LDS latency exposed). */
#if SW_DEV
#define SW_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SW_SCHED_FENCE() ((void)0)
#endif

/* ---- the three non-trivial instructions, with host restatements of their ISA semantics -------------------------- */
/* v_perm_b32: byte i of the result = BYTE_PERMUTE({s0, s1}, sel byte i): 0..3 bytes of s1, 4..7 bytes of s0,
 * 8 / 9 / 10 / 11 bit 7 of byte 1 / 3 / 5 / 7 replicated, 12 -> 0x00, >= 13 -> 0xff */
SW_FN uint32_t sw_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
#if SW_DEV
    return __builtin_amdgcn_perm(s0, s1, sel);
#else
    const uint64_t in = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = (sel >> (8 * i)) & 0xffu;
        uint32_t b;
        if (s >= 13) b = 0xff;
        else if (s == 12) b = 0;
        else if (s >= 8) b = ((in >> (8 * (2 * (s - 8) + 1) + 7)) & 1u) ? 0xffu : 0u;
        else b = (uint32_t)(in >> (8 * s)) & 0xffu;
        r |= b << (8 * i);
    }
    return r;
#endif
}
/* v_alignbyte_b32: ({s0, s1} >> (8 * s2[1:0])) & 0xffffffff (gfx950 reads two bits of s2: checked on the device by
 * tests/test_gpu_swar.py); used with s0 == s1 as a byte rotation, where 4 and 0 both mean "no rotation" */
SW_FN uint32_t sw_alignbyte(uint32_t s0, uint32_t s1, uint32_t s2)
{
#if SW_DEV
    return __builtin_amdgcn_alignbyte(s0, s1, s2);
#else
    const uint64_t in = ((uint64_t)s0 << 32) | s1;
    return (uint32_t)(in >> (8 * (s2 & 3u)));
#endif
}
/* v_bitop3_b32: bit i of the result = bit ((a_i << 2) | (b_i << 1) | c_i) of the truth table */
template <int TT>
SW_FN uint32_t sw_bitop3(uint32_t a, uint32_t b, uint32_t c)
{
#if SW_DEV
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0;
    for (int m = 0; m < 8; ++m)
        if ((TT >> m) & 1) r |= ((m & 4) ? a : ~a) & ((m & 2) ? b : ~b) & ((m & 1) ? c : ~c);
    return r;
#endif
}
#define SW_TT_SEL 0xca   /* a ? b : c            */
#define SW_TT_XOR3 0x96  /* a ^ b ^ c            */
#define SW_TT_AND3 0x80  /* a & b & c            */
#define SW_TT_ANDOR 0xea /* (a & b) | c          */
#define SW_TT_NANDOR 0xae /* (~a & b) | c        */
#define SW_TT_A_AND_BORC 0xe0 /* a & (b | c)     */
#define SW_TT_XORAND 0x28 /* (a ^ b) & c         */
#define SW_TT_BFI_C 0xd8 /* c ? b : a            */
#define SW_TT_ANDNOT_OR 0xf8 /* a | (b & c)      */
#define SW_TT_XNOR_AND 0x82 /* ~(a ^ b) & c        */

/* a constant that must live in a VGPR (VOP3 takes no literal, and an SGPR operand halves the issue rate) */
/* `dep`: any uniform value of the scope the constant belongs to.  The asm is not volatile, so the compiler hoists it out of every
 * loop in which its inputs do not change: out of the layer loop, as wanted - and, without `dep`, out of the iteration loop too,
 * across the syndrome stage, where the seventeen registers are then spilled (an asm result cannot be rematerialised). */
SW_FN uint32_t sw_vconst(uint32_t k, uint32_t dep = 0u)
{
#if SW_DEV
    uint32_t r;
    asm("v_mov_b32 %0, %1 ; %2" : "=v"(r) : "s"(k), "s"(dep));
    return r;
#else
    (void)dep;
    return k;
#endif
}

/* the register constants of the layer step, built ONCE per kernel: inside the layer function they would sit in a conditionally
 * executed block (the per-degree instances), out of which the compiler may not move an asm statement */
struct SwK {
    uint32_t c78, c0642, cfc, sel_sign, tt_lo, tt_hi, oh_lo, oh_hi;
    uint32_t cbit[8]; /* 0x01010101 << e; [0] = 0x01010101, [7] = 0x80808080 */
    uint32_t c7f;
    uint32_t c70;     /* NMS only: its minima keep 16 levels */
};
SW_FN SwK sw_consts(uint32_t dep = 0u)
{
    SwK k;
    k.c78 = sw_vconst(0x78787878u, dep); k.c0642 = sw_vconst(0x06040200u, dep); k.cfc = sw_vconst(0xfcu, dep);
    k.sel_sign = sw_vconst(0x0b090a08u, dep);
    /* thermometer code of min(|t|, 7); entry 7 equals what v_perm returns for a saturated selector */
    k.tt_lo = sw_vconst(0x07030100u, dep); k.tt_hi = sw_vconst(0xff3f1f0fu, dep);
    k.oh_lo = sw_vconst(0x08040201u, dep); k.oh_hi = sw_vconst(0x80402010u, dep);
    for (int e = 0; e < 8; ++e) k.cbit[e] = sw_vconst(0x01010101u << e, dep);
    k.c7f = sw_vconst(0x7f7f7f7fu, dep);
    k.c70 = sw_vconst(0x70707070u, dep);
    return k;
}

/* 0xff in every byte whose bit 7 is set, else 0x00 (two instructions: shift + v_perm with selectors 8..11) */
SW_FN uint32_t sw_mask7(uint32_t x, uint32_t sel_sign) { return sw_perm(x, x << 8, sel_sign); }
#define SW_SEL_SIGN 0x0b090a08u

/* per-byte population count of the low seven bits (thermometer code -> number) */
SW_FN uint32_t sw_popcount7(uint32_t x)
{
    x &= 0x7f7f7f7fu;
    x = x - ((x >> 1) & 0x55555555u);
    x = (x & 0x33333333u) + ((x >> 2) & 0x33333333u);
    return (x + (x >> 4)) & 0x0f0f0f0fu;
}
/* 0xff in every byte that is zero; bytes must be <= 0x7f */
SW_FN uint32_t sw_zero_mask(uint32_t x, uint32_t sel_sign) { return ~sw_mask7(x + 0x7f7f7f7fu, sel_sign); }

/* ---- compressed messages of the four rows of one lane in one layer (24 bytes, streamed through HBM) ---------------
 * x[g] bit 8 k + e : the message on edge 8 g + e of row k is negative (0 for a zero message, FAID / 2B1C)
 * cw byte k        : c2 | c1 << 3 | (message on the arg-min edge negative) << 6
 * pa[h]            : LDS byte addresses of the arg-min variable nodes of rows 2 h (low half) and 2 h + 1 (high half) */
struct SwRow {
    uint32_t x[3];
    uint32_t cw;
    uint32_t pa[2];
};

/* decoder constants a layer needs (uniform over the wave) */
struct SwParams {
    uint32_t lut_lo, lut_hi;       /* V2C_map_it* as 8 bytes (entries 0..3 / 4..7) for this iteration           */
    uint32_t ef_lo, ef_hi;         /* error-floor table (DecodeMethod 5)                                        */
    int32_t f1, f2;                /* Factor_1 / Factor_2 (OMS offsets, NMS numerators)                          */
    int32_t window;                /* nombre_iterations <= floor_iter_thresh                                     */
    int32_t ef_tables;             /* EF_ELIMINATION >= 1 (always so for DecodeMethod 5; 0, 1 or 2 for DecodeMethod 2)  */
    uint32_t nms_t[4];             /* NMS with one factor: thermometer code of cste(m) = min((m * Factor) >> 5, 7) for m = 0..15 as 16
                                    * bytes (sw_nms_tables); m >= 16 saturates to 0xff, the code of 7                     */
    uint32_t oms_lo[2], oms_hi[2]; /* min-sum decoders: the selective offset + clamp as 8-entry byte tables over the minimum
                                    * (0..7), [0] the ordinary rule, [1] the rule of an unsatisfied row inside the window
                                    * (sw_oms_tables fills them from f1 / f2)                                         */
};

/* LDS seen by the layer step: byte offsets inside the codeword's En image (block column cb at cb * 256). */
#if SW_DEV
struct SwLds {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    SW_FN uint32_t rd32(uint32_t a) const { return *(const lds_u32*)(size_t)a; }
    SW_FN void wr32(uint32_t a, uint32_t v) const { *(lds_u32*)(size_t)a = v; }
    SW_FN uint32_t rd8(uint32_t a) const { return *(const lds_u8*)(size_t)a; }
    SW_FN void wr8(uint32_t a, uint32_t v) const { *(lds_u8*)(size_t)a = (uint8_t)v; }
};
#else
struct SwLds {
    uint8_t* base;
    uint32_t rd32(uint32_t a) const { return (uint32_t)base[a] | ((uint32_t)base[a + 1] << 8) | ((uint32_t)base[a + 2] << 16) | ((uint32_t)base[a + 3] << 24); }
    void wr32(uint32_t a, uint32_t v) const { base[a] = (uint8_t)v; base[a + 1] = (uint8_t)(v >> 8); base[a + 2] = (uint8_t)(v >> 16); base[a + 3] = (uint8_t)(v >> 24); }
    uint32_t rd8(uint32_t a) const { return base[a]; }
    void wr8(uint32_t a, uint32_t v) const { base[a] = (uint8_t)v; }
};
#endif

#define SW_BIAS_EN 120 /* LDS byte = En + 120 */

/* LDS byte position of variable node v (index inside the code word) in the interleaved image */
SW_FN uint32_t sw_en_pos(uint32_t v) { return (v & ~255u) | ((v & 63u) << 2) | ((v >> 6) & 3u); }

/* DecodeMethods whose rows follow Decode_OMS (1, 3, 4) / use the plain sign and no upper clamp on t (those and NMS, 0) */
#define SW_OMS(M) ((M) == 1 || (M) == 3 || (M) == 4)
#define SW_MINSUM(M) ((M) == 0 || SW_OMS(M))

/* selective offset of OMS_MODE 1 on one minimum (CDecoder_OMS.cpp:388-425) */
SW_FN int sw_oms_offset(int x, bool window, bool F, int f1, int f2)
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

/* CDecoder_OMS.cpp:388-432 as two tables: entry x = min(offset(x), SAT_POS_MSG) for both branches of the rule */
SW_FN void sw_oms_tables(SwParams& p)
{
    for (int r = 0; r < 2; ++r) {
        uint32_t lo = 0, hi = 0;
        for (int x = 0; x < 8; ++x) {
            const int a = sw_oms_offset(x, r != 0, r != 0, p.f1, p.f2);
            const uint32_t v = (uint32_t)((a > 7 ? 7 : a) & 0xff);
            if (x < 4) lo |= v << (8 * x); else hi |= v << (8 * (x - 4));
        }
        p.oms_lo[r] = lo; p.oms_hi[r] = hi;
    }
}

/* Normalised min-sum (CLDPC::Decode, CLDPC.cpp:337-352): cste(m) = min(pack_s8((uint16)(m * Factor) >> 5), 7) in 16-bit lanes
 * (VECTOR_MUL = _mm256_mullo_epi16, VECTOR_DIV32 = _mm256_srli_epi16(.., 5), VECTOR_PACK = _mm256_packs_epi16), m = 0..31. */
SW_HD int sw_nms_cste(int m, int factor)
{
    const uint16_t p = (uint16_t)((uint16_t)m * (uint16_t)(int16_t)factor);
    const int16_t q = (int16_t)(p >> 5);
    const int r = q > 127 ? 127 : q;
    return r > 7 ? 7 : r;
}
/* The byte-parallel layer step applies a row's magnitude function to the LEVELS of the minimum search: with one factor for both
 * minima cste() plays the part of the FAID table (a non-decreasing map commutes with the minimum, and edges that tie after the
 * map get the same message whichever of them is called the arg-min - DESIGN.md 3.2).  The search keeps 16 levels of |t| (two
 * v_perm_b32 tables), so cste() must be non-decreasing and have reached its end value 7 at m = 15: true for Factor >= 15 up to
 * the 16-bit wrap (Factor <= 2114), i.e. for every normalisation factor in use (24 / 32 = 0.75 ...); other factors, and two
 * different factors (a tie then gives the two minima different messages), stay on the two-rows-per-lane kernel. */
SW_HD bool sw_nms_fits(int f1, int f2)
{
    if (f1 != f2) return false;
    for (int m = 1; m < 32; ++m)
        if (sw_nms_cste(m, f1) < sw_nms_cste(m - 1, f1)) return false;
    return sw_nms_cste(15, f1) == 7;
}
SW_HD void sw_nms_tables(int factor, uint32_t nms_t[4])
{
    for (int w = 0; w < 4; ++w) nms_t[w] = 0;
    for (int m = 0; m < 16; ++m) {
        const int c = sw_nms_cste(m, factor);
        const uint32_t code = c >= 7 ? 0xffu : ((1u << c) - 1u); /* bit i set iff c > i */
        nms_t[m >> 2] |= code << (8 * (m & 3));
    }
}

/* the two-stage saturating update En' = sat31(tc + L), tc = sat31(t) (FAID, CDecoder_FAID.cpp:672, :919-920) or
 * max(t, -31) (min-sum decoders, CDecoder_OMS.cpp:371, :466), written as ONE clamp whose limits do not depend on the sign of
 * the new message L = +-c:
 *   L = +c : En' = c + clamp(t, -31, 31 - c)        L = -c : En' = clamp(t - c, -31, 31 - c)      (min-sum, L = -c: ..., 31)
 * i.e. z = t - (L < 0 ? c : 0), En' = clamp(z, -31, hi) + (L < 0 ? 0 : c): two sign-selected byte constants per edge (za, sb),
 * the rest is per row.  Working byte zb = z + 159, so that "not under" is bit 7 of zb and "over" is bit 7 of zb - oc. */
struct SwUpd {
    uint32_t za[2], sb[2]; /* [1]: chosen where the mask is 0xff, [0]: where it is 0x00 */
    uint32_t oc[2], hi[2]; /* the two entries differ for the min-sum decoders only */
};
/* c: magnitude bytes (0..7); flip: byte mask, 0xff where mask polarity is inverted (row parity F) */
/* `bias`: what the caller's tb carries on top of t + 128 per byte (FAID keeps the v_perm selector's 0 / 2 / 4 / 6 in it, see
 * pass 1): folded into the constant that meets tb */
template <bool MINSUM, bool FLIP = true>
SW_FN SwUpd sw_update_consts(uint32_t c, uint32_t flip, uint32_t bias)
{
    const uint32_t za_p = 0x1f1f1f1fu - bias, za_n = za_p - c;                  /* zb = tb + za = t + 159 - (L < 0 ? c : 0)   */
    const uint32_t sb_p = 0x27272727u - c, sb_n = 0x27272727u;                  /* En' + 120 = clamped zb - sb                */
    const uint32_t oc_p = 0x3f3f3f3fu - c, oc_n = MINSUM ? 0x3f3f3f3fu : oc_p;  /* over <=> z > hi <=> zb - oc >= 128         */
    const uint32_t hi_p = 0xbebebebeu - c, hi_n = MINSUM ? 0xbebebebeu : hi_p;  /* hi + 159                                   */
    SwUpd u;
    if (!FLIP) { /* the caller's mask already is "new message not negative" */
        u.za[1] = za_p; u.za[0] = za_n; u.sb[1] = sb_p; u.sb[0] = sb_n; u.oc[1] = oc_p; u.oc[0] = oc_n; u.hi[1] = hi_p; u.hi[0] = hi_n;
        return u;
    }
    u.za[1] = sw_bitop3<SW_TT_SEL>(flip, za_n, za_p); u.za[0] = sw_bitop3<SW_TT_SEL>(flip, za_p, za_n);
    u.sb[1] = sw_bitop3<SW_TT_SEL>(flip, sb_n, sb_p); u.sb[0] = sw_bitop3<SW_TT_SEL>(flip, sb_p, sb_n);
    if (MINSUM) {
        u.oc[1] = sw_bitop3<SW_TT_SEL>(flip, oc_n, oc_p); u.oc[0] = sw_bitop3<SW_TT_SEL>(flip, oc_p, oc_n);
        u.hi[1] = sw_bitop3<SW_TT_SEL>(flip, hi_n, hi_p); u.hi[0] = sw_bitop3<SW_TT_SEL>(flip, hi_p, hi_n);
    } else {
        u.oc[1] = u.oc[0] = oc_p; u.hi[1] = u.hi[0] = hi_p;
    }
    return u;
}
/* tb: t + 128 (+ bias) per byte; ms: byte mask picking entry [1] of the constants; c80: 0x80808080 in a register */
template <bool MINSUM>
SW_FN uint32_t sw_update(uint32_t tb, uint32_t ms, const SwUpd& u, uint32_t sel_sign, uint32_t c80)
{
    const uint32_t za = sw_bitop3<SW_TT_SEL>(ms, u.za[1], u.za[0]);
    const uint32_t sb = sw_bitop3<SW_TT_SEL>(ms, u.sb[1], u.sb[0]);
    const uint32_t oc = MINSUM ? sw_bitop3<SW_TT_SEL>(ms, u.oc[1], u.oc[0]) : u.oc[1];
    const uint32_t hi = MINSUM ? sw_bitop3<SW_TT_SEL>(ms, u.hi[1], u.hi[0]) : u.hi[1];
    const uint32_t zb = tb + za;
    const uint32_t mo = sw_mask7(zb - oc, sel_sign); /* over      */
    const uint32_t mq = sw_mask7(zb, sel_sign);      /* not under */
    return sw_bitop3<SW_TT_SEL>(mo, hi, sw_bitop3<SW_TT_SEL>(mq, zb, c80)) - sb;
}

/* ERA helpers.  Plane bit of the variable node at LDS byte address a (interleaved image: block column in the high bits,
 * dword = node mod 64, byte = node div 64); returns 0 / 1. */
SW_FN uint32_t sw_plane_bit(const SwLds& lds, uint32_t era_plane, uint32_t a)
{
    const uint32_t v = (a & ~255u) | ((a >> 2) & 63u) | ((a & 3u) << 6); /* node index */
    return (lds.rd32(era_plane + ((v >> 5) << 2)) >> (v & 31u)) & 1u;
}

/* ---- one layer -----------------------------------------------------------------------------------------------------
 * Tab: tab.s4(j) = 4 * shift and tab.cb256(j) = block column * 256 of edge j (uniform), tab.sb_dyn4(4 * idx) =
 * (block column * 256) << 16 | 4 * shift for a per-lane edge index.
 * rowpar: byte mask, 0xff in byte k if the syndrome bit of row i + 64 k is set (only read by the OMS selective offset
 * and the 2B1C error-floor tables); lme: unsat < floor_err_count for this codeword.
 * DecodeMethod 0 (NMS, one factor): the minimum search keeps 16 levels of |t| and maps them through cste() (sw_nms_tables). */
/* Two waves per codeword (WAVES == 2, lnsfaid_kernel5.hip): the layer's edges are dealt to the two waves by the parity of their index,
 * every lane of either wave still works on its four rows.  Wave 0 does everything that is per row (old arg-min patch, the merge of
 * the two partial minimum searches, the new magnitudes, the arg-min edge, the row's record); wave 1 only passes 1 and 2 over its
 * edges.  They meet four times per layer (xch.barrier()) and exchange through LDS (xch.put / xch.get, one dword per lane and slot):
 *   A  wave 0 has patched the old arg-min nodes                                  -> both start pass 1
 *   B  wave 1 has published its partial minima / index accumulators / sign XOR   -> wave 0 merges, finishes the row
 *   C  wave 0 has READ the new arg-min nodes and published c2, F                  -> both start pass 2 (which overwrites the nodes)
 *   D  wave 1 has written its En and published its sign bits                     -> wave 0 writes the exact arg-min En, the record
 * SwNoXch: the single-wave build (every call compiles away). */
#if SW_DEV
#define SW_MFN __device__ __forceinline__
#else
#define SW_MFN inline
#endif
struct SwNoXch {
    SW_MFN void put(int, uint32_t) const {}
    SW_MFN uint32_t get(int) const { return 0u; }
    SW_MFN void barrier() const {}
};
#define SW_OWN(j) (WAVES == 1 || (((j) & 1) == WAVE))

template <int METHOD, int DEG, bool ERA = false, int WAVES = 1, int WAVE = 0, class Tab, class Xch = SwNoXch>
SW_FN SwRow sw_layer_step(const SwLds& lds, const Tab& tab, const SwParams& p, const SwK& K, uint32_t lane, int deg, SwRow cur, bool fresh,
                          uint32_t rowpar, bool lme, uint32_t era_edges = 0u, uint32_t era_plane = 0u, const Xch& xch = Xch())
{
    static_assert(WAVES == 1 || !ERA, "the erasing variant is built for one wave per codeword only");
    /* ERA (EF_ELIMINATION 2, CDecoder_FAID.cpp:673-680; the caller instantiates it only inside the error-floor window of a
     * codeword with few unsatisfied checks): era_edges bit j = edge j is the first edge, in row order, of a block column of
     * weight REGULAR_COL_WEIGHT; era_plane = LDS byte offset of a bit plane over the variable nodes, bit set = every check of
     * the node was unsatisfied at this iteration's syndrome stage.  On those edges the V2C becomes 0 for such nodes and its
     * sign is the sign of En itself (the back-track of :682 with vContr == 0). */
    static_assert(!ERA || METHOD == 2, "the erasure exists in Decode_FAID only");
    constexpr int NJ = DEG > 0 ? DEG : SW_MAX_DEG;
    constexpr bool MINSUM = SW_MINSUM(METHOD);
    const uint32_t c01 = K.cbit[0], c80 = K.cbit[7], c7f = K.c7f, c78 = K.c78, c0642 = K.c0642, cfc = K.cfc, sel_sign = K.sel_sign;
    const uint32_t tid4 = lane << 2;
    /* the layer's circulants first (scalar loads share the LDS counter: in flight together with LDS reads they force full drains) */
    uint32_t s4j[NJ], cbj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { s4j[j] = (DEG > 0 || j < deg) ? tab.s4(j) : 0u; cbj[j] = (DEG > 0 || j < deg) ? tab.cb256(j) : 0u; }

    /* ---- the row's old messages: +-c2 on every edge (8 +- c2 as a v_perm table indexed 2 k + negative) ---- */
    const uint32_t c2o = cur.cw & 0x07070707u, c1o = (cur.cw >> 3) & 0x07070707u;
    const uint32_t kp = 0x08080808u - c2o, kn = 0x08080808u + c2o;
    /* FAID: tb = En + 120 + K + selector, with the selector's row part (0 / 2 / 4 / 6 per byte) folded into the table, so that
     * tb - selector = t + 128 - b is the back-tracked sign word in ONE subtraction; pass 2 compensates in its constants */
    const uint32_t bias = MINSUM ? 0u : 0x06040200u;
    const uint32_t kt_lo = sw_perm(kn, kp, 0x05010400u) + (MINSUM ? 0u : 0x02020000u), kt_hi = sw_perm(kn, kp, 0x07030602u) + (MINSUM ? 0u : 0x06060404u);
    const uint32_t tt_lo = K.tt_lo, tt_hi = K.tt_hi; /* thermometer code of min(|t|, 7) */

    /* ---- the old arg-min edge carries c1, not c2: move its En by the difference so that "every edge carries c2" holds ---- */
    uint32_t padd = 0, psub = 0; /* what the patch below adds to / takes from the old arg-min nodes' LDS bytes */
#ifndef SW_EXP_NO_OLDPATCH
    if (!fresh && WAVE == 0) {
        const uint32_t a0 = cur.pa[0] & 0xffffu, a1 = cur.pa[0] >> 16, a2 = cur.pa[1] & 0xffffu, a3 = cur.pa[1] >> 16;
        SW_SCHED_FENCE(); /* the four reads back to back: one LDS round trip, not one per read */
        const uint32_t g0 = lds.rd8(a0), g1 = lds.rd8(a1), g2 = lds.rd8(a2), g3 = lds.rd8(a3);
        const uint32_t mneg = sw_mask7(cur.cw << 1, sel_sign); /* bit 6: the arg-min message is negative */
        /* En - L(c1) = (En - sigma (c1 - c2)) - sigma c2 */
        padd = sw_bitop3<SW_TT_SEL>(mneg, c1o, c2o); psub = sw_bitop3<SW_TT_SEL>(mneg, c2o, c1o);
        SW_SCHED_FENCE();
        const uint32_t g = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        const uint32_t r = g + padd - psub;
        lds.wr8(a0, r); lds.wr8(a1, r >> 8); lds.wr8(a2, r >> 16); lds.wr8(a3, r >> 24);
    }
#endif
    if (WAVES == 2) xch.barrier(); /* A */

    SW_SCHED_FENCE();
    uint32_t tb[NJ], ts[NJ], ms[NJ], ad[NJ], rq[NJ], ld[NJ];
    uint32_t t1 = 0xffffffffu, t2 = 0xffffffffu, ta[5] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
    uint32_t sx = 0;

    /* ---- pass 1 (CDecoder_FAID.cpp:662-861, CDecoder_OMS.cpp:363-380): all addresses, then all reads, then the arithmetic ---- */
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if ((DEG > 0 || j < deg) && SW_OWN(j)) {
            const uint32_t x4 = tid4 + s4j[j];
#if SW_DEV
            uint32_t a;
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(x4), "v"(cfc), "s"(cbj[j]));
#else
            const uint32_t a = (x4 & cfc) | cbj[j];
#endif
            ad[j] = a; rq[j] = x4 >> 8;
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if ((DEG > 0 || j < deg) && SW_OWN(j)) ld[j] = lds.rd32(ad[j]);
    SW_SCHED_FENCE();
    /* The arithmetic of an edge is one long dependency chain and the hardware issues a wave's instructions in order, so the
     * statements below are written stage by stage over groups of SW_ILP edges: consecutive instructions then belong to
     * different edges and do not wait for each other (two waves per SIMD cannot hide that latency). */
#pragma unroll
    for (int j0 = 0; j0 < NJ; j0 += SW_ILP) {
        uint32_t r_[SW_ILP], x_[SW_ILP], s_[SW_ILP], k_[SW_ILP], a_[SW_ILP], i_[SW_ILP], u_[SW_ILP];
#define SW_EDGES(...) _Pragma("unroll") for (int g = 0; g < SW_ILP; ++g) { const int j = j0 + g; if (j < NJ && (DEG > 0 || j < deg) && SW_OWN(j)) { __VA_ARGS__ } }
        SW_EDGES(r_[g] = sw_alignbyte(ld[j], ld[j], rq[j]);)                                  /* byte k = En + 120 of row k */
        SW_EDGES(x_[g] = (j & 7) ? cur.x[j >> 3] >> (j & 7) : cur.x[j >> 3];)
        SW_EDGES(s_[g] = sw_bitop3<SW_TT_ANDOR>(x_[g], c01, c0642);)
        SW_EDGES(k_[g] = sw_perm(kt_hi, kt_lo, s_[g]);)
        SW_EDGES(tb[j] = r_[g] + k_[g];)                          /* t + 128, VECTOR_SUB_AND_SATURATE comes in pass 2 */
        if (ERA) {
            SW_EDGES(x_[g] &= c01;)                               /* b: the old message on this edge is negative */
            SW_EDGES(
                if ((era_edges >> j) & 1u) {
                    uint32_t em = 0, match = 0;
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) {
                        const uint32_t a = ad[j] + ((rq[j] + (uint32_t)k) & 3u); /* LDS byte of row k's node on this edge */
                        em |= sw_plane_bit(lds, era_plane, a) ? (0xffu << (8 * k)) : 0u;
                        const uint32_t pold = (k & 1) ? cur.pa[k >> 1] >> 16 : cur.pa[k >> 1] & 0xffffu;
                        match |= (!fresh && a == pold) ? (0xffu << (8 * k)) : 0u;    /* that byte carries the patch */
                    }
                    const uint32_t en_true = r_[g] - (padd & match) + (psub & match);
                    const uint32_t neg01 = ~((en_true + 0x08080808u) >> 7) & c01;      /* En < 0 */
                    tb[j] = sw_bitop3<SW_TT_SEL>(em, c80 + bias, tb[j]);               /* vContr = 0 */
                    x_[g] = sw_bitop3<SW_TT_SEL>(em, neg01, x_[g]);                    /* its sign: the sign of En */
                })
        }
        /* FAID: a zero V2C takes the sign of En (CDecoder_FAID.cpp:682); En == Lold there, so it is the stored sign b: bit 7 of
         * t + 128 - b.  k_ still holds the selector 2 k + b of the edge (the erasing variant rebuilds it from its own b) */
        SW_EDGES(ts[j] = MINSUM ? tb[j] : tb[j] - (ERA ? (x_[g] | c0642) : s_[g]);)
        SW_EDGES(ms[j] = sw_mask7(ts[j], sel_sign);)
        /* |t| -> min(|t|, 7) -> thermometer.  With m = 0xff where the (back-tracked) sign is "not negative":
         *   not negative: ts ^ 0x80 = t - b,  |t| = that + b          negative: ts ^ 0x7f = -t - 1 + b,  |t| = that + 1 - b
         * (b = 0 for the min-sum decoders); 0x78 is added on top so that |t| >= 8 reaches bit 7, which the table look-up
         * turns into the saturated code */
        SW_EDGES(a_[g] = sw_bitop3<SW_TT_XOR3>(ts[j], c7f, ms[j]);)
        if (METHOD == 0) {
            /* NMS: 16 levels.  0x70 on top so that |t| >= 16 reaches bit 7 (both look-ups then return the saturated code); bit 3
             * picks the table: entries 8..15 (nms_t[3], nms_t[2]) or 0..7 (nms_t[1], nms_t[0]) */
            uint32_t h_[SW_ILP];
            SW_EDGES(i_[g] = sw_bitop3<SW_TT_NANDOR>(ms[j], c01, K.c70);)
            SW_EDGES(a_[g] = (a_[g] + i_[g]) & 0x8f8f8f8fu;)
            SW_EDGES(h_[g] = sw_mask7(a_[g] << 4, sel_sign);)
            SW_EDGES(a_[g] &= 0x87878787u;)
            SW_EDGES(i_[g] = sw_perm(p.nms_t[3], p.nms_t[2], a_[g]);)
            SW_EDGES(u_[g] = sw_perm(p.nms_t[1], p.nms_t[0], a_[g]);)
            SW_EDGES(u_[g] = sw_bitop3<SW_TT_SEL>(h_[g], i_[g], u_[g]);)
        } else {
            SW_EDGES(i_[g] = MINSUM ? sw_bitop3<SW_TT_NANDOR>(ms[j], c01, c78) : sw_bitop3<SW_TT_XNOR_AND>(x_[g], ms[j], c01);)
            SW_EDGES(a_[g] = (MINSUM ? a_[g] + i_[g] : a_[g] + i_[g] + c78) & 0x87878787u;)     /* v_add3_u32 */
            SW_EDGES(u_[g] = sw_perm(tt_hi, tt_lo, a_[g]);)
        }
        SW_EDGES(
            if (WAVES == 2) sx ^= ts[j]; /* (the neighbour edge belongs to the other wave) */
            else if (j & 1) sx = sw_bitop3<SW_TT_XOR3>(sx, ts[j - 1], ts[j]); else if (j == (DEG > 0 ? DEG : deg) - 1) sx ^= ts[j];
            t2 = sw_bitop3<SW_TT_A_AND_BORC>(t2, t1, u_[g]);      /* VECTOR_MIN_2 with the old min1 */
            t1 &= u_[g];
            _Pragma("unroll") for (int b = 0; b < 5; ++b) if (((j >> b) & 1) == (b >= 3 ? 1 : 0)) ta[b] &= u_[g];)
    }

    if (WAVES == 2) {
        /* index bit 0 is the wave: ta[0] (edges with the bit clear) is wave 0's own minimum */
        if (WAVE == 1) {
            xch.put(0, t1); xch.put(1, t2); xch.put(2, ta[1]); xch.put(3, ta[2]); xch.put(4, ta[3]); xch.put(5, ta[4]); xch.put(6, sx);
        }
        xch.barrier(); /* B */
        if (WAVE == 0) {
            const uint32_t o1 = xch.get(0), o2 = xch.get(1);
            ta[0] = t1;
            t2 = sw_bitop3<SW_TT_AND3>(t2, o2, t1 | o1); /* second minimum of the union: min(max(m1a, m1b), m2a, m2b) */
            t1 &= o1;
            ta[1] &= xch.get(2); ta[2] &= xch.get(3); ta[3] &= xch.get(4); ta[4] &= xch.get(5);
            sx ^= xch.get(6);
        }
    }
    uint32_t c2n_x = 0, fm_x = 0; /* what wave 1 takes over from wave 0 */
    if (WAVES == 2 && WAVE == 1) {
        xch.barrier(); /* C */
        c2n_x = xch.get(0); fm_x = xch.get(1);
    }

    /* ---- the row's new magnitudes ---- */
    uint32_t min1 = sw_popcount7(t1), min2 = sw_popcount7(t2);
    uint32_t c1n, c2n;
    if (METHOD == 0) {
        /* the levels of the search ARE cste() of the minima (CLDPC.cpp:337-352: cste_2 from min1, cste_1 from min2, one factor) */
        c2n = min1;
        c1n = min2;
    } else if (SW_OMS(METHOD)) {
        /* selective offset and clamp (cste_2 from min1, cste_1 from min2, CDecoder_OMS.cpp:431-432) as table look-ups; rows
         * with an unsatisfied check take the other rule inside the error-floor window of a codeword with few of them (:388) */
        c2n = sw_perm(p.oms_hi[0], p.oms_lo[0], min1);
        c1n = sw_perm(p.oms_hi[0], p.oms_lo[0], min2);
        if (p.window && lme) {
            c2n = sw_bitop3<SW_TT_SEL>(rowpar, sw_perm(p.oms_hi[1], p.oms_lo[1], min1), c2n);
            c1n = sw_bitop3<SW_TT_SEL>(rowpar, sw_perm(p.oms_hi[1], p.oms_lo[1], min2), c1n);
        }
    } else {
        /* uniform non-decreasing table applied after the search; offset 0 (CDecoder_FAID.cpp:864-866) */
        c2n = sw_perm(p.lut_hi, p.lut_lo, min1);
        c1n = sw_perm(p.lut_hi, p.lut_lo, min2);
        if ((METHOD == 5 || p.ef_tables) && p.window && lme) { /* mask_eef per row (CDecoder_FAID.cpp:713-720) */
            c2n = sw_bitop3<SW_TT_SEL>(rowpar, sw_perm(p.ef_hi, p.ef_lo, min1), c2n);
            c1n = sw_bitop3<SW_TT_SEL>(rowpar, sw_perm(p.ef_hi, p.ef_lo, min2), c1n);
        }
    }
    /* new message on edge j is negative iff s_j ^ XOR_all(s) ^ (deg odd) (the 0xC0 / 0x40 constants of
     * CDecoder_FAID.cpp:902-917); with nn_j = bit 7 of ts_j = "V2C not negative" that is: not negative iff nn_j ^ F,
     * F = bit 7 of the XOR of all ts */
    const uint32_t fm = (WAVES == 2 && WAVE == 1) ? fm_x : sw_mask7(sx, sel_sign);
    if (WAVES == 2 && WAVE == 1) { c2n = c2n_x; c1n = 0u; } /* (everything per row below is wave 0's: dead code here) */

    /* ---- binary index of an edge that attains the minimum.  Bits 0..2: ta[b] runs over the edges whose index has bit b clear,
     * the bit is 1 iff none of them attains it.  Bits 3 and 4 the other way round (fewer edges have them set): ta[3] runs over
     * edges 8..15, ta[4] over edges 16..23, the bit is 1 iff one of them attains it; bit 3 gives way to bit 4.  With a unique
     * minimum that is its index; a tie leaves 16 / 8 / 0 + the AND of the tied indices' low bits, not above one of the tied
     * edges in that range, so still an edge of the row - and in a tie c1 == c2 ---- */
    uint32_t idx = 0;
#pragma unroll
    for (int b = 4; b >= 0; --b) {
        const uint32_t dd = sw_bitop3<SW_TT_XORAND>(ta[b], t1, c7f);
        const uint32_t ne = (dd + c7f) >> (7 - b); /* bit b of every byte: ta[b] != t1 */
        /* (a row of the generic instance with no edge in 16..23 / 8..15 leaves that accumulator untouched: bit stays 0) */
        if (b == 4) idx = (DEG == 0 && deg <= 16) ? 0u : ~ne & (0x01010101u << 4);
        else if (b == 3) idx = (DEG == 0 && deg <= 8) ? idx : sw_bitop3<0xf2>(idx, ne | (idx >> 1), 0x01010101u << 3); /* a | (~b & c) */
        else idx = sw_bitop3<SW_TT_ANDNOT_OR>(idx, ne, 0x01010101u << b);
    }

    /* ---- the new arg-min edge: address, exact V2C, exact new En.  Its node address comes from the layer's edge table through
     * ds_bpermute_b32 and its (still old) En from LDS: two dependent round trips, which nothing else of the wave would cover
     * (two waves per SIMD).  Pass 2 does not depend on them, so its first group of edges is computed between the issue of
     * the table look-up and its use, and the rest of it between the issue of the En reads and theirs.  LDS operations of a
     * wave execute in order: the En reads are ISSUED before the first write of pass 2, so they return the old values. ---- */
    uint32_t pa[4], sbk[4], gb = 0, xb = 0;
    const uint32_t idx4 = idx << 2;
#pragma unroll
#ifdef SW_EXP_NO_ARGMIN /* timing experiment only: results are wrong */
    for (int k = 0; k < 4; ++k) sbk[k] = (cbj[k] << 16) | s4j[k];
#else
    for (int k = 0; k < 4; ++k) sbk[k] = WAVE == 0 ? tab.sb_dyn4((idx4 >> (8 * k)) & 0x7cu) : 0u;
#endif
    SW_SCHED_FENCE();

    /* ---- pass 2 (CDecoder_FAID.cpp:909-929, CDecoder_OMS.cpp:452-471): every edge as if it carried c2 ---- */
    const SwUpd u2 = sw_update_consts<MINSUM>(c2n, fm, bias);
    uint32_t ns[3] = { 0u, 0u, 0u };
    uint32_t cbit[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cbit[e] = K.cbit[e];
    /* The two clamp tests of an edge, "over" = bit 7 of zb - oc and "not under" = bit 7 of zb, as select masks.  SW_CLAMP7: 0x7f
     * masks from three full-rate operations each (b = x & 0x80.., b - (b >> 7)) instead of 0xff masks from a left shift and a
     * v_perm_b32 (both half rate): enough here, because bit 7 is set in everything the selects choose between (zb where it is not
     * under, 0x80 where it is, hi = 0xbe - c) */
#ifdef SW_CLAMP7
#define SW_CLAMP_MASKS(MO, MQ, OC, ZA)                                                                               \
        SW_EDGES(MO[g] = OC[g] & c80;)                                                                               \
        SW_EDGES(MQ[g] = ZA[g] & c80;)                                                                               \
        SW_EDGES(MO[g] = MO[g] - (MO[g] >> 7);)                                                                      \
        SW_EDGES(MQ[g] = MQ[g] - (MQ[g] >> 7);)
#define SW_CLAMP_LOW(MQ, ZA) sw_bitop3<0xea>(ZA, MQ, c80) /* (a & b) | c */
#else
#define SW_CLAMP_MASKS(MO, MQ, OC, ZA)                                                                               \
        SW_EDGES(MO[g] = sw_mask7(OC[g], sel_sign);)     /* over      */                                             \
        SW_EDGES(MQ[g] = sw_mask7(ZA[g], sel_sign);)     /* not under */
#define SW_CLAMP_LOW(MQ, ZA) sw_bitop3<SW_TT_SEL>(MQ, ZA, c80)
#endif
    /* one group of edges: the arithmetic (ARITH) and the LDS writes (STORE) separately, stage by stage over the group as in pass 1 */
#define SW_PASS2_ARITH(J0, EN)                                                                                       \
    {                                                                                                                \
        const int j0 = (J0);                                                                                         \
        uint32_t za[SW_ILP], sb[SW_ILP], oc[SW_ILP], hi[SW_ILP], mo[SW_ILP], mq[SW_ILP];                             \
        SW_EDGES(za[g] = sw_bitop3<SW_TT_SEL>(ms[j], u2.za[1], u2.za[0]);)                                           \
        SW_EDGES(sb[g] = sw_bitop3<SW_TT_SEL>(ms[j], u2.sb[1], u2.sb[0]);)                                           \
        SW_EDGES(oc[g] = MINSUM ? sw_bitop3<SW_TT_SEL>(ms[j], u2.oc[1], u2.oc[0]) : u2.oc[1];)                       \
        SW_EDGES(hi[g] = MINSUM ? sw_bitop3<SW_TT_SEL>(ms[j], u2.hi[1], u2.hi[0]) : u2.hi[1];)                       \
        SW_EDGES(za[g] = tb[j] + za[g];)                 /* zb */                                                    \
        SW_EDGES(oc[g] = za[g] - oc[g];)                                                                             \
        SW_CLAMP_MASKS(mo, mq, oc, za)                                                                               \
        SW_EDGES(EN[g] = SW_CLAMP_LOW(mq[g], za[g]);)                                                                \
        SW_EDGES(EN[g] = sw_bitop3<SW_TT_SEL>(mo[g], hi[g], EN[g]);)                                                 \
        SW_EDGES(EN[g] = EN[g] - sb[g];)                                                                             \
        SW_EDGES(EN[g] = sw_alignbyte(EN[g], EN[g], 4u - rq[j]);)                                                    \
        SW_EDGES(ns[j >> 3] = sw_bitop3<SW_TT_ANDNOT_OR>(ns[j >> 3], ms[j], cbit[j & 7]);) /* bit e of byte k: V2C on edge 8 g + e not negative */ \
    }
#define SW_PASS2_STORE(J0, EN)                                                                                       \
    {                                                                                                                \
        const int j0 = (J0);                                                                                         \
        SW_EDGES(lds.wr32(ad[j], EN[g]);)                                                                            \
    }
    uint32_t en0[SW_ILP], en1[SW_ILP];
    SW_PASS2_ARITH(0, en0)
    SW_SCHED_FENCE();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = tid4 + sbk[k]; /* low half: 4 * (lane + shift) < 2048, high half: block column * 256 */
        const uint32_t a = (x & 0xfcu) | (x >> 16);
        const uint32_t q = (x >> 8) & 3u;
        pa[k] = k ? a | ((q + (uint32_t)k) & 3u) : a | q;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#ifdef SW_EXP_NO_ARGMIN
        gb |= (ld[k] & 0xffu) << (8 * k);
#else
        if (WAVE == 0) gb |= lds.rd8(pa[k]) << (8 * k);
#endif
    }
    if (WAVES == 2 && WAVE == 0) { /* the barrier also waits for the reads above: wave 1 may overwrite those nodes from here on */
        xch.put(0, c2n); xch.put(1, fm);
        xch.barrier(); /* C */
    }
    SW_SCHED_FENCE();
    SW_PASS2_STORE(0, en0)
    if (SW_ILP < NJ) { SW_PASS2_ARITH(SW_ILP, en1) SW_PASS2_STORE(SW_ILP, en1) }
    SW_SCHED_FENCE();
    /* the arg-min edges as one-hot bits inside their 8-edge word (byte k: 1 << (index mod 8)) and the word they are in
     * (index div 8 = 0, 1, 2) as byte masks, all four rows at once */
    const uint32_t oh8 = sw_perm(K.oh_hi, K.oh_lo, idx & 0x07070707u);
    const uint32_t in1 = sw_mask7(idx << 4, sel_sign), in2 = sw_mask7(idx << 3, sel_sign); /* index bit 3 / bit 4 */
    {   /* old message on that edge negative: bit (index mod 8) of byte k of sign word (index div 8) */
        const uint32_t w = sw_bitop3<SW_TT_SEL>(in2, cur.x[2], sw_bitop3<SW_TT_SEL>(in1, cur.x[1], cur.x[0]));
        xb = (((w & oh8) + c7f) & c80) >> 7;
    }
    const uint32_t selA = xb | c0642;
    uint32_t tbA = gb + sw_perm(kt_hi, kt_lo, selA); /* carries `bias` like tb[] */
    if (ERA) { /* the arg-min edge of a row may be an erased one (a V2C of 0 usually IS the minimum) */
        uint32_t em = 0, match = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t ik = (idx >> (8 * k)) & 31u;
            if ((era_edges >> ik) & 1u) em |= sw_plane_bit(lds, era_plane, pa[k]) ? (0xffu << (8 * k)) : 0u;
            const uint32_t pold = (k & 1) ? cur.pa[k >> 1] >> 16 : cur.pa[k >> 1] & 0xffffu;
            match |= (!fresh && pa[k] == pold) ? (0xffu << (8 * k)) : 0u;
        }
        const uint32_t en_true = gb - (padd & match) + (psub & match);
        tbA = sw_bitop3<SW_TT_SEL>(em, c80 + bias, tbA);
        xb = sw_bitop3<SW_TT_SEL>(em, ~((en_true + 0x08080808u) >> 7) & c01, xb);
    }
    const uint32_t tsA = MINSUM ? tbA : tbA - (xb | c0642);
    const uint32_t msA = sw_mask7(tsA, sel_sign);
    const uint32_t negA = ~(msA ^ fm); /* byte mask: the new message on the arg-min edge is negative */
    /* used once: flip = 0 leaves [1] = "not negative" constants, [0] = "negative" ones, picked by the combined mask */
    const SwUpd u1 = sw_update_consts<MINSUM, false>(c1n, 0u, bias);
    const uint32_t enA = sw_update<MINSUM>(tbA, ~negA, u1, sel_sign, c80);

#pragma unroll
    for (int jg = 2 * SW_ILP; jg < NJ; jg += SW_ILP) { /* the remaining groups of pass 2 */
        uint32_t en[SW_ILP];
        SW_PASS2_ARITH(jg, en)
        SW_PASS2_STORE(jg, en)
    }
#undef SW_EDGES
#undef SW_PASS2_ARITH
#undef SW_PASS2_STORE
#undef SW_CLAMP_MASKS
#undef SW_CLAMP_LOW
    if (WAVES == 2) {
        if (WAVE == 1) { xch.put(0, ns[0]); xch.put(1, ns[1]); xch.put(2, ns[2]); }
        xch.barrier(); /* D: wave 1's En is in LDS */
        if (WAVE == 1) { SwRow none = { { 0u, 0u, 0u }, 0u, { 0u, 0u } }; return none; }
        ns[0] |= xch.get(0); ns[1] |= xch.get(1); ns[2] |= xch.get(2);
    }
    /* the arg-min edge carries c1: its exact En replaces the as-if value pass 2 wrote (same lane, LDS operations in order; with two
     * waves: after barrier D, behind the other wave's as-if value) */
#ifndef SW_EXP_NO_ARGMIN
#pragma unroll
    for (int k = 0; k < 4; ++k) lds.wr8(pa[k], enA >> (8 * k));
#endif

    /* ---- the row's new compressed messages ---- */
    SwRow out;
    const int n = DEG > 0 ? DEG : deg;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int cnt = n - 8 * g < 0 ? 0 : (n - 8 * g > 8 ? 8 : n - 8 * g);
        const uint32_t valid = 0x01010101u * ((1u << cnt) - 1u);
        out.x[g] = ~(ns[g] ^ fm) & valid; /* negative iff not (nn ^ F) */
    }
    if (!MINSUM) {
        /* A zero message has no sign: stored as "not negative" so that the next iteration's back-track reads
         * "Lmn < 0" straight from the bit.  c2 == 0 zeroes every message of the row but the arg-min's. */
        const uint32_t z2 = sw_zero_mask(c2n, sel_sign), nz1 = ~sw_zero_mask(c1n, sel_sign);
        const uint32_t keep = oh8 & nz1; /* the arg-min edge's bit survives where its message c1 is not zero */
        const uint32_t oh[3] = { keep & ~(in1 | in2), keep & in1, keep & in2 };
#pragma unroll
        for (int g = 0; g < 3; ++g) out.x[g] &= ~z2 | oh[g];
    }
    out.cw = c2n | (c1n << 3) | (negA & 0x40404040u);
    out.pa[0] = pa[0] | (pa[1] << 16);
    out.pa[1] = pa[2] | (pa[3] << 16);
    return out;
}

#endif /* LNSFAID_SWAR_H */
