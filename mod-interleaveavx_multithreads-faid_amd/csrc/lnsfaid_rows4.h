/*
 * lnsfaid_rows4.h - what the four-rows-per-lane decode kernels share outside the layer step: the code-table view the layer step
 * reads, staging copies, the hard-decision / confidence bit planes, the live-progress words, output staging, the erasure plane of
 * EF_ELIMINATION 2.  Included by lnsfaid_kernel4.hip (one wave per codeword) and lnsfaid_kernel5.hip (two waves per codeword, where
 * LF_WG_SYNC() is redefined to the one-wave fence before this header is read: everything here runs on ONE wave).
 */
#ifndef LNSFAID_ROWS4_H
#define LNSFAID_ROWS4_H

#include "lnsfaid_device.h"
#include "lnsfaid_phases.h"
#include "lnsfaid_swar.h"

#define LF_T4 64

#define LF4_OMS(M) ((M) == 1 || (M) == 3 || (M) == 4)

struct DevTab4 {
    CCode c;
    int br;
    uint32_t sbv; /* lane j < 32: (block column * 256) << 16 | 4 * shift of edge j of this layer */
    __device__ __forceinline__ uint32_t sb(int j) const { return c->circ[br][j].sb; }
    /* the same split on the host (4 * shift, block column * 256): contiguous tables, so the 23 values of a layer arrive in a
     * few wide scalar loads and no scalar arithmetic is left per edge */
    __device__ __forceinline__ uint32_t s4(int j) const { return c->s4tab[br][j]; }
    __device__ __forceinline__ uint32_t cb256(int j) const { return c->cbtab[br][j]; }
    __device__ __forceinline__ uint32_t sb_dyn4(uint32_t idx4) const
    {
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx4, (int)sbv);
    }
};

typedef __attribute__((address_space(3))) uint32_t lds4_u32;
__device__ __forceinline__ uint32_t lds4_rd(uint32_t a) { return *(const lds4_u32*)(size_t)a; }
__device__ __forceinline__ void lds4_wr(uint32_t a, uint32_t v) { *(lds4_u32*)(size_t)a = v; }

/* n words from global memory into LDS, B loads per lane in flight at a time: with two waves per SIMD a "load, wait, store" loop
 * exposes one memory round trip per word */
template <int B>
__device__ __forceinline__ void copy_in(uint32_t* dst, const uint32_t* __restrict__ src, int n, int tid)
{
    for (int i0 = 0; i0 < n; i0 += B * LF_T4) {
        uint32_t w[B];
#pragma unroll
        for (int u = 0; u < B; ++u) { const int i = i0 + u * LF_T4 + tid; w[u] = i < n ? src[i] : 0u; }
#pragma unroll
        for (int u = 0; u < B; ++u) { const int i = i0 + u * LF_T4 + tid; if (i < n) dst[i] = w[u]; }
    }
}

/* n words from LDS to global memory (or LDS), B reads per lane in flight at a time */
template <int B>
__device__ __forceinline__ void copy_out(uint32_t* __restrict__ dst, const uint32_t* src, int n, int tid)
{
    for (int i0 = 0; i0 < n; i0 += B * LF_T4) {
        uint32_t w[B];
#pragma unroll
        for (int u = 0; u < B; ++u) { const int i = i0 + u * LF_T4 + tid; w[u] = i < n ? src[i] : 0u; }
#pragma unroll
        for (int u = 0; u < B; ++u) { const int i = i0 + u * LF_T4 + tid; if (i < n) dst[i] = w[u]; }
    }
}

/* hard decision En > 0 on the biased bytes (En + 120 >= 121): bit 7 of every byte of x + 7 */
__device__ __forceinline__ uint32_t hard_flags(uint32_t x) { return x + 0x07070707u; }

/* ---- bit plane from the interleaved En image: hard decision (CDecoder_FAID.cpp:299, :6416-6419) or, with CONF, the 2B1C
 * confidence bit |En| >= thr (CDecoder_FAID_2B1C.cpp:6132-6136).  Lane d holds variable nodes d, d + 64, d + 128, d + 192 of a
 * block column in one dword, and plane word (column, k, h) wants the flags of byte k of lanes 32 h .. 32 h + 31.  Eight columns
 * at a time: every lane collects its 8 x 4 flags in one word (bit 8 k + u: byte k of column cb0 + u), the two halves of the
 * wave transpose their 32 x 32 bit matrices in five exchange steps (lane ^ j for j = 16 .. 1: rows and columns swap bit j),
 * after which lane 8 k + u of half h holds plane word (cb0 + u, k, h).  About 60 instructions per eight columns, against
 * 32 ballots + 64 v_writelane_b32 before. */
template <int J>
__device__ __forceinline__ uint32_t plane_exchange(uint32_t x, uint32_t l5)
{
    constexpr uint32_t m_lo = J == 16 ? 0x0000ffffu : J == 8 ? 0x00ff00ffu : J == 4 ? 0x0f0f0f0fu : J == 2 ? 0x33333333u : 0x55555555u;
    uint32_t t;
    if (J == 1) t = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xb1, 0xf, 0xf, false);      /* quad_perm [1,0,3,2] */
    else if (J == 2) t = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4e, 0xf, 0xf, false); /* quad_perm [2,3,0,1] */
    else t = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x1f | (J << 10));               /* lane ^ J inside 32 lanes */
    const bool up = (l5 & (uint32_t)J) != 0u;
    /* rows with bit J clear keep their low columns and take the partner's low columns as their high ones, and vice versa */
    const uint32_t r = __builtin_amdgcn_alignbit(t, t, up ? (uint32_t)J : 32u - (uint32_t)J);
    const uint32_t keep = up ? ~m_lo : m_lo;
    return (x & keep) | (r & ~keep);
}

template <bool CONF>
__device__ __forceinline__ void build_plane4(CCode c, uint32_t* plane, int thr, int lane)
{
    /* (inlined on purpose: as a function of its own the column count is no longer known to be uniform, and every column
     * becomes a masked branch with an LDS round trip of its own) */
    const int nbc = c->nbc;
    const int th = thr < 1 ? 0 : (thr > 32 ? 32 : thr); /* |En| <= 31: a threshold above 31 means "never" */
    const uint32_t th4 = (uint32_t)th * 0x01010101u;
    const uint32_t b8 = (uint32_t)(128 - SW_BIAS_EN) * 0x01010101u, b7 = (uint32_t)(127 - SW_BIAS_EN) * 0x01010101u;
    const uint32_t l5 = (uint32_t)lane & 31u;
    const int word_of_lane = (int)((l5 & 7u) * 8u + 2u * (l5 >> 3) + ((uint32_t)lane >> 5)); /* + cb0 * 8 */
    for (int cb0 = 0; cb0 < nbc; cb0 += 8) {
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) /* columns beyond the last one re-read it; their words are not stored */
            x[u] = lds4_rd((uint32_t)(cb0 + u < nbc ? cb0 + u : nbc - 1) * 256u + 4u * (uint32_t)lane);
        __builtin_amdgcn_sched_barrier(0); /* the eight reads in flight together */
        uint32_t g = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t fl;
            /* En >= thr  <=>  Eb + 8 - thr >= 128;  En <= -thr  <=>  Eb + 7 + thr < 128 (no carries: Eb in [89, 151]) */
            if (CONF) fl = th == 0 ? 0x80808080u : (((x[u] + b8) - th4) | ~(x[u] + b7 + th4));
            else fl = hard_flags(x[u]);
            g |= (fl >> (7 - u)) & (0x01010101u << u); /* bit 7 of byte k -> bit 8 k + u */
        }
        g = plane_exchange<16>(g, l5);
        g = plane_exchange<8>(g, l5);
        g = plane_exchange<4>(g, l5);
        g = plane_exchange<2>(g, l5);
        g = plane_exchange<1>(g, l5);
        if (cb0 + (int)(l5 & 7u) < nbc) plane[cb0 * 8 + word_of_lane] = g;
    }
    LF_WG_SYNC();
}

/* ---- cheap "certainly dirty" test (DecodeMethod 2, see lnsfaid_kernels.hip): parity of the lane's four rows of layer 0
 * straight from En; the XOR of the hard-decision flags is bit 7 of the XOR of the flag words. */
__device__ __forceinline__ bool layer0_dirty4(CCode c, int lane)
{
    const int deg = c->deg[0];
    const uint32_t tid4 = (uint32_t)lane << 2;
    uint32_t x4[LF_MAX_DEG], d[LF_MAX_DEG];
    /* all addresses, then all reads, then the arithmetic: one LDS round trip instead of one per circulant */
#pragma unroll
    for (int j = 0; j < LF_MAX_DEG; ++j) x4[j] = j < deg ? tid4 + c->s4tab[0][j] : 0u;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < LF_MAX_DEG; ++j)
        if (j < deg) d[j] = lds4_rd((x4[j] & 0xfcu) | c->cbtab[0][j]);
    __builtin_amdgcn_sched_barrier(0);
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < LF_MAX_DEG; ++j)
        if (j < deg) acc ^= hard_flags(__builtin_amdgcn_alignbyte(d[j], d[j], x4[j] >> 8));
    return __ballot((acc & 0x80808080u) != 0u) != 0ull;
}

/* ---- live progress of the group -----------------------------------------------------------------
 * A codeword may pass decision point t once it is PROVEN that its group does not stop there.  The snapshot of the previous
 * launch gives such proofs (a lane parked beyond t); this gives more of them while the launch runs: every codeword
 * publishes the point it is about to pass (agent-scope store, monotonic), and whoever passed t first must have been dirty
 * at t, so "some lane of my group has passed t" proves that the group goes on.  A clean codeword looks once, never waits:
 * without a proof it parks exactly as before, so results do not depend on timing, only the number of relaunches does.
 * MEASURED (profiles/r02_kernel4/live_progress.txt): bit-exact, one launch fewer, but not faster - the same iterations are
 * executed either way (SQ_INSTS_VALU 9.55 G against 9.68 G per batch at 3.6 dB) and the codewords that find no proof leave a
 * thin, long second launch (41 Gb/s against 51 Gb/s at 3.6 dB, 102 Gb/s either way at 4.2 dB).  Built only with
 * -DLF4_LIVE_PROOF; what is kept from the experiment is the speculative output of a parking codeword (below). */
__device__ __forceinline__ void publish_pass(int32_t* live, int cw, int t, int tid)
{
#ifdef LF4_LIVE_PROOF
    if (tid == 0) __hip_atomic_store(&live[cw], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ bool group_passed(const int32_t* live, int g, int t, int tid)
{
#ifndef LF4_LIVE_PROOF
    return false;
#else
    int v = 0;
    if (tid < LNSFAID_GROUP) v = __hip_atomic_load(&live[g * LNSFAID_GROUP + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __ballot(v >= t) != 0ull;
#endif
}

/* decodedBits of this codeword from the hard-decision plane (CDecoder_FAID.cpp:7091-7102, CDecoder_OMS.cpp:2966-2967): one plane
 * word = 32 output bytes per lane and round, the plane words of all rounds read before the first store */
__device__ __forceinline__ void write_decoded(const uint32_t* sHard, int8_t* g_out, int N, int tid)
{
    const int nw = N >> 5;
    if (((size_t)g_out) & 15u) { /* caller's buffer not 16-byte aligned: dword stores */
        uint32_t* out32 = (uint32_t*)g_out;
        for (int i = tid; i < (N >> 2); i += LF_T4) out32[i] = (((sHard[i >> 3] >> ((i & 7) * 4)) & 15u) * 0x00204081u) & 0x01010101u;
        return;
    }
    uint4* out = (uint4*)g_out;
    for (int r0 = 0; r0 * LF_T4 < nw; r0 += 9) {
        uint32_t w[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) { const int i = (r0 + u) * LF_T4 + tid; w[u] = i < nw ? sHard[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = (r0 + u) * LF_T4 + tid;
            if (i < nw) {
                uint32_t d[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) d[q] = (((w[u] >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u; /* bit k -> byte k */
                out[2 * i] = make_uint4(d[0], d[1], d[2], d[3]);
                out[2 * i + 1] = make_uint4(d[4], d[5], d[6], d[7]);
            }
        }
    }
}

/* ---- EF_ELIMINATION 2: bit plane "every check of this variable node is unsatisfied" over the block columns of weight W
 * (flip_vote[v] >= REGULAR_COL_WEIGHT, CDecoder_FAID.cpp:306-309, :675), bit-sliced like the flip decision: per (column,
 * 64-node window) the AND of the W rotated windows of the parity plane.  Written over the hard-decision plane, which is dead
 * between the syndrome stage and the next one. */
__device__ void build_erasure_plane4(CCode c, const LfDevCode* gc, uint32_t* plane, const uint32_t* sP, int W, int tid)
{
    const int units = c->n_wcols * 4;
    for (int u = tid; u < units; u += LF_T4) {
        const int cb = gc->wcol[u >> 2];
        const uint32_t win = (uint32_t)(u & 3);
        uint32_t lo = 0xffffffffu, hi = 0xffffffffu;
        for (int k = 0; k < W; ++k) {
            const uint32_t cc = gc->colcirc[cb][k];
            uint32_t a, b;
            window64(sP + (cc & 0xffu) * 8u, (64u * win - ((cc >> 8) & 0xffu)) & 255u, a, b);
            lo &= a; hi &= b;
        }
        plane[cb * 8 + 2 * (int)win] = lo;
        plane[cb * 8 + 2 * (int)win + 1] = hi;
    }
    LF_WG_SYNC();
}

#endif /* LNSFAID_ROWS4_H */
