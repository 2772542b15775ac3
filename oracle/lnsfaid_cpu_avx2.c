/*
 * lnsfaid_cpu_avx2.c — a vectorised CPU port of the decode path: 32 codewords in the int8 lanes of one
 * 256-bit register, like the reference (CLDPC.h:21), written with AVX2 intrinsics only.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as lnsfaid_oracle.c).  Purpose: bench.py's `cpu_baseline` leg.  The
 * scalar oracle (lnsfaid_oracle.c) is the statement-by-statement restatement that is pinned to the
 * reference; this file is a faster implementation of the same arithmetic, accepted only because
 * tests/test_oracle.py::test_avx2_port_equals_oracle compares it bit for bit with the oracle.  It differs from
 * the reference's AVX-512 code in two ways that matter for speed, not for results: the FAID look-up table
 * is one byte shuffle instead of nine masked adds (CDecoder_FAID.cpp:710-851), and the dead `flip_vote`
 * accumulation of the syndrome stage (CDecoder_FAID.cpp:306-309, only used with EF_ELIMINATION 2) is not done.
 */
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

#include "lnsfaid_oracle.h"

typedef __m256i V;
#define L 32
#define MAXDEG 64

struct lnsfaid_cpu {
    lnsfaid_code code;
    lnsfaid_cfg cfg;
    uint16_t* pos;
    int32_t* row_deg;
    int8_t* vn_weight;
    V* En;      /* [n_var]   */
    V* Lmn;     /* [n_edges] */
    V* chk;     /* [n_check] byte mask 0 / -1 per lane: l_checksum_ */
    V *hard, *hard2, *hard_ch, *flip, *vote; /* [n_var] byte masks / counts */
};

static inline V set1(int a) { return _mm256_set1_epi8((char)a); }
static inline V gt(V a, V b) { return _mm256_cmpgt_epi8(a, b); }
static inline V eq(V a, V b) { return _mm256_cmpeq_epi8(a, b); }
static inline V sel(V mask, V a, V b) { return _mm256_blendv_epi8(b, a, mask); } /* mask ? a : b */

int lnsfaid_cpu_create(lnsfaid_cpu** out, const lnsfaid_code* code, const lnsfaid_cfg* cfg)
{
    if (!out || !code || !cfg || !code->pos_vn) return LNSFAID_E_INVAL;
    if (cfg->decode_method < 0 || cfg->decode_method > 5) return LNSFAID_E_INVAL;
    lnsfaid_cpu* o = (lnsfaid_cpu*)calloc(1, sizeof(*o));
    if (!o) return LNSFAID_E_NOMEM;
    o->code = *code;
    o->cfg = *cfg;
    const size_t E = (size_t)code->n_edges, N = (size_t)code->n_var, M = (size_t)code->n_check;
    o->pos = (uint16_t*)malloc(E * sizeof(uint16_t));
    o->row_deg = (int32_t*)malloc(M * sizeof(int32_t));
    o->vn_weight = (int8_t*)calloc(N, 1);
    o->En = (V*)aligned_alloc(32, N * sizeof(V));
    o->Lmn = (V*)aligned_alloc(32, E * sizeof(V));
    o->chk = (V*)aligned_alloc(32, M * sizeof(V));
    o->hard = (V*)aligned_alloc(32, N * sizeof(V));
    o->hard2 = (V*)aligned_alloc(32, N * sizeof(V));
    o->hard_ch = (V*)aligned_alloc(32, N * sizeof(V));
    o->flip = (V*)aligned_alloc(32, N * sizeof(V));
    o->vote = (V*)aligned_alloc(32, N * sizeof(V));
    if (!o->pos || !o->row_deg || !o->vn_weight || !o->En || !o->Lmn || !o->chk || !o->hard || !o->hard2 || !o->hard_ch
        || !o->flip || !o->vote) { lnsfaid_cpu_destroy(o); return LNSFAID_E_NOMEM; }
    memcpy(o->pos, code->pos_vn, E * sizeof(uint16_t));
    size_t r = 0, e = 0;
    for (int k = 0; k < code->nb_degres; ++k)
        for (int i = 0; i < code->deg_rows[k]; ++i) {
            if (r >= M || code->deg[k] < 1 || code->deg[k] > MAXDEG) { lnsfaid_cpu_destroy(o); return LNSFAID_E_CODE; }
            o->row_deg[r++] = code->deg[k];
            e += (size_t)code->deg[k];
        }
    if (r != M || e != E) { lnsfaid_cpu_destroy(o); return LNSFAID_E_CODE; }
    for (size_t i = 0; i < E; ++i) o->vn_weight[o->pos[i]]++;
    *out = o;
    return LNSFAID_OK;
}

void lnsfaid_cpu_destroy(lnsfaid_cpu* o)
{
    if (!o) return;
    free(o->pos); free(o->row_deg); free(o->vn_weight); free(o->En); free(o->Lmn); free(o->chk); free(o->hard);
    free(o->hard2); free(o->hard_ch); free(o->flip); free(o->vote);
    free(o);
}

static inline int wclass(int w) { return w == 3 ? 0 : (w == 6 ? 1 : (w == 11 ? 2 : 3)); }

/* 8-entry byte table, duplicated in both 128-bit halves for _mm256_shuffle_epi8 */
static inline V lut_vec(const int8_t t[8])
{
    return _mm256_setr_epi8(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[7], t[7], t[7], t[7], t[7], t[7], t[7], t[7],
                            t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[7], t[7], t[7], t[7], t[7], t[7], t[7], t[7]);
}

static V oms_off(V x, int window, V F, V f1, V f2, V one)
{
    if (window) {
        V k = _mm256_and_si256(F, gt(f2, x));
        x = _mm256_add_epi8(x, _mm256_and_si256(k, one));
        k = _mm256_and_si256(F, _mm256_or_si256(gt(f1, x), eq(x, f1)));
        x = _mm256_add_epi8(x, _mm256_and_si256(k, one));
        k = _mm256_andnot_si256(F, gt(x, f1));
        x = _mm256_sub_epi8(x, _mm256_and_si256(k, one));
        k = _mm256_andnot_si256(F, _mm256_or_si256(gt(x, f2), eq(x, f2)));
        x = _mm256_sub_epi8(x, _mm256_and_si256(k, one));
    } else {
        V k = gt(x, f1);
        x = _mm256_sub_epi8(x, _mm256_and_si256(k, one));
        k = _mm256_or_si256(gt(x, f2), eq(x, f2));
        x = _mm256_sub_epi8(x, _mm256_and_si256(k, one));
    }
    return x;
}

/* (zero-extended v * factor) >> 5 in 16-bit lanes, signed-saturating pack, limit to 7 (CLDPC.cpp:337-352) */
static V nms_scale(V v, int factor, V v7)
{
    const V f = _mm256_set1_epi16((short)factor), z = _mm256_setzero_si256();
    V lo = _mm256_srli_epi16(_mm256_mullo_epi16(_mm256_unpacklo_epi8(v, z), f), 5);
    V hi = _mm256_srli_epi16(_mm256_mullo_epi16(_mm256_unpackhi_epi8(v, z), f), 5);
    return _mm256_min_epi8(_mm256_packs_epi16(lo, hi), v7);
}

static void decode_group(lnsfaid_cpu* o, const int8_t* fix, int8_t* out, lnsfaid_group_stats* st)
{
    const lnsfaid_cfg* c = &o->cfg;
    const int N = o->code.n_var, M = o->code.n_check, K = N - M, E = o->code.n_edges;
    const int oms = c->decode_method == 1 || c->decode_method == 3 || c->decode_method == 4, ef = c->ef_elimination >= 1;
    const int with_bf = c->decode_method != 1 && c->decode_method != 0;
    const int nms = c->decode_method == 0;
    const V zero = _mm256_setzero_si256(), one = set1(1), ones = set1(-1);
    const V vmin = set1(-31), vmax = set1(31), v7 = set1(7), sbit = set1((char)0x80);
    const V f1 = set1(c->factor_1), f2 = set1(c->factor_2);

    /* staging: [32][K] | [32][M] -> lane-major, tail erase, Lmn = 0 */
    {
        int8_t* en = (int8_t*)o->En;
        for (int l = 0; l < L; ++l) {
            const int8_t* si = fix + (size_t)l * K;
            const int8_t* sp = fix + (size_t)L * K + (size_t)l * M;
            for (int v = 0; v < K; ++v) en[(size_t)v * L + l] = si[v];
            for (int j = 0; j < M; ++j) en[(size_t)(K + j) * L + l] = sp[j];
        }
        for (int i = 0; i < o->code.puncture_tail; ++i) o->En[N - 1 - i] = zero;
        memset(o->Lmn, 0, (size_t)E * sizeof(V));
    }

    int executed = 0;
    for (int itn = 0; nms && itn < c->max_iteration; ++itn) { /* CLDPC::Decode: no syndrome stage, fixed iterations */
        size_t e = 0;
        for (int r = 0; r < M; ++r) {
            const int deg = o->row_deg[r];
            V tv[MAXDEG];
            V sign = zero, min1 = vmax, min2 = vmax;
            for (int j = 0; j < deg; ++j) {
                const V t = _mm256_max_epi8(_mm256_subs_epi8(o->En[o->pos[e + j]], o->Lmn[e + j]), vmin);
                const V a = _mm256_abs_epi8(t);
                tv[j] = t;
                sign = _mm256_xor_si256(sign, _mm256_and_si256(t, sbit));
                min2 = _mm256_min_epi8(min2, _mm256_max_epi8(min1, a));
                min1 = _mm256_min_epi8(min1, a);
            }
            const V c2 = nms_scale(min1, c->factor_1, v7), c1 = nms_scale(min2, c->factor_2, v7);
            if (deg & 1) sign = _mm256_xor_si256(sign, sbit);
            for (int j = 0; j < deg; ++j) {
                const V mag = sel(eq(_mm256_abs_epi8(tv[j]), min1), c1, c2);
                const V neg = _mm256_xor_si256(sign, _mm256_and_si256(tv[j], sbit));
                const V l2 = sel(neg, _mm256_sub_epi8(zero, mag), mag);
                o->Lmn[e + j] = l2;
                o->En[o->pos[e + j]] = _mm256_min_epi8(_mm256_max_epi8(_mm256_adds_epi8(tv[j], l2), vmin), vmax);
            }
            e += (size_t)deg;
        }
        executed++;
    }
    for (int rem = c->max_iteration - 1; rem >= 0 && !nms; --rem) {
        /* syndrome stage */
        V esum = zero;
        const uint16_t* p = o->pos;
        const uint16_t* pv = o->pos;
        const int erase = (c->ef_elimination == 2); /* EF_ELIMINATION 2: votes per VN from the syndrome stage, era_ flags */
        if (erase) for (int i = 0; i < N; ++i) { o->vote[i] = zero; o->flip[i] = zero; }
        for (int r = 0; r < M; ++r) {
            V par = zero;
            for (int j = 0; j < o->row_deg[r]; ++j) par = _mm256_xor_si256(par, gt(o->En[*p++], zero));
            o->chk[r] = par;
            esum = oms ? _mm256_adds_epu8(esum, _mm256_and_si256(par, one)) : _mm256_adds_epi8(esum, _mm256_and_si256(par, one));
            if (erase) for (int j = 0; j < o->row_deg[r]; ++j, ++pv) o->vote[*pv] = _mm256_adds_epu8(o->vote[*pv], _mm256_and_si256(par, one));
        }
        if (_mm256_testz_si256(esum, esum)) break;
        V lme;
        if (oms) { /* unsigned esum < (uint8) floor_err_count */
            const V fc = set1((uint8_t)c->floor_err_count);
            lme = _mm256_andnot_si256(eq(_mm256_max_epu8(esum, fc), esum), ones);
        } else {
            lme = gt(set1((int8_t)c->floor_err_count), esum);
        }
        const int it = c->max_iteration - rem;
        const int itx = (it >= 1 && it <= 5) ? it - 1 : 5;
        const int window = rem <= c->floor_iter_thresh;
        V lut[4], lute[4];
        for (int w = 0; w < 4; ++w) { lut[w] = lut_vec(c->v2c_map[itx][w]); lute[w] = lut_vec(c->v2c_map_ef[itx][w]); }

        size_t e = 0;
        for (int r = 0; r < M; ++r) {
            const int deg = o->row_deg[r];
            V tv[MAXDEG], mv[MAXDEG], sv[MAXDEG];
            V sign = zero, min1 = vmax, min2 = vmax;
            const V efm = (ef && window) ? _mm256_and_si256(lme, o->chk[r]) : zero;
            for (int j = 0; j < deg; ++j) {
                const int col = o->pos[e + j];
                const V en = o->En[col];
                V t = _mm256_max_epi8(_mm256_subs_epi8(en, o->Lmn[e + j]), vmin);
                V s, m;
                if (oms) {
                    s = _mm256_and_si256(t, sbit);
                    m = _mm256_min_epi8(_mm256_abs_epi8(t), v7);
                } else {
                    t = _mm256_min_epi8(t, vmax);
                    if (erase && window && o->vn_weight[col] == c->regular_col_weight) { /* CDecoder_FAID.cpp:673-680 */
                        const V wv = set1((char)c->regular_col_weight);
                        const V ge = _mm256_or_si256(gt(o->vote[col], wv), eq(o->vote[col], wv));
                        const V mask = _mm256_andnot_si256(o->flip[col], _mm256_and_si256(ge, lme));
                        t = _mm256_andnot_si256(mask, t);
                        o->flip[col] = _mm256_or_si256(o->flip[col], mask);
                    }
                    s = _mm256_and_si256(sel(eq(t, zero), en, t), sbit); /* sign back-track */
                    const V a = _mm256_min_epi8(_mm256_abs_epi8(t), v7);
                    const int w = wclass(o->vn_weight[col]);
                    m = _mm256_shuffle_epi8(lut[w], a);
                    if (ef && window) m = sel(efm, _mm256_shuffle_epi8(lute[w], a), m);
                }
                tv[j] = t; mv[j] = m; sv[j] = s;
                sign = _mm256_xor_si256(sign, s);
                min2 = _mm256_min_epi8(min2, _mm256_max_epi8(min1, m));
                min1 = _mm256_min_epi8(min1, m);
            }
            V c1, c2;
            if (oms) {
                const V F = _mm256_and_si256(o->chk[r], lme);
                c1 = _mm256_min_epi8(oms_off(min2, window, F, f1, f2, one), v7);
                c2 = _mm256_min_epi8(oms_off(min1, window, F, f1, f2, one), v7);
            } else {
                c1 = _mm256_min_epi8(min2, v7);
                c2 = _mm256_min_epi8(min1, v7);
            }
            if (deg & 1) sign = _mm256_xor_si256(sign, sbit);
            for (int j = 0; j < deg; ++j) {
                /* OMS compares the un-clamped |t| with min1, FAID the mapped value */
                const V key = oms ? _mm256_abs_epi8(tv[j]) : mv[j];
                const V mag = sel(eq(key, min1), c1, c2);
                const V neg = _mm256_xor_si256(sign, sv[j]); /* bit 7 set: negative */
                const V l2 = sel(neg, _mm256_sub_epi8(zero, mag), mag); /* blendv looks at bit 7 */
                o->Lmn[e + j] = l2;
                o->En[o->pos[e + j]] = _mm256_min_epi8(_mm256_max_epi8(_mm256_adds_epi8(tv[j], l2), vmin), vmax);
            }
            e += (size_t)deg;
        }
        executed++;
    }

    int bf = 0;
    if (c->decode_method == 3) { /* plain bit flipping, CDecoder_OMSBF.cpp:2959-3517 */
        for (int i = 0; i < N; ++i) { o->hard[i] = gt(o->En[i], zero); o->flip[i] = zero; }
        while (bf < c->max_bf_iter) {
            for (int i = 0; i < N; ++i) o->vote[i] = zero;
            V esum = zero, maxv = one;
            const uint16_t* p = o->pos;
            const uint16_t* p2 = o->pos;
            for (int r = 0; r < M; ++r) {
                V par = zero;
                for (int j = 0; j < o->row_deg[r]; ++j) par = _mm256_xor_si256(par, o->hard[*p++]);
                const V inc = _mm256_and_si256(par, one);
                esum = _mm256_adds_epu8(esum, inc);
                for (int j = 0; j < o->row_deg[r]; ++j, ++p2) {
                    o->vote[*p2] = _mm256_adds_epu8(o->vote[*p2], inc);
                    maxv = _mm256_max_epi8(maxv, o->vote[*p2]);
                }
            }
            if (_mm256_testz_si256(esum, esum)) break;
            const V thr = _mm256_min_epi8(maxv, set1(c->bf_vote_cap));
            for (int v = 0; v < N; ++v)
                if (o->vn_weight[v] > 0) o->flip[v] = _mm256_or_si256(gt(o->vote[v], thr), eq(o->vote[v], thr));
            for (int i = 0; i < N; ++i) o->hard[i] = _mm256_xor_si256(o->hard[i], o->flip[i]);
            bf++;
        }
    } else if (with_bf) {
        const int W = c->regular_col_weight, two_bit = c->decode_method == 5;
        const V thr = set1(c->hard2_threshold), nthr = set1(-c->hard2_threshold);
        for (int i = 0; i < N; ++i) {
            o->hard[i] = gt(o->En[i], zero);
            o->hard_ch[i] = o->hard[i];
            o->hard2[i] = two_bit ? _mm256_or_si256(_mm256_or_si256(gt(o->En[i], thr), eq(o->En[i], thr)),
                                                    _mm256_or_si256(gt(nthr, o->En[i]), eq(o->En[i], nthr))) : zero;
            o->flip[i] = zero;
        }
        V t = ones, Th = set1(W), l0 = zero, l1 = zero;
        const V L0 = set1(c->bf_L0), L1 = set1(c->bf_L1), alpha = set1(c->bf_alpha), delta = set1(c->bf_delta);
        while (bf < c->max_bf_iter) {
            for (int i = 0; i < N; ++i) o->vote[i] = zero;
            V esum = zero;
            const uint16_t* p = o->pos;
            const uint16_t* p2 = o->pos;
            for (int r = 0; r < M; ++r) {
                V par = zero;
                for (int j = 0; j < o->row_deg[r]; ++j) par = _mm256_xor_si256(par, o->hard[*p++]);
                const V inc = _mm256_and_si256(par, one);
                esum = _mm256_adds_epu8(esum, inc);
                for (int j = 0; j < o->row_deg[r]; ++j, ++p2) o->vote[*p2] = _mm256_adds_epu8(o->vote[*p2], inc);
            }
            if (_mm256_testz_si256(esum, esum)) break;
            /* threshold state machine (CDecoder_FAID.cpp:6787-6799) */
            Th = sel(t, Th, _mm256_subs_epi8(Th, delta));
            const V maxTh = _mm256_and_si256(t, gt(L0, l0));
            Th = sel(maxTh, set1(W + c->bf_alpha), Th);
            l0 = _mm256_adds_epi8(l0, _mm256_and_si256(maxTh, one));
            const V sub = _mm256_and_si256(_mm256_andnot_si256(maxTh, t), gt(L1, l1));
            Th = sel(sub, set1(W + c->bf_alpha - c->bf_delta), Th);
            l1 = _mm256_adds_epi8(l1, _mm256_and_si256(sub, one));
            const V ssub = _mm256_andnot_si256(sub, _mm256_andnot_si256(maxTh, t));
            Th = sel(ssub, set1(W + c->bf_alpha - 2 * c->bf_delta), Th);
            Th = _mm256_max_epi8(Th, one);
            t = zero;
            for (int v = 0; v < N; ++v) {
                if (o->vn_weight[v] != W) continue;
                const V flipped = _mm256_xor_si256(o->hard[v], o->hard_ch[v]);
                const V x = _mm256_adds_epi8(o->vote[v], _mm256_and_si256(flipped, alpha));
                const V m = _mm256_or_si256(gt(x, Th), eq(x, Th));
                o->flip[v] = m;
                t = _mm256_or_si256(t, m);
            }
            if (!two_bit) {
                for (int i = 0; i < N; ++i) o->hard[i] = _mm256_xor_si256(o->hard[i], o->flip[i]);
            } else {
                const V big = _mm256_or_si256(gt(Th, set1(W)), eq(Th, set1(W)));
                for (int i = 0; i < N; ++i) {
                    const V x3 = _mm256_and_si256(big, o->flip[i]);
                    V h = _mm256_xor_si256(o->hard[i], x3), h2 = _mm256_xor_si256(o->hard2[i], x3);
                    const V sm = _mm256_andnot_si256(big, o->flip[i]);
                    h = _mm256_xor_si256(h, _mm256_andnot_si256(h2, sm));
                    h2 = _mm256_xor_si256(h2, _mm256_and_si256(sm, h2));
                    o->hard[i] = h; o->hard2[i] = h2;
                }
            }
            bf++;
        }
    }
    /* output: [32][N] 0 / 1 */
    {
        int8_t tmp[L];
        for (int v = 0; v < N; ++v) {
            const V b = !with_bf ? _mm256_and_si256(gt(o->En[v], zero), one) : _mm256_and_si256(o->hard[v], one);
            _mm256_storeu_si256((V*)tmp, b);
            for (int l = 0; l < L; ++l) out[(size_t)l * N + v] = tmp[l];
        }
    }
    if (st) { st->iterations = executed; st->bf_iterations = bf; }
}

int lnsfaid_cpu_decode(lnsfaid_cpu* o, const int8_t* fixInput, size_t n_groups, int8_t* decodedBits, lnsfaid_group_stats* stats)
{
    if (!o || (n_groups && (!fixInput || !decodedBits))) return LNSFAID_E_INVAL;
    const size_t stride = (size_t)L * (size_t)o->code.n_var;
    for (size_t g = 0; g < n_groups; ++g) decode_group(o, fixInput + g * stride, decodedBits + g * stride, stats ? stats + g : NULL);
    return LNSFAID_OK;
}
