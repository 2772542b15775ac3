/*
 * swar_emul.cpp — TEST INFRASTRUCTURE: runs the product's 4-rows-per-lane layer step (csrc/lnsfaid_swar.h, the very
 * statements the HIP kernel executes, compiled for the host with the ISA semantics of v_perm_b32 / v_alignbyte_b32 /
 * v_bitop3_b32 restated in that header) lane by lane on the CPU, so that tests/ can compare it with the oracle's
 * a-posteriori LLRs after every layered iteration without a GPU.  Never part of the product path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "lnsfaid.h"
#include "../mod-interleaveavx_multithreads-faid_amd/csrc/lnsfaid_swar.h"

namespace {
struct HostTab {
    const uint32_t* row; /* sb of the layer's edges */
    uint32_t sb(int j) const { return row[j]; }
    uint32_t s4(int j) const { return (row[j] & 255u) << 2; }
    uint32_t cb256(int j) const { return row[j] & ~255u; }
    uint32_t sb_dyn4(uint32_t j4) const { const uint32_t v = row[j4 >> 2]; return ((v & ~255u) << 16) | ((v & 255u) << 2); }
};

template <int METHOD>
SwRow step(int deg, bool spec, const SwLds& lds, const HostTab& tab, const SwParams& p, uint32_t lane, SwRow cur, bool fresh,
           uint32_t rowpar, bool lme, bool erase = false, uint32_t era_edges = 0, uint32_t era_plane = 0)
{
    const SwK K = sw_consts();
    if (erase) {
        if constexpr (METHOD == 2) return sw_layer_step<METHOD, 0, true>(lds, tab, p, K, lane, deg, cur, fresh, rowpar, lme, era_edges, era_plane);
    }
    if (spec && deg == 23) return sw_layer_step<METHOD, 23>(lds, tab, p, K, lane, deg, cur, fresh, rowpar, lme);
    if (spec && deg == 22) return sw_layer_step<METHOD, 22>(lds, tab, p, K, lane, deg, cur, fresh, rowpar, lme);
    return sw_layer_step<METHOD, 0>(lds, tab, p, K, lane, deg, cur, fresh, rowpar, lme);
}
}

/* n_iter layered iterations (no early stop) of every codeword of ONE group; en_out[l * N + v] = En of lane l.
 * specialised != 0 uses the compile-time-degree instances for degrees 22 / 23 (what the kernel runs on the 50G-PON code). */
extern "C" int swar_emul_layered(const lnsfaid_code* code, const lnsfaid_cfg* cfg, const int8_t* fixInput, int n_iter,
                                 int specialised, int8_t* en_out)
{
    const int N = code->n_var, M = code->n_check, K = N - M, Z = code->z;
    if (Z != 256) return LNSFAID_E_CODE;
    const int nbr = M / Z;
    std::vector<int> deg(nbr);
    std::vector<uint32_t> sb((size_t)nbr * SW_MAX_DEG, 0u);
    {
        size_t e = 0; int r = 0;
        std::vector<int> row_deg;
        for (int k = 0; k < code->nb_degres; ++k) for (int i = 0; i < code->deg_rows[k]; ++i) row_deg.push_back(code->deg[k]);
        for (int br = 0; br < nbr; ++br) {
            deg[br] = row_deg[(size_t)br * Z];
            if (deg[br] > SW_MAX_DEG) return LNSFAID_E_CODE;
            for (int j = 0; j < deg[br]; ++j) sb[(size_t)br * SW_MAX_DEG + j] = code->pos_vn[e + j];
            e += (size_t)deg[br] * Z; r += Z;
        }
    }
    const int method = cfg->decode_method;
    const bool oms = SW_OMS(method);
    for (int l = 0; l < 32; ++l) {
        const uint32_t plane_off = ((uint32_t)N + 15u) & ~15u; /* where the kernel keeps its hard / erasure plane */
        std::vector<uint8_t> img((size_t)plane_off + (size_t)N / 8 + 8, 0);
        for (int v = 0; v < N; ++v) {
            int llr = v < K ? fixInput[(size_t)l * K + v] : fixInput[(size_t)32 * K + (size_t)l * M + (v - K)];
            if (v >= N - code->puncture_tail) llr = 0;
            img[sw_en_pos((uint32_t)v)] = (uint8_t)(llr + SW_BIAS_EN);
        }
        SwLds lds; lds.base = img.data();
        std::vector<SwRow> rows((size_t)nbr * 64);
        memset(rows.data(), 0, rows.size() * sizeof(SwRow));
        for (int it = 1; it <= n_iter; ++it) {
            /* syndrome stage: parity of every row and the (saturated) number of unsatisfied rows */
            std::vector<uint8_t> par(M);
            int unsat = 0;
            {
                size_t e = 0;
                for (int r = 0; r < M; ++r) {
                    const int d = deg[r / Z];
                    int p = 0;
                    for (int j = 0; j < d; ++j) p ^= ((int)img[sw_en_pos(code->pos_vn[e + j])] - SW_BIAS_EN) > 0;
                    par[r] = (uint8_t)p; unsat += p; e += d;
                }
            }
            const int rem = cfg->max_iteration - it;
            bool lme;
            if (oms) lme = (unsat > 255 ? 255 : unsat) < (int)(uint8_t)cfg->floor_err_count;
            else lme = (unsat > 127 ? 127 : unsat) < (int)(int8_t)cfg->floor_err_count;
            const int itx = (it >= 1 && it <= 5) ? it - 1 : 5;
            SwParams p;
            p.lut_lo = p.lut_hi = p.ef_lo = p.ef_hi = 0;
            for (int a = 0; a < 8; ++a) {
                (a < 4 ? p.lut_lo : p.lut_hi) |= (uint32_t)(uint8_t)cfg->v2c_map[itx][0][a] << (8 * (a & 3));
                (a < 4 ? p.ef_lo : p.ef_hi) |= (uint32_t)(uint8_t)cfg->v2c_map_ef[itx][0][a] << (8 * (a & 3));
            }
            p.f1 = (int8_t)cfg->factor_1; p.f2 = (int8_t)cfg->factor_2;
            p.window = rem <= cfg->floor_iter_thresh;
            p.ef_tables = cfg->ef_elimination >= 1;
            sw_oms_tables(p);
            if (method == 0) {
                if (!sw_nms_fits((int16_t)cfg->factor_1, (int16_t)cfg->factor_2)) return LNSFAID_E_INVAL; /* (the kernel would not run it) */
                p.f1 = p.f2 = (int16_t)cfg->factor_1;
                sw_nms_tables(p.f1, p.nms_t);
            }
            /* EF_ELIMINATION 2: erase in this iteration?  plane bit v = all checks of v unsatisfied (weight-W columns) */
            const int W = cfg->regular_col_weight;
            const bool erase = method == 2 && cfg->ef_elimination == 2 && p.window && lme;
            std::vector<uint32_t> era_edges(nbr, 0u);
            if (erase) {
                std::vector<int> votes(N, 0), weight(N, 0);
                size_t e = 0;
                for (int r = 0; r < M; ++r) { const int d = deg[r / Z]; for (int j = 0; j < d; ++j) { votes[code->pos_vn[e + j]] += par[r]; weight[code->pos_vn[e + j]]++; } e += d; }
                memset(img.data() + plane_off, 0, (size_t)N / 8);
                for (int v = 0; v < N; ++v)
                    if (weight[v] == W && votes[v] >= W) img[plane_off + (size_t)(v >> 3)] |= (uint8_t)(1u << (v & 7));
                std::vector<char> seen(N / Z, 0);
                for (int br = 0; br < nbr; ++br)
                    for (int j = 0; j < deg[br]; ++j) {
                        const int cb = (int)(sb[(size_t)br * SW_MAX_DEG + j] / (uint32_t)Z);
                        if (weight[(size_t)cb * Z] == W && !seen[cb]) { seen[cb] = 1; era_edges[br] |= 1u << j; }
                    }
            }
            for (int br = 0; br < nbr; ++br) {
                HostTab tab; tab.row = &sb[(size_t)br * SW_MAX_DEG];
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    uint32_t rowpar = 0;
                    for (int k = 0; k < 4; ++k) if (par[(size_t)br * Z + lane + 64 * k]) rowpar |= 0xffu << (8 * k);
                    SwRow& cur = rows[(size_t)br * 64 + lane];
                    const bool fresh = (it == 1);
                    if (method == 2) cur = step<2>(deg[br], specialised != 0, lds, tab, p, lane, cur, fresh, rowpar, lme, erase, era_edges[br], plane_off);
                    else if (method == 5) cur = step<5>(deg[br], specialised != 0, lds, tab, p, lane, cur, fresh, rowpar, lme);
                    else if (oms) cur = step<1>(deg[br], specialised != 0, lds, tab, p, lane, cur, fresh, rowpar, lme);
                    else if (method == 0) cur = step<0>(deg[br], specialised != 0, lds, tab, p, lane, cur, fresh, rowpar, lme);
                    else return LNSFAID_E_INVAL;
                }
            }
        }
        for (int v = 0; v < N; ++v) en_out[(size_t)l * N + v] = (int8_t)((int)img[sw_en_pos((uint32_t)v)] - SW_BIAS_EN);
    }
    return LNSFAID_OK;
}

/* host restatements of the header's building blocks, for tests/test_gpu_swar.py */
extern "C" void swar_emul_ops(int n, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* out /* [8][n] */)
{
    for (int i = 0; i < n; ++i) {
        out[0 * n + i] = sw_perm(a[i], b[i], c[i]);
        out[1 * n + i] = sw_alignbyte(a[i], b[i], c[i] % 5u);
        out[2 * n + i] = sw_bitop3<SW_TT_SEL>(a[i], b[i], c[i]);
        out[3 * n + i] = sw_bitop3<SW_TT_NANDOR>(a[i], b[i], c[i]);
        out[4 * n + i] = sw_mask7(a[i], SW_SEL_SIGN);
        out[5 * n + i] = sw_bitop3<SW_TT_A_AND_BORC>(a[i], b[i], c[i]);
        out[6 * n + i] = sw_bitop3<SW_TT_BFI_C>(a[i], b[i], c[i]);
        out[7 * n + i] = sw_bitop3<SW_TT_XORAND>(a[i], b[i], c[i]);
    }
}
