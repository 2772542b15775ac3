#include "CEncoder.h"

#include <cstring>
#include <ostream>

bool CEncoder::Initial(int n_var, int n_check, const int* row_deg, const uint16_t* pos_vn)
{
    m_N = n_var; m_M = n_check; m_K = n_var - n_check;
    const int M = m_M, K = m_K, W = (M + 63) / 64;
    m_row_start.assign(M + 1, 0);
    m_info_cols.clear();
    /* [B | I], bit-packed rows of 2 W words */
    std::vector<uint64_t> aug((size_t)M * 2 * W, 0);
    size_t e = 0;
    for (int r = 0; r < M; ++r) {
        for (int j = 0; j < row_deg[r]; ++j) {
            const int v = pos_vn[e + j];
            if (v < K) m_info_cols.push_back((uint16_t)v);
            else aug[(size_t)r * 2 * W + (size_t)((v - K) >> 6)] ^= 1ull << ((v - K) & 63);
        }
        e += (size_t)row_deg[r];
        m_row_start[r + 1] = (uint32_t)m_info_cols.size();
        aug[(size_t)r * 2 * W + W + (size_t)(r >> 6)] |= 1ull << (r & 63);
    }
    /* A's rows were appended in the order of the checks; the elimination permutes rows of [B | I] only, which is fine:
     * the right half ends up as B^-1 whatever the row order was. */
    std::vector<uint64_t> tmp(2 * W);
    for (int col = 0; col < M; ++col) {
        const size_t w = (size_t)(col >> 6);
        const uint64_t bit = 1ull << (col & 63);
        int piv = -1;
        for (int r = col; r < M; ++r)
            if (aug[(size_t)r * 2 * W + w] & bit) { piv = r; break; }
        if (piv < 0) return false;
        if (piv != col) {
            memcpy(tmp.data(), &aug[(size_t)col * 2 * W], sizeof(uint64_t) * 2 * W);
            memcpy(&aug[(size_t)col * 2 * W], &aug[(size_t)piv * 2 * W], sizeof(uint64_t) * 2 * W);
            memcpy(&aug[(size_t)piv * 2 * W], tmp.data(), sizeof(uint64_t) * 2 * W);
        }
        const uint64_t* prow = &aug[(size_t)col * 2 * W];
        for (int r = 0; r < M; ++r) {
            if (r == col || !(aug[(size_t)r * 2 * W + w] & bit)) continue;
            uint64_t* row = &aug[(size_t)r * 2 * W];
            for (int x = (int)w; x < 2 * W; ++x) row[x] ^= prow[x]; /* words left of the pivot are already zero in prow */
        }
    }
    m_binv.assign((size_t)M * W, 0);
    for (int r = 0; r < M; ++r) memcpy(&m_binv[(size_t)r * W], &aug[(size_t)r * 2 * W + W], sizeof(uint64_t) * W);
    return true;
}

void CEncoder::Encode32(const int8_t* inputBits, int8_t* outputBits) const
{
    const int M = m_M, K = m_K, W = (M + 63) / 64;
    /* bit-slice: U[v] bit l = information bit v of frame l (the reference's uchar_transpose_avx, CLDPC.cpp:75-86) */
    std::vector<uint32_t> U(K, 0), S(M, 0);
    for (int l = 0; l < 32; ++l)
        for (int v = 0; v < K; ++v) U[v] |= (uint32_t)(inputBits[(size_t)l * K + v] & 1) << l;
    for (int r = 0; r < M; ++r) { /* s = A u */
        uint32_t acc = 0;
        for (uint32_t x = m_row_start[r]; x < m_row_start[r + 1]; ++x) acc ^= U[m_info_cols[x]];
        S[r] = acc;
    }
    /* p = B^-1 s with the method of the four Russians: all 256 XOR combinations of every 8 consecutive s words */
    const int blocks = (M + 7) / 8;
    std::vector<uint32_t> T((size_t)blocks * 256);
    for (int b = 0; b < blocks; ++b) {
        uint32_t* t = &T[(size_t)b * 256];
        t[0] = 0;
        for (int x = 1; x < 256; ++x) {
            const int low = x & -x, i = __builtin_ctz((unsigned)x);
            const int idx = b * 8 + i;
            t[x] = t[x ^ low] ^ (idx < M ? S[idx] : 0u);
        }
    }
    memcpy(outputBits, inputBits, (size_t)32 * K); /* systematic part: [32][K] */
    int8_t* par = outputBits + (size_t)32 * K;     /* [32][M] */
    for (int i = 0; i < M; ++i) {
        const uint8_t* row = (const uint8_t*)&m_binv[(size_t)i * W]; /* little endian: byte b holds columns 8b .. 8b+7 */
        uint32_t p = 0;
        for (int b = 0; b < blocks; ++b) p ^= T[(size_t)b * 256 + row[b]];
        for (int l = 0; l < 32; ++l) par[(size_t)l * M + i] = (int8_t)((p >> l) & 1u);
    }
}

void CEncoder::WriteGenMatrix(std::ostream& os) const
{
    /* parity i = XOR of the information bits j with (B^-1 A)[i][j] = 1 */
    const int M = m_M, K = m_K, W = (M + 63) / 64;
    std::vector<uint8_t> g(K);
    for (int i = 0; i < M; ++i) {
        std::fill(g.begin(), g.end(), 0);
        for (int r = 0; r < M; ++r)
            if (m_binv[(size_t)i * W + (size_t)(r >> 6)] >> (r & 63) & 1)
                for (uint32_t x = m_row_start[r]; x < m_row_start[r + 1]; ++x) g[m_info_cols[x]] ^= 1;
        int weight = 0;
        for (int j = 0; j < K; ++j) weight += g[j];
        os << weight << ",";
        for (int j = 0; j < K; ++j) if (g[j]) os << j << ",";
        os << "\n";
    }
}
