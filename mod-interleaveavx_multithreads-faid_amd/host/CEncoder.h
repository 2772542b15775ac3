/*
 * CEncoder.h — systematic encoder derived from the code table.
 *
 * The reference's CLDPC::Encode (reference CLDPC.cpp:68-155) walks `GenMatrix` (per parity row: weight, then that many
 * indices into the code bits already known, CLDPC.cpp:88-94), but the table is not shipped (reference
 * Constants_SSE.h:3106-3107 is empty, .MISSING_LARGE_BLOBS), so the reference can only send the fixed CodeWord_sym
 * (FakeEncoder).  The codeword of given information bits is nevertheless determined by H: with H = [A | B]
 * (A: M x K information part, B: M x M parity part) the parity bits solve B p = A u over GF(2).  This class inverts B
 * once by bit-packed Gauss-Jordan elimination and encodes 32 frames at a time, bit-sliced like the reference
 * (one 32-bit word per code bit, lane l = frame l).
 */
#ifndef CENCODER_H
#define CENCODER_H
#include <cstdint>
#include <iosfwd>
#include <vector>

class CEncoder {
public:
    /* row_deg[r], r < n_check: degree of check r; pos_vn: the PosNoeudsVariable table (row-major).  Returns false when
     * the parity part of H is singular (no systematic encoder with the information bits in front). */
    bool Initial(int n_var, int n_check, const int* row_deg, const uint16_t* pos_vn);
    /* inputBits [32][K] (0 / 1) -> outputBits [32][K] followed by [32][M]: the layout CLDPC::Encode leaves in
     * CLDPC::outputBits with no punctured / shortened bits (reference CLDPC.cpp:96-123) */
    void Encode32(const int8_t* inputBits, int8_t* outputBits) const;
    /* the reference's GenMatrix format (unsigned short stream: weight, indices..., one record per parity bit,
     * reference CLDPC.cpp:88-94); dense records over the information bits */
    void WriteGenMatrix(std::ostream& os) const;
    int K() const { return m_K; }

private:
    int m_N = 0, m_M = 0, m_K = 0;
    std::vector<uint32_t> m_row_start;  /* M + 1 offsets into m_info_cols                      */
    std::vector<uint16_t> m_info_cols;  /* information columns of every check row (A, sparse)  */
    std::vector<uint64_t> m_binv;       /* B^-1, M rows of M / 64 words                         */
};
#endif
