/* Self-test of CEncoder on the host (no GPU): derives the encoder from the code table, encodes 32 random frames and checks
 * every parity check of every frame; also checks that the systematic part is the input.  Used by tests/test_host_encoder.py. */
#include "CEncoder.h"
#include "Constants/Constants_SSE.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main() {
    std::vector<int> row_deg;
    const int deg[] = { DEG_1, DEG_2, DEG_3 }, rows[] = { DEG_1_COMPUTATIONS, DEG_2_COMPUTATIONS, DEG_3_COMPUTATIONS };
    for (int k = 0; k < 3; ++k) row_deg.insert(row_deg.end(), rows[k], deg[k]);
    CEncoder e;
    auto t0 = std::chrono::steady_clock::now();
    bool ok = e.Initial(_NoVar, _NoCheck, row_deg.data(), PosNoeudsVariable);
    auto t1 = std::chrono::steady_clock::now();
    printf("init ok=%d %.2f s\n", ok, std::chrono::duration<double>(t1 - t0).count());
    const int K = _NoVar - _NoCheck, N = _NoVar, M = _NoCheck;
    std::vector<int8_t> in(32 * K), out(32 * N);
    for (auto& b : in) b = rand() % 2;
    t0 = std::chrono::steady_clock::now();
    e.Encode32(in.data(), out.data());
    t1 = std::chrono::steady_clock::now();
    printf("encode32 %.4f s\n", std::chrono::duration<double>(t1 - t0).count());
    // syndrome check
    int bad = 0; size_t ed = 0;
    for (int r = 0; r < M; ++r) { for (int l = 0; l < 32; ++l) { int p = 0; for (int j = 0; j < row_deg[r]; ++j) { int v = PosNoeudsVariable[ed + j]; p ^= v < K ? out[(size_t)l * K + v] : out[(size_t)32 * K + (size_t)l * M + (v - K)]; } bad += p; } ed += row_deg[r]; }
    printf("unsatisfied checks over 32 frames: %d\n", bad);
    int diff = 0;
    for (int i = 0; i < 32 * K; ++i) diff += in[i] != out[i];
    printf("systematic part differs in %d positions\n", diff);
    return (bad != 0 || diff != 0 || !ok) ? 1 : 0;
}
