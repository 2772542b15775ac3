/* Sanitizer self-test of the CPU-side code (oracle/Makefile target "sanitize"): oracle, AVX2 port, channel restatement and
 * table code on two groups of every DecodeMethod; exit code = number of methods whose port differs from the oracle. */
#include "lnsfaid_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
/* sanitizer run of the oracle + AVX2 port + front-end on a few groups, all methods (CPU build only) */
int lnsfaid_code_50gpon(lnsfaid_code*, uint16_t*, int32_t*, int32_t*);
int lnsfaid_cfg_default(lnsfaid_cfg*, int32_t, int32_t);
int main(void) {
    static uint16_t pos[70400]; int32_t deg[3], rows[3]; lnsfaid_code code;
    if (lnsfaid_code_50gpon(&code, pos, deg, rows)) return 2;
    const int N = code.n_var;
    int8_t* fix = malloc(2 * 32 * N); int8_t* out = malloc(2 * 32 * N); int8_t* out2 = malloc(2 * 32 * N);
    lnsfaid_frontend fe; lnsfaid_frontend_seed(&fe, 101);
    float sigma = lnsfaid_frontend_sigma(3.5f, 2, 0.8444444);
    for (int g = 0; g < 2; ++g) lnsfaid_frontend_qpsk_group(&fe, N, code.n_check, 0, sigma, 13.0f, fix + (size_t)g * 32 * N);
    int bad = 0;
    for (int m = 0; m <= 5; ++m) {
        lnsfaid_cfg cfg; lnsfaid_cfg_default(&cfg, m, 10);
        if (m == 0) { cfg.factor_1 = 24; cfg.factor_2 = 26; }
        lnsfaid_oracle* o = 0; lnsfaid_cpu* a = 0;
        if (lnsfaid_oracle_create(&o, &code, &cfg) || lnsfaid_cpu_create(&a, &code, &cfg)) { printf("create failed %d\n", m); return 3; }
        lnsfaid_group_stats st[2], st2[2];
        lnsfaid_oracle_decode(o, fix, 2, out, st); lnsfaid_cpu_decode(a, fix, 2, out2, st2);
        int d = memcmp(out, out2, 2 * 32 * N) != 0 || memcmp(st, st2, sizeof st) != 0;
        printf("method %d: I/J %d/%d %d/%d  port %s\n", m, st[0].iterations, st[0].bf_iterations, st[1].iterations, st[1].bf_iterations, d ? "DIFFERS" : "equal");
        bad += d;
        lnsfaid_oracle_destroy(o); lnsfaid_cpu_destroy(a);
    }
    free(fix); free(out); free(out2);
    return bad;
}
