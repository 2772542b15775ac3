/*
 * dropin_bench.cpp — the decoder in the reference's OWN call shape, measured.
 *
 * The reference calls a decoder with ONE group of 32 frames per call from T pinned threads, each owning a CLDPC
 * (reference CSimulate.cpp:136-164, thread fan-out main.cpp:164-172).  INTEGRATION.md section 2 binds exactly that:
 * every CLDPC gets its own context (lnsfaid_create(..., max_groups 1)) and Decode_FAID() becomes
 * lnsfaid_decode(ctx, fixInput, 1, decodedBits, nullptr) on the object's plain malloc'd buffers.  This program is that
 * binding without the rest of the simulator: T host threads, one context each, `--calls` decode calls per thread on
 * pageable (default) or page-locked (--register: lnsfaid_host_register, what host/CLDPC.cpp does at Initial) buffers,
 * all on GPU `--device`.  It prints one JSON line: per-call latency and the aggregate information rate.
 *
 * Input: all-zero codeword, QPSK, AWGN, 4-bit quantiser with the formulas of CSimulate::Configure / Run
 * (CSimulate.cpp:69-74, :126-132) from a per-thread xorshift generator (the noise stream is not the reference's: what
 * matters here is the decoder's work per call, which depends on Eb/N0 only).
 */
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "lnsfaid.h"

struct Rng {
    uint64_t s;
    double uni() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return ((s >> 11) + 0.5) / 9007199254740992.0; }
    double gauss() { return std::sqrt(-2.0 * std::log(uni())) * std::cos(6.283185307179586 * uni()); }
};

int main(int argc, char** argv)
{
    int threads = 8, calls = 50, method = 2, max_iter = 10, device = 0, warmup = 3, groups = 1;
    double eb_n0 = 3.0, scale = 13.0;
    bool reg = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--calls") && i + 1 < argc) calls = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--method") && i + 1 < argc) method = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-iter") && i + 1 < argc) max_iter = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--eb-n0") && i + 1 < argc) eb_n0 = atof(argv[++i]);
        else if (!strcmp(argv[i], "--groups-per-call") && i + 1 < argc) groups = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--register")) reg = true;
        else { fprintf(stderr, "usage: %s [--threads T] [--calls C] [--warmup W] [--eb-n0 X] [--method M] [--max-iter I] [--groups-per-call G] [--register] [--device d]\n", argv[0]); return 2; }
    }
    if (threads < 1 || calls < 1 || groups < 1) return 2;

    static uint16_t pos_vn[70400];
    static int32_t deg[3], deg_rows[3];
    lnsfaid_code code;
    if (lnsfaid_code_50gpon(&code, pos_vn, deg, deg_rows)) return 1;
    lnsfaid_cfg cfg;
    if (lnsfaid_cfg_default(&cfg, method, max_iter)) return 1;
    const int N = code.n_var, K = N - code.n_check;
    const size_t group_bytes = (size_t)LNSFAID_GROUP * N, call_bytes = group_bytes * groups;
    const double sigma = 1.0 / std::sqrt(0.8444444 * 2.0 * std::pow(10.0, eb_n0 / 10.0)); /* CSimulate.cpp:73, QPSK */
    const double sigma_ch = sigma / std::sqrt(2.0);                                         /* CSimulate.cpp:126    */

    std::vector<lnsfaid_ctx*> ctx(threads, nullptr);
    std::vector<int8_t*> fix(threads, nullptr), out(threads, nullptr);
    std::vector<std::vector<double>> lat(threads);
    std::atomic<int> ready(0), failed(0);
    std::atomic<bool> go(false);
    std::vector<std::thread> pool;
    std::chrono::steady_clock::time_point t_start, t_end;
    std::vector<std::chrono::steady_clock::time_point> t_done(threads);

    for (int t = 0; t < threads; ++t) {
        pool.emplace_back([&, t]() {
            /* CLDPC::Initial of thread t (reference main.cpp:31-34: one CSimulate / CLDPC per thread) */
            int rc = lnsfaid_create(&ctx[t], &code, &cfg, device, (size_t)groups);
            fix[t] = (int8_t*)malloc(call_bytes);
            out[t] = (int8_t*)malloc(call_bytes);
            if (rc || !fix[t] || !out[t]) { fprintf(stderr, "thread %d: create failed: %s (%s)\n", t, lnsfaid_strerror(rc), lnsfaid_last_hip_error()); failed++; ready++; return; }
            if (reg && (lnsfaid_host_register(fix[t], call_bytes) || lnsfaid_host_register(out[t], call_bytes))) { failed++; ready++; return; }
            Rng rng{ 0x9e3779b97f4a7c15ull * (uint64_t)(t + 1) };
            for (size_t i = 0; i < call_bytes; ++i) {
                const double y = -0.707107 + sigma_ch * rng.gauss();
                double q = std::trunc(scale * y);
                fix[t][i] = (int8_t)(q > 7 ? 7 : (q < -7 ? -7 : q)); /* float2LimitChar_4bit, CLDPC.cpp:4553-4573 */
            }
            for (int w = 0; w < warmup; ++w)
                if (lnsfaid_decode(ctx[t], fix[t], (size_t)groups, out[t], nullptr)) { failed++; break; }
            ready++;
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            lat[t].reserve(calls);
            for (int c = 0; c < calls; ++c) {
                const auto a = std::chrono::steady_clock::now();
                rc = lnsfaid_decode(ctx[t], fix[t], (size_t)groups, out[t], nullptr);
                const auto b = std::chrono::steady_clock::now();
                if (rc) { fprintf(stderr, "thread %d call %d: %s (%s)\n", t, c, lnsfaid_strerror(rc), lnsfaid_last_hip_error()); failed++; break; }
                lat[t].push_back(std::chrono::duration<double, std::milli>(b - a).count());
            }
            t_done[t] = std::chrono::steady_clock::now();
        });
    }
    while (ready.load() < threads) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    t_start = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto& th : pool) th.join();
    t_end = *std::max_element(t_done.begin(), t_done.end());
    if (failed.load()) { fprintf(stderr, "%d thread(s) failed\n", failed.load()); return 1; }

    std::vector<double> all;
    for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    double sum = 0;
    for (double v : all) sum += v;
    const double wall_s = std::chrono::duration<double>(t_end - t_start).count();
    const double bits = (double)all.size() * groups * LNSFAID_GROUP * K;
    unsigned long ones = 0; /* keep the outputs alive */
    for (int t = 0; t < threads; ++t) for (size_t i = 0; i < call_bytes; i += 4099) ones += (unsigned long)out[t][i];
    printf("{\"threads\": %d, \"calls_per_thread\": %d, \"groups_per_call\": %d, \"host_buffers\": \"%s\", \"eb_n0_db\": %.2f, \"method\": %d, "
           "\"per_call_ms_mean\": %.4f, \"per_call_ms_p50\": %.4f, \"per_call_ms_p95\": %.4f, \"wall_s\": %.4f, \"aggregate_Gbps\": %.4f, \"check\": %lu}\n",
           threads, calls, groups, reg ? "registered" : "pageable", eb_n0, method, sum / all.size(), all[all.size() / 2],
           all[(size_t)(all.size() * 0.95)], wall_s, bits / wall_s / 1e9, ones);
    for (int t = 0; t < threads; ++t) {
        if (reg) { lnsfaid_host_unregister(fix[t]); lnsfaid_host_unregister(out[t]); }
        lnsfaid_destroy(ctx[t]);
        free(fix[t]); free(out[t]);
    }
    return 0;
}
