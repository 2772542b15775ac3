// Micro-benchmark: VALU issue rate of ONE SIMD as a function of resident waves per SIMD and of the independent chains inside a
// wave (gfx950).  Single-wave workgroups; dynamic LDS size sets the residency (160 KB per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 4000
template <int ILP, int MIX>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t seed)
{
    extern __shared__ uint32_t dyn[];
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (2 * i + 3) + threadIdx.x;
    uint32_t b = seed * 31 + 1, c = seed ^ 0x12345;
    for (int r = 0; r < REP; ++r) {
        // 8 instructions per iteration; chain c uses register a[c % ILP]: ILP independent dependency chains
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t& x = a[i % ILP];
            if (MIX == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            else if (MIX == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(b), "v"(c));
            else if (MIX == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            else { // 3 full-rate : 1 half-rate, like the byte-parallel layer step
                if (i % 4 == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
                else asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(b), "v"(c));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (dyn[0] == 0x12345678u) s ^= 1; // keep the LDS allocation
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <typename K> void run(K kern, const char* name, uint32_t* d, int waves_per_simd, double ghz)
{
    const size_t lds = waves_per_simd >= 16 ? 0 : (size_t)(160 * 1024 / (4 * waves_per_simd)) - 512; // waves per CU = 4 x waves per SIMD
    const int blocks = 256 * 4 * waves_per_simd * 4; // 4 rounds
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const size_t use = lds > 65536 ? 65536 : lds;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), use, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), use, 0, d, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)blocks / 1024.0 * REP * 8;
    printf("%-28s %2d waves/SIMD (LDS %6zu B/wave)  %7.3f ms  %5.2f SIMD-cycles per wave-instruction\n", name, waves_per_simd, use, ms,
           ms * 1e-3 * ghz * 1e9 / inst_per_simd);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    printf("%s  CUs %d  clock %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    uint32_t* d; hipMalloc(&d, 256 * 4 * 16 * 4 * 64 * 4);
    for (int w : {1, 2, 3, 4, 8}) {
        if (w == 3) continue; // 160 KB / 12 is not reachable with the 64 KB per-workgroup cap only for w = 1, 2: those use 64 KB (1 or 2 fit)
        run(k<1, 1>, "bitop3, 1 chain", d, w, ghz);
        run(k<2, 1>, "bitop3, 2 chains", d, w, ghz);
        run(k<4, 1>, "bitop3, 4 chains", d, w, ghz);
        run(k<8, 1>, "bitop3, 8 chains", d, w, ghz);
        run(k<1, 2>, "perm, 1 chain", d, w, ghz);
        run(k<4, 2>, "perm, 4 chains", d, w, ghz);
        run(k<4, 3>, "3 bitop3 : 1 perm, 4 chains", d, w, ghz);
        run(k<8, 3>, "3 bitop3 : 1 perm, 8 chains", d, w, ghz);
        run(k<4, 0>, "add, 4 chains", d, w, ghz);
    }
    return 0;
}
