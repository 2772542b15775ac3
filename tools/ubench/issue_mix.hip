// Micro-benchmark: SIMD cycles per wave64 instruction of the byte-parallel layer step's VALU mix (70 % full-rate class: v_bitop3_b32,
// two-source add / shift; 30 % half-rate class: v_perm_b32, v_alignbyte_b32) as a function of the waves resident per SIMD.
// Unlike issue_rate.hip the loop body is long - BODY straight-line instructions per trip (template recursion, no per-instruction
// control flow) - so the loop's own scalar instructions and branch do not count; CHAINS independent dependency chains per wave,
// as in the layer step's stage-by-stage order.  Single-wave workgroups; the dynamic LDS size sets the residency.
// build: hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define TRIPS 400
#define BODY 480

template <int I, int CHAINS>
__device__ __forceinline__ void body(uint32_t (&a)[8], uint32_t b, uint32_t c)
{
    if constexpr (I < BODY) {
        uint32_t& x = a[I % CHAINS];
        constexpr int m = I % 10; // add, shl, bitop3, perm, bitop3, add, alignbyte, bitop3, perm, bitop3
        if constexpr (m == 3 || m == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        else if constexpr (m == 6) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        else if constexpr (m == 0 || m == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        else if constexpr (m == 1) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
        else asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(b), "v"(c));
        body<I + 1, CHAINS>(a, b, c);
    }
}

// one instruction class per kernel: what each costs at the decode kernel's residency (two waves per SIMD)
// 0 v_add_u32 (two sources)  1 v_bitop3_b32  2 v_perm_b32  3 v_alignbyte_b32  4 v_add3_u32  5 v_and_or_b32  6 v_add_u32 with an SGPR source
// 7 v_and_or_b32 with an SGPR source  8 v_lshlrev_b32 (inline constant)  9 v_and_b32 with a 32-bit literal
// 10 v_mov_b32_sdwa (byte 1 of the source)  11 v_or_b32_sdwa (byte 0 of one source)  12 the same with an SGPR source  13 v_lshrrev_b32 by 8
// 14 v_lshlrev_b32 by 8  15 v_lshrrev_b32 by 1  16 / 17 the shifts by a register  18 v_sub_u32
template <int I, int OP>
__device__ __forceinline__ void body_one(uint32_t (&a)[8], uint32_t b, uint32_t c, uint32_t sg)
{
    if constexpr (I < BODY) {
        uint32_t& x = a[I % 4];
        if constexpr (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        else if constexpr (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(b), "v"(c));
        else if constexpr (OP == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        else if constexpr (OP == 3) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        else if constexpr (OP == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        else if constexpr (OP == 5) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        else if constexpr (OP == 6) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "s"(sg));
        else if constexpr (OP == 7) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(sg));
        else if constexpr (OP == 8) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
        else if constexpr (OP == 10) asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(x));
        else if constexpr (OP == 11) asm volatile("v_or_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(x) : "v"(b));
        else if constexpr (OP == 12) asm volatile("v_or_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(x) : "s"(sg));
        else if constexpr (OP == 13) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(x));
        else if constexpr (OP == 14) asm volatile("v_lshlrev_b32 %0, 8, %0" : "+v"(x));
        else if constexpr (OP == 15) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x));
        else if constexpr (OP == 16) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(b));
        else if constexpr (OP == 17) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(b));
        else if constexpr (OP == 18) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        else asm volatile("v_and_b32 %0, 0x87878787, %0" : "+v"(x));
        body_one<I + 1, OP>(a, b, c, sg);
    }
}

template <int OP>
__global__ __launch_bounds__(64) void k_one(uint32_t* out, uint32_t seed)
{
    extern __shared__ uint32_t dyn[];
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (2 * i + 3) + threadIdx.x;
    uint32_t b = seed * 31 + 1, c = seed ^ 0x12345;
    const uint32_t sg = seed * 7;
    for (int r = 0; r < TRIPS; ++r) body_one<0, OP>(a, b, c, sg);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (dyn[0] == 0x12345678u) s ^= 1;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t seed)
{
    extern __shared__ uint32_t dyn[];
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (2 * i + 3) + threadIdx.x;
    uint32_t b = seed * 31 + 1, c = seed ^ 0x12345;
    for (int r = 0; r < TRIPS; ++r) body<0, CHAINS>(a, b, c);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (dyn[0] == 0x12345678u) s ^= 1; // keep the LDS allocation
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// one wave ALONE on its SIMD: 64 KB of LDS per single-wave workgroup, so a CU holds two of them (on two of its four SIMDs)
template <typename K> void run_alone(K kern, const char* name, uint32_t* d, double ghz)
{
    const size_t lds = 65536 - 512;
    const int rounds = 4, blocks = 256 * 2 * rounds;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, d, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, d, 2u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s  1 wave alone on its SIMD (2 per CU)      %8.3f ms  %5.2f SIMD-cycles per wave-instruction\n", name, ms,
           ms * 1e-3 * ghz * 1e9 / ((double)rounds * TRIPS * BODY));
}

template <typename K> void run(K kern, const char* name, uint32_t* d, int waves_per_simd, double ghz)
{
    // waves per CU = 4 x waves per SIMD: dynamic LDS such that exactly that many single-wave workgroups fit (160 KB per CU)
    const size_t lds = (size_t)(160 * 1024 / (4 * waves_per_simd)) - 512;
    const int rounds = 4;
    const int blocks = 256 * 4 * waves_per_simd * rounds;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, d, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, d, 2u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)blocks / 1024.0 * TRIPS * BODY;
    printf("%-10s %2d waves/SIMD (LDS %6zu B/wave)  %8.3f ms  %5.2f SIMD-cycles per wave-instruction\n", name, waves_per_simd, lds, ms,
           ms * 1e-3 * ghz * 1e9 / inst_per_simd);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    printf("%s  CUs %d  clock %.2f GHz; mix: 7 full-rate : 3 half-rate VALU, %d straight-line instructions per loop trip\n", p.name,
           p.multiProcessorCount, ghz, BODY);
    uint32_t* d;
    (void)hipMalloc(&d, (size_t)256 * 4 * 8 * 4 * 64 * 4);
    run_alone(k<4>, "4 chains", d, ghz);
    run_alone(k<8>, "8 chains", d, ghz);
    for (int w : { 2, 3, 4, 5, 8 }) {
        run(k<4>, "4 chains", d, w, ghz);
        run(k<8>, "8 chains", d, w, ghz);
    }
    printf("one instruction class per kernel, four chains:\n");
    for (int w : { 2, 8 }) {
        run(k_one<0>, "add", d, w, ghz);
        run(k_one<8>, "shl 1", d, w, ghz);
        run(k_one<9>, "and lit", d, w, ghz);
        run(k_one<1>, "bitop3", d, w, ghz);
        run(k_one<2>, "perm", d, w, ghz);
        run(k_one<3>, "alignbyte", d, w, ghz);
        run(k_one<4>, "add3", d, w, ghz);
        run(k_one<5>, "and_or", d, w, ghz);
        run(k_one<6>, "add sgpr", d, w, ghz);
        run(k_one<7>, "and_or sg", d, w, ghz);
        run(k_one<13>, "shr 8", d, w, ghz);
        run(k_one<14>, "shl 8", d, w, ghz);
        run(k_one<15>, "shr 1", d, w, ghz);
        run(k_one<16>, "shl vgpr", d, w, ghz);
        run(k_one<17>, "shr vgpr", d, w, ghz);
        run(k_one<18>, "sub", d, w, ghz);
        run(k_one<10>, "mov sdwa", d, w, ghz);
        run(k_one<11>, "or sdwa", d, w, ghz);
        run(k_one<12>, "or sdwa sg", d, w, ghz);
    }
    run_alone(k_one<0>, "add", d, ghz);
    run_alone(k_one<1>, "bitop3", d, ghz);
    run_alone(k_one<2>, "perm", d, ghz);
    return 0;
}
