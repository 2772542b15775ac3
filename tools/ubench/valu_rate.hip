// Micro-benchmark: issue rate of the VALU / LDS instructions the decode kernel is built from (gfx950).
// Each kernel runs REP x 64 independent instances of one instruction per wave; 16 waves per CU.
// Prints SIMD cycles per wave-instruction assuming the shader clock reported by the runtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP 2000
#define OPS8(x) x x x x x x x x
#define DEF_KERNEL(name, asm_line)                                              \
    __global__ void name(uint32_t* out, uint32_t seed) {                        \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t b = seed * 31 + 1, c = seed ^ 0x12345;                          \
        for (int i = 0; i < REP; ++i) {                                          \
            asm volatile(asm_line : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
    }
#define L8(op, tail) op " %0, %0, " tail "\n" op " %1, %1, " tail "\n" op " %2, %2, " tail "\n" op " %3, %3, " tail "\n" \
                     op " %4, %4, " tail "\n" op " %5, %5, " tail "\n" op " %6, %6, " tail "\n" op " %7, %7, " tail "\n"
#define L8S(op, src0) op " %0, " src0 ", %0\n" op " %1, " src0 ", %1\n" op " %2, " src0 ", %2\n" op " %3, " src0 ", %3\n" \
                      op " %4, " src0 ", %4\n" op " %5, " src0 ", %5\n" op " %6, " src0 ", %6\n" op " %7, " src0 ", %7\n"
// is the cost the opcode class or the encoding size?  4-byte VOP2, VOP2 + 32-bit literal (8 bytes), VOP2 forced to the
// 8-byte VOP3 encoding, VOP2 with an inline constant / an SGPR (4 bytes)
DEF_KERNEL(k_and_lit, L8S("v_and_b32", "0x10001"))
DEF_KERNEL(k_and_inl, L8S("v_and_b32", "15"))
DEF_KERNEL(k_and_sgpr, L8S("v_and_b32", "s2"))
DEF_KERNEL(k_xor_lit128, L8S("v_xor_b32", "0x80"))
DEF_KERNEL(k_add_e64, L8("v_add_u32_e64", "%8"))
DEF_KERNEL(k_lshr_inl, L8S("v_lshrrev_b32", "3"))
DEF_KERNEL(k_pk_min_inl, L8("v_pk_min_i16", "31 op_sel_hi:[1,0]"))
__global__ void k_lshr_b64(uint32_t* out, uint32_t seed) {
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t sh = (seed & 31) + 1;
    for (int i = 0; i < REP; ++i) {
        asm volatile("v_lshrrev_b64 %0, %4, %0\nv_lshrrev_b64 %1, %4, %1\nv_lshrrev_b64 %2, %4, %2\nv_lshrrev_b64 %3, %4, %3\n"
                     "v_lshlrev_b64 %0, %4, %0\nv_lshlrev_b64 %1, %4, %1\nv_lshlrev_b64 %2, %4, %2\nv_lshlrev_b64 %3, %4, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3);
}
DEF_KERNEL(k_mul_lo, L8("v_mul_lo_u32", "%8"))
DEF_KERNEL(k_bitop3, L8("v_bitop3_b32", "%8, %9 bitop3:0x96"))
DEF_KERNEL(k_dot4, L8("v_dot4_u32_u8", "%8, %9"))
DEF_KERNEL(k_add_u32, L8("v_add_u32", "%8"))
DEF_KERNEL(k_xor, L8("v_xor_b32", "%8"))
DEF_KERNEL(k_pk_add_u16, L8("v_pk_add_u16", "%8"))
DEF_KERNEL(k_pk_sub_i16, L8("v_pk_sub_i16", "%8"))
DEF_KERNEL(k_pk_min_i16, L8("v_pk_min_i16", "%8"))
DEF_KERNEL(k_pk_max_u16, L8("v_pk_max_u16", "%8"))
DEF_KERNEL(k_pk_lshl, L8("v_pk_lshlrev_b16", "%8"))
DEF_KERNEL(k_pk_ashr, L8("v_pk_ashrrev_i16", "%8"))
DEF_KERNEL(k_pk_mad, L8("v_pk_mad_u16", "%8, %9"))
DEF_KERNEL(k_perm, L8("v_perm_b32", "%8, %9"))
DEF_KERNEL(k_and_or, L8("v_and_or_b32", "%8, %9"))
DEF_KERNEL(k_lshl_or, L8("v_lshl_or_b32", "%8, %9"))
DEF_KERNEL(k_bfi, L8("v_bfi_b32", "%8, %9"))
DEF_KERNEL(k_med3, L8("v_med3_i32", "%8, %9"))
DEF_KERNEL(k_cndmask, L8("v_cndmask_b32", "%8, vcc"))
// dependent chain of packed ops with the compiler-style s_nop in between vs none
DEF_KERNEL(k_pk_dep_nop, "v_pk_add_u16 %0, %0, %8\ns_nop 0\nv_pk_sub_i16 %0, %0, %9\ns_nop 0\nv_pk_add_u16 %0, %0, %8\ns_nop 0\nv_pk_sub_i16 %0, %0, %9\ns_nop 0\n"
                          "v_pk_add_u16 %0, %0, %8\ns_nop 0\nv_pk_sub_i16 %0, %0, %9\ns_nop 0\nv_pk_add_u16 %0, %0, %8\ns_nop 0\nv_pk_sub_i16 %0, %0, %9\ns_nop 0\n")
DEF_KERNEL(k_pk_dep, "v_pk_add_u16 %0, %0, %8\nv_pk_sub_i16 %0, %0, %9\nv_pk_add_u16 %0, %0, %8\nv_pk_sub_i16 %0, %0, %9\n"
                      "v_pk_add_u16 %0, %0, %8\nv_pk_sub_i16 %0, %0, %9\nv_pk_add_u16 %0, %0, %8\nv_pk_sub_i16 %0, %0, %9\n")
DEF_KERNEL(k_add_dep, "v_add_u32 %0, %0, %8\nv_xor_b32 %0, %0, %9\nv_add_u32 %0, %0, %8\nv_xor_b32 %0, %0, %9\n"
                       "v_add_u32 %0, %0, %8\nv_xor_b32 %0, %0, %9\nv_add_u32 %0, %0, %8\nv_xor_b32 %0, %0, %9\n")

__global__ void k_lds_read_i8(uint32_t* out, uint32_t seed) {
    __shared__ int8_t s[20480];
    for (int i = threadIdx.x; i < 20480; i += blockDim.x) s[i] = (int8_t)(i * 7 + seed);
    __syncthreads();
    uint32_t ad = (threadIdx.x * 1 + seed) & 16383; int acc = 0;
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += s[(ad + k * 256) & 16383];
        ad = (ad + 37) & 16383;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_lds_write_b8(uint32_t* out, uint32_t seed) {
    __shared__ int8_t s[20480];
    uint32_t ad = (threadIdx.x * 1 + seed) & 16383;
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[(ad + k * 256) & 16383] = (int8_t)(i + k);
        ad = (ad + 37) & 16383;
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[threadIdx.x];
}

static size_t g_dyn_lds = 0; /* dynamic LDS per block: 40 KB with 256-thread blocks caps residency at 4 waves per SIMD, as in the decode kernel */
template <typename T> __global__ void k_lds_write_w(uint32_t* out, uint32_t seed) {
    __shared__ T s[20480 / sizeof(T)];
    uint32_t ad = (threadIdx.x + seed) & (16384 / sizeof(T) - 1);
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[(ad + k * (256 / sizeof(T))) & (16384 / sizeof(T) - 1)] = (T)(i + k);
        ad = (ad + 37) & (16384 / sizeof(T) - 1);
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s[threadIdx.x];
}
template <typename T> __global__ void k_lds_read_w(uint32_t* out, uint32_t seed) {
    __shared__ T s[20480 / sizeof(T)];
    for (int i = threadIdx.x; i < (int)(20480 / sizeof(T)); i += blockDim.x) s[i] = (T)(i * 7 + seed);
    __syncthreads();
    uint32_t ad = (threadIdx.x + seed) & (16384 / sizeof(T) - 1); uint32_t acc = 0;
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += (uint32_t)s[(ad + k * (256 / sizeof(T))) & (16384 / sizeof(T) - 1)];
        ad = (ad + 37) & (16384 / sizeof(T) - 1);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename K> double run(K kern, const char* name, uint32_t* d, int blocks, int threads, double ghz, int inst_per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), g_dyn_lds, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), g_dyn_lds, 0, d, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = blocks*threads/64 / (256 CUs * 4 SIMDs)
    double waves_per_simd = (double)blocks * threads / 64.0 / 1024.0;
    double inst_per_simd = waves_per_simd * (double)REP * inst_per_iter;
    double cyc = ms * 1e-3 * ghz * 1e9 / inst_per_simd;
    printf("%-16s %8.3f ms  %6.2f SIMD-cycles per wave-instruction (at %.2f GHz, %g waves/SIMD)\n", name, ms, cyc, ghz, waves_per_simd);
    return cyc;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    double ghz = p.clockRate / 1e6;
    printf("%s  CUs %d  clock %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    uint32_t* d; hipMalloc(&d, 4096 * 256 * 4);
    const int B = 4096, T = 256; // 16 waves/CU x 4 rounds
    run(k_add_u32, "v_add_u32", d, B, T, ghz, 8); run(k_xor, "v_xor_b32", d, B, T, ghz, 8);
    run(k_and_lit, "v_and lit32", d, B, T, ghz, 8); run(k_and_inl, "v_and inline", d, B, T, ghz, 8); run(k_and_sgpr, "v_and sgpr", d, B, T, ghz, 8);
    run(k_xor_lit128, "v_xor lit 0x80", d, B, T, ghz, 8); run(k_add_e64, "v_add_u32_e64", d, B, T, ghz, 8); run(k_lshr_inl, "v_lshrrev inl", d, B, T, ghz, 8);
    run(k_pk_min_inl, "pk_min inline", d, B, T, ghz, 8);
    run(k_lshr_b64, "v_lsh{l,r}rev_b64", d, B, T, ghz, 8); run(k_mul_lo, "v_mul_lo_u32", d, B, T, ghz, 8);
    run(k_bitop3, "v_bitop3_b32", d, B, T, ghz, 8); run(k_dot4, "v_dot4_u32_u8", d, B, T, ghz, 8);
    run(k_pk_add_u16, "v_pk_add_u16", d, B, T, ghz, 8); run(k_pk_sub_i16, "v_pk_sub_i16", d, B, T, ghz, 8);
    run(k_pk_min_i16, "v_pk_min_i16", d, B, T, ghz, 8); run(k_pk_max_u16, "v_pk_max_u16", d, B, T, ghz, 8);
    run(k_pk_lshl, "v_pk_lshlrev_b16", d, B, T, ghz, 8); run(k_pk_ashr, "v_pk_ashrrev_i16", d, B, T, ghz, 8);
    run(k_pk_mad, "v_pk_mad_u16", d, B, T, ghz, 8); run(k_perm, "v_perm_b32", d, B, T, ghz, 8);
    run(k_and_or, "v_and_or_b32", d, B, T, ghz, 8); run(k_lshl_or, "v_lshl_or_b32", d, B, T, ghz, 8);
    run(k_bfi, "v_bfi_b32", d, B, T, ghz, 8); run(k_med3, "v_med3_i32", d, B, T, ghz, 8); run(k_cndmask, "v_cndmask_b32", d, B, T, ghz, 8);
    run(k_pk_dep_nop, "pk dep +s_nop", d, B, T, ghz, 8); run(k_pk_dep, "pk dep no nop", d, B, T, ghz, 8); run(k_add_dep, "u32 dep", d, B, T, ghz, 8);
    run(k_lds_read_i8, "ds_read_i8", d, B, T, ghz, 8); run(k_lds_write_b8, "ds_write_b8", d, B, T, ghz, 8);
    // low occupancy: 4 waves per CU (1 per SIMD)
    run(k_pk_add_u16, "pk_add 1w/SIMD", d, 1024, 64, ghz, 8); run(k_add_u32, "add 1w/SIMD", d, 1024, 64, ghz, 8);
    run(k_pk_add_u16, "pk_add 2w/SIMD", d, 2048, 64, ghz, 8);
    run(k_lds_write_w<uint16_t>, "ds_write_b16", d, B, T, ghz, 8); run(k_lds_write_w<uint32_t>, "ds_write_b32", d, B, T, ghz, 8);
    run(k_lds_read_w<uint16_t>, "ds_read_u16", d, B, T, ghz, 8); run(k_lds_read_w<uint32_t>, "ds_read_b32", d, B, T, ghz, 8);
    g_dyn_lds = 40 * 1024; /* 4 resident waves per SIMD */
    printf("-- 4 resident waves per SIMD (40 KB dynamic LDS per 256-thread block) --\n");
    run(k_add_u32, "v_add_u32 4w", d, B, T, ghz, 8); run(k_xor, "v_xor_b32 4w", d, B, T, ghz, 8); run(k_and_lit, "v_and lit 4w", d, B, T, ghz, 8);
    run(k_bitop3, "v_bitop3 4w", d, B, T, ghz, 8); run(k_pk_add_u16, "v_pk_add_u16 4w", d, B, T, ghz, 8); run(k_pk_mad, "v_pk_mad_u16 4w", d, B, T, ghz, 8);
    run(k_perm, "v_perm_b32 4w", d, B, T, ghz, 8); run(k_add_dep, "u32 dep 4w", d, B, T, ghz, 8); run(k_pk_dep, "pk dep 4w", d, B, T, ghz, 8);
    g_dyn_lds = 0;
    return 0;
}
