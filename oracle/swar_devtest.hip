/*
 * swar_devtest.hip — TEST INFRASTRUCTURE (GPU side of tests/test_gpu_swar.py): runs the building blocks of
 * csrc/lnsfaid_swar.h on the device in isolation, so that a difference between the ISA and the header's host restatements
 * (v_perm_b32 selectors 8..13, v_alignbyte_b32, v_bitop3_b32) or in the layer step itself shows up without the decode
 * kernel around it.  Not part of the product library.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../mod-interleaveavx_multithreads-faid_amd/csrc/lnsfaid_swar.h"

__global__ void k_ops(int n, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[0 * n + i] = sw_perm(a[i], b[i], c[i]);
    out[1 * n + i] = sw_alignbyte(a[i], b[i], c[i] % 5u);
    out[2 * n + i] = sw_bitop3<SW_TT_SEL>(a[i], b[i], c[i]);
    out[3 * n + i] = sw_bitop3<SW_TT_NANDOR>(a[i], b[i], c[i]);
    out[4 * n + i] = sw_mask7(a[i], sw_vconst(SW_SEL_SIGN));
    out[5 * n + i] = sw_bitop3<SW_TT_A_AND_BORC>(a[i], b[i], c[i]);
    out[6 * n + i] = sw_bitop3<SW_TT_BFI_C>(a[i], b[i], c[i]);
    out[7 * n + i] = sw_bitop3<SW_TT_XORAND>(a[i], b[i], c[i]);
}

extern "C" int swar_devtest_ops(int n, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* out /* [8][n] */)
{
    uint32_t *da, *db, *dc, *dout;
    if (hipMalloc(&da, n * 4) || hipMalloc(&db, n * 4) || hipMalloc(&dc, n * 4) || hipMalloc(&dout, 8 * n * 4)) return -1;
    hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_ops, dim3((n + 255) / 256), dim3(256), 0, 0, n, da, db, dc, dout);
    const int rc = hipMemcpy(out, dout, 8 * n * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
    hipFree(da); hipFree(db); hipFree(dc); hipFree(dout);
    return rc;
}

struct TabDev {
    const uint32_t* row; /* global: sb of the layer's edges */
    uint32_t sbv;
    __device__ __forceinline__ uint32_t sb(int j) const { return __builtin_amdgcn_readfirstlane(row[j]); }
    __device__ __forceinline__ uint32_t s4(int j) const { return (sb(j) & 255u) << 2; }
    __device__ __forceinline__ uint32_t cb256(int j) const { return sb(j) & ~255u; }
    __device__ __forceinline__ uint32_t sb_dyn4(uint32_t idx4) const { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)idx4, (int)sbv); }
};

/* one wave per codeword: n_iter layered iterations of DecodeMethod 2 (no syndrome needed) on an interleaved, biased En image */
__global__ __launch_bounds__(64) void k_layers(int n_var, int nbr, const int* deg, const uint32_t* sb /* [nbr][24] */, SwParams p6[6],
                                               int n_iter, uint8_t* img /* [n_cw][n_var] */, SwRow* rows /* [n_cw][nbr][64] */)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x, cw = blockIdx.x;
    uint32_t* s32 = (uint32_t*)smem;
    const uint32_t* g32 = (const uint32_t*)(img + (size_t)cw * n_var);
    for (int i = lane; i < n_var / 4; i += 64) s32[i] = g32[i];
    __syncthreads();
    SwLds lds = SwLds();
    const SwK K = sw_consts();
    SwRow* r = rows + (size_t)cw * nbr * 64;
    for (int it = 1; it <= n_iter; ++it) {
        const SwParams p = p6[it <= 5 ? it - 1 : 5];
        for (int br = 0; br < nbr; ++br) {
            TabDev tab; tab.row = sb + br * 24; { const uint32_t v = sb[br * 24 + (lane % 24)]; tab.sbv = ((v & ~255u) << 16) | ((v & 255u) << 2); }
            SwRow cur = { { 0u, 0u, 0u }, 0u, { 0u, 0u } };
            if (it > 1) cur = r[br * 64 + lane];
            SwRow st;
            if (deg[br] == 23) st = sw_layer_step<2, 23>(lds, tab, p, K, (uint32_t)lane, 23, cur, it == 1, 0u, false);
            else st = sw_layer_step<2, 0>(lds, tab, p, K, (uint32_t)lane, deg[br], cur, it == 1, 0u, false);
            r[br * 64 + lane] = st;
            __syncthreads();
        }
    }
    uint32_t* o32 = (uint32_t*)(img + (size_t)cw * n_var);
    for (int i = lane; i < n_var / 4; i += 64) o32[i] = s32[i];
}

/* layout check for the ctypes mirror in tests/test_gpu_swar.py */
extern "C" int swar_devtest_sizeof_params(void) { return (int)sizeof(SwParams); }

extern "C" int swar_devtest_layers(int n_cw, int n_var, int nbr, const int* deg, const uint32_t* sb, const SwParams* p6, int n_iter,
                                   uint8_t* img)
{
    int* ddeg; uint32_t* dsb; SwParams* dp; uint8_t* dimg; SwRow* drows;
    if (hipMalloc(&ddeg, nbr * 4) || hipMalloc(&dsb, nbr * 24 * 4) || hipMalloc(&dp, 6 * sizeof(SwParams))
        || hipMalloc(&dimg, (size_t)n_cw * n_var) || hipMalloc(&drows, (size_t)n_cw * nbr * 64 * sizeof(SwRow))) return -1;
    hipMemcpy(ddeg, deg, nbr * 4, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, nbr * 24 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, p6, 6 * sizeof(SwParams), hipMemcpyHostToDevice); hipMemcpy(dimg, img, (size_t)n_cw * n_var, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layers, dim3(n_cw), dim3(64), (size_t)n_var, 0, n_var, nbr, ddeg, dsb, dp, n_iter, dimg, drows);
    const int rc = hipMemcpy(img, dimg, (size_t)n_cw * n_var, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
    hipFree(ddeg); hipFree(dsb); hipFree(dp); hipFree(dimg); hipFree(drows);
    return rc;
}
