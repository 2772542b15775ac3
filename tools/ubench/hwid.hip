// Where does the dispatcher put the two waves of a 128-thread workgroup?  Every wave records HW_ID (gfx9 layout: WAVE_ID [3:0],
// SIMD_ID [5:4], CU_ID [11:8], SH_ID [12], SE_ID [15:13]) and XCC_ID, then idles long enough for the CU to fill (same LDS
// footprint as a codeword of the decode kernels: 8 workgroups per CU).  Prints, for the first waves of the grid, the SIMD pair
// of each workgroup and the slot parity of its first wave, and the per-SIMD count of first / second waves over the whole grid.
//   hipcc --offload-arch=gfx950 -O2 -o hwid hwid.hip && ./hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(128) void probe(unsigned* out)
{
    extern __shared__ unsigned char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 400000) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (threadIdx.x == 1000) smem[0] = 1;
}
int main()
{
    const int wgs = 2048;
    unsigned* d;
    hipMalloc(&d, wgs * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(128), 20424, 0, d);
    std::vector<unsigned> h(wgs * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int same_simd = 0, hist[4][2] = {}, role0[4] = {};
    std::map<int, int> pairs;
    for (int w = 0; w < wgs; ++w) {
        const unsigned a = h[w * 4], b = h[w * 4 + 2];
        const int sa = (a >> 4) & 3, sb = (b >> 4) & 3;
        same_simd += sa == sb;
        hist[sa][0]++; hist[sb][1]++;
        pairs[sa * 4 + sb]++;
        const int swap = a & 1; /* the rule of lnsfaid_kernel5.hip: slot parity of the first wave swaps the roles */
        role0[swap ? sb : sa]++;
        if (w < 24)
            printf("wg %4d  xcc %u se %u cu %2u | wave0 simd %d slot %u | wave1 simd %d slot %u\n", w, h[w * 4 + 1] & 7, (a >> 13) & 7, (a >> 8) & 15,
                   sa, a & 15, sb, b & 15);
    }
    printf("workgroups with both waves on one SIMD: %d of %d\n", same_simd, wgs);
    for (int s = 0; s < 4; ++s) printf("SIMD %d: first waves %d, second waves %d, role-0 waves under the slot-parity rule %d\n", s, hist[s][0], hist[s][1], role0[s]);
    for (auto& p : pairs) printf("pair (wave0 simd %d, wave1 simd %d): %d workgroups\n", p.first / 4, p.first % 4, p.second);
    return 0;
}
