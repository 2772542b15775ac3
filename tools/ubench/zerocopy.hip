// PCIe from inside a kernel against the copy engines: a kernel that reads host-mapped (zero-copy) memory, one that writes it, both
// at once, and hipMemcpyAsync of the same buffers, each for a few transfer sizes and numbers of workgroups (the call combiner of
// lnsfaid_capi.hip lets the decode kernel read / write its pinned staging slots directly: what is that worth per direction?).
//   hipcc --offload-arch=gfx950 -O2 -o zerocopy zerocopy.hip && ./zerocopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t max_bytes = 256u << 20;
    uint4 *h_in, *h_out, *d_a, *d_b, *dh_in, *dh_out;
    hipHostMalloc((void**)&h_in, max_bytes, hipHostMallocMapped);
    hipHostMalloc((void**)&h_out, max_bytes, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&dh_in, h_in, 0);
    hipHostGetDevicePointer((void**)&dh_out, h_out, 0);
    hipMalloc((void**)&d_a, max_bytes);
    hipMalloc((void**)&d_b, max_bytes);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (size_t mb : { 18u, 36u, 256u }) {
        const size_t bytes = mb << 20, n = bytes / 16;
        for (int wgs : { 256, 1024, 2048, 8192 }) {
            double t[3];
            for (int mode = 0; mode < 3; ++mode) {
                for (int rep = 0; rep < 3; ++rep) { /* last repetition counts */
                    hipDeviceSynchronize();
                    const double t0 = now();
                    if (mode == 0 || mode == 2) hipLaunchKernelGGL(copy_k, dim3(wgs), dim3(256), 0, s1, dh_in, d_a, n);   /* kernel reads host */
                    if (mode == 1 || mode == 2) hipLaunchKernelGGL(copy_k, dim3(wgs), dim3(256), 0, s2, d_b, dh_out, n);  /* kernel writes host */
                    hipDeviceSynchronize();
                    t[mode] = now() - t0;
                }
            }
            printf("%3zu MB, %5d workgroups: kernel reads host %.1f GB/s, kernel writes host %.1f GB/s, both at once %.1f GB/s each\n", mb, wgs,
                   bytes / t[0] / 1e9, bytes / t[1] / 1e9, bytes / t[2] / 1e9);
        }
        double t[3];
        for (int mode = 0; mode < 3; ++mode)
            for (int rep = 0; rep < 3; ++rep) {
                hipDeviceSynchronize();
                const double t0 = now();
                if (mode == 0 || mode == 2) hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s1);
                if (mode == 1 || mode == 2) hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s2);
                hipDeviceSynchronize();
                t[mode] = now() - t0;
            }
        printf("%3zu MB, hipMemcpyAsync: host to device %.1f GB/s, device to host %.1f GB/s, both at once %.1f GB/s each\n", mb, bytes / t[0] / 1e9,
               bytes / t[1] / 1e9, bytes / t[2] / 1e9);
    }
    return 0;
}
