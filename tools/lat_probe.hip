// Latency of one dependent access as a lone wavefront sees it (pointer chase): vector L1, L2, Infinity Cache, HBM, LDS,
// and ds_bpermute.  hipcc --offload-arch=gfx950 -O3 -o tools/lat_probe tools/lat_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const uint32_t* __restrict__ next, uint32_t start, int hops, unsigned long long* out) {
    uint32_t i = start;
    // warm-up pass (brings the chain into whatever cache holds it)
    for (int k = 0; k < hops; k++) i = __builtin_nontemporal_load(&next[i]) , i = next[i];
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int k = 0; k < hops; k++) i = next[i];
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void chase_lds(int hops, unsigned long long* out) {
    __shared__ uint32_t s[4096];
    for (int k = threadIdx.x; k < 4096; k += blockDim.x) s[k] = (k * 1237u + 17u) & 4095u;
    __syncthreads();
    uint32_t i = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int k = 0; k < hops; k++) i = s[i];
    unsigned long long t1 = clock64();
    uint32_t j = threadIdx.x;
    unsigned long long t2 = clock64();
    for (int k = 0; k < hops; k++) j = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((j * 5u + 1u) & 63u) << 2, (int)j);
    unsigned long long t3 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; out[2] = t3 - t2; out[3] = j; }
}
int main() {
    unsigned long long* d_out; hipMalloc(&d_out, 64); unsigned long long h[4];
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("shader clock %d kHz, wall clock %d kHz\n", clk, wclk);
    for (size_t bytes : {(size_t)8 << 10, (size_t)24 << 10, (size_t)256 << 10, (size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)2 << 30}) {
        size_t n = bytes / 4, stride = 32;  // one hop per 128-byte line
        size_t lines = n / stride;
        std::vector<uint32_t> perm(lines); std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> next(n, 0);
        for (size_t k = 0; k < lines; k++) next[(size_t)perm[k] * stride] = perm[(k + 1) % lines] * (uint32_t)stride;
        uint32_t* d; hipMalloc(&d, bytes); hipMemcpy(d, next.data(), bytes, hipMemcpyHostToDevice);
        int hops = (int)std::min<size_t>(lines, 20000);
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, perm[0] * (uint32_t)stride, hops, d_out);
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("chain of %8zu KB: %7.1f clock64 ticks per dependent load (%d hops)\n", bytes >> 10, (double)h[0] / hops, hops);
        hipFree(d);
    }
    hipLaunchKernelGGL(chase_lds, dim3(1), dim3(64), 0, 0, 20000, d_out);
    hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost);
    printf("LDS read: %.1f ticks per dependent ds_read_b32; ds_bpermute: %.1f ticks per dependent op\n", (double)h[0] / 20000, (double)h[2] / 20000);
    return 0;
}
