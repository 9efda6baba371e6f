// calib_scatter.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of k_search
// (profiles/hbm_traffic.json): scattered 16-byte reads, scattered 64-byte (4 x 16 B, one lane) reads, scattered 16-byte
// and 4-byte writes, against coalesced 16-B-per-lane streams.  Every kernel touches a known number of bytes in an
// 8 GiB buffer (far beyond the 256 MiB Infinity Cache), so counter / bytes is the factor for that shape.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/calib_scatter tools/calib_scatter.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./tools/calib_scatter
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out2 -- ./tools/calib_scatter
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHK(x)                                                                         \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// n16 = number of 16-byte units in the buffer (a power of two)
__global__ void k_read16_scatter(const uint4* __restrict__ buf, uint64_t n16, uint32_t reps, uint32_t* sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < reps; r++) acc += buf[mix(tid * reps + r) & (n16 - 1)].x;
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read64_scatter(const uint4* __restrict__ buf, uint64_t n16, uint32_t reps, uint32_t* sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < reps; r++) {
        const uint64_t b = (mix(tid * reps + r) & (n16 - 1)) & ~3ull;  // one 64-byte line, four 16-byte loads by one lane
        acc += buf[b].x + buf[b + 1].x + buf[b + 2].x + buf[b + 3].x;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_write16_scatter(uint4* __restrict__ buf, uint64_t n16, uint32_t reps) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t r = 0; r < reps; r++) buf[mix(tid * reps + r) & (n16 - 1)] = make_uint4((uint32_t)tid, r, 1, 2);
}
__global__ void k_write4_scatter(uint32_t* __restrict__ buf, uint64_t n4, uint32_t reps) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t r = 0; r < reps; r++) buf[mix(tid * reps + r) & (n4 - 1)] = (uint32_t)tid;
}
__global__ void k_read16_stream(const uint4* __restrict__ buf, uint64_t n16, uint32_t* sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = tid; i < n16; i += nt) acc += buf[i].x;
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_write16_stream(uint4* __restrict__ buf, uint64_t n16) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = tid; i < n16; i += nt) buf[i] = make_uint4((uint32_t)i, 0, 1, 2);
}

int main() {
    const uint64_t bytes = 8ull << 30, n16 = bytes / 16;
    uint4* buf = nullptr;
    uint32_t* sink = nullptr;
    CHK(hipMalloc((void**)&buf, bytes));
    CHK(hipMalloc((void**)&sink, 4));
    CHK(hipMemset(buf, 1, bytes));
    const uint32_t blocks = 256 * 16, threads = 256, reps = 64;
    const uint64_t n = (uint64_t)blocks * threads * reps;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    float ms;
#define RUN(name, bytes_touched, ...)                                                                              \
    CHK(hipEventRecord(e0));                                                                                       \
    hipLaunchKernelGGL(name, dim3(blocks), dim3(threads), 0, 0, __VA_ARGS__);                                      \
    CHK(hipEventRecord(e1));                                                                                       \
    CHK(hipEventSynchronize(e1));                                                                                  \
    CHK(hipEventElapsedTime(&ms, e0, e1));                                                                         \
    printf("%-18s bytes %llu  %.3f ms  %.1f GB/s\n", #name, (unsigned long long)(bytes_touched), ms, (double)(bytes_touched) / ms / 1e6);
    for (int rep = 0; rep < 2; rep++) {
        RUN(k_read16_scatter, n * 16, buf, n16, reps, sink)
        RUN(k_read64_scatter, n * 64, buf, n16, reps, sink)
        RUN(k_write16_scatter, n * 16, buf, n16, reps)
        RUN(k_write4_scatter, n * 4, reinterpret_cast<uint32_t*>(buf), n16 * 4, reps)
        RUN(k_read16_stream, bytes, buf, n16, sink)
        RUN(k_write16_stream, bytes, buf, n16)
    }
    CHK(hipFree(buf));
    CHK(hipFree(sink));
    return 0;
}
