// Which XCD / CU do the blocks of a launch land on -- on the default stream and on streams made with
// hipExtStreamCreateWithCUMask?  (the head launch on an XCD of its own.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/cumask_probe tools/cumask_probe.hip && tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_where(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID: cu 11:8, sh 12, se 15:13
        out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xFFu);
    }
    // keep the CU busy for a while so that the blocks spread over everything the stream may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {
    }
}

static void run(const char* name, hipStream_t st, int nblocks) {
    unsigned* d = nullptr;
    hipMalloc(&d, nblocks * sizeof(unsigned));
    hipMemsetAsync(d, 0xFF, nblocks * sizeof(unsigned), st);
    hipLaunchKernelGGL(k_where, dim3(nblocks), dim3(512), 60 << 10, st, d, 2000000);  // 60 KB of LDS: two blocks per CU at most
    std::vector<unsigned> h(nblocks);
    hipMemcpyAsync(h.data(), d, nblocks * sizeof(unsigned), hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    int per_xcc[16] = {0};
    bool cu_seen[16][256];
    memset(cu_seen, 0, sizeof(cu_seen));
    for (unsigned v : h) {
        per_xcc[(v >> 16) & 15]++;
        cu_seen[(v >> 16) & 15][v & 0xFF] = true;
    }
    printf("%-34s blocks per XCC:", name);
    for (int x = 0; x < 8; x++) {
        int cus = 0;
        for (int c = 0; c < 256; c++) cus += cu_seen[x][c];
        printf(" %d:%d(%d cu)", x, per_xcc[x], cus);
    }
    printf("   first blocks -> xcc:");
    for (int i = 0; i < 16 && i < nblocks; i++) printf(" %u", (h[i] >> 16) & 15);
    printf("\n");
    hipFree(d);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_where), hipFuncAttributeMaxDynamicSharedMemorySize, 60 << 10);
    hipStream_t s0;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    run("plain stream, 512 blocks", s0, 512);
    run("plain stream, 16 blocks", s0, 16);
    const int nw = (p.multiProcessorCount + 31) / 32;
    auto masked = [&](const char* name, auto pred, int nblocks) {
        std::vector<uint32_t> m(nw, 0u);
        for (int i = 0; i < p.multiProcessorCount; i++)
            if (pred(i)) m[i >> 5] |= 1u << (i & 31);
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nw, m.data());
        if (e != hipSuccess) {
            printf("%s: hipExtStreamCreateWithCUMask: %s\n", name, hipGetErrorString(e));
            return;
        }
        run(name, s, nblocks);
        hipStreamDestroy(s);
    };
    masked("mask bits i % 8 == 0, 128 blocks", [](int i) { return i % 8 == 0; }, 128);
    masked("mask bits i % 8 != 0, 512 blocks", [](int i) { return i % 8 != 0; }, 512);
    masked("mask bits i < 32, 128 blocks", [](int i) { return i < 32; }, 128);
    masked("mask bits i % 8 < 2, 128 blocks", [](int i) { return i % 8 < 2; }, 128);
    masked("mask bits i % 8 == 0, 16 blocks", [](int i) { return i % 8 == 0; }, 16);
    return 0;
}
