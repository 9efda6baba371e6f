// tools/bitonic_probe.hip -- hardware check of the pieces r_sort_bitonic (fxjps_kernels.hip.inc) is made of: where each DPP
// pattern / ds_swizzle pattern reads from, one compare-exchange stage, and the whole sort against std::sort.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bitonic_probe tools/bitonic_probe.hip && tools/bitonic_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
#include "../fuxi-planner_amd/csrc/fxjps_kernels.hip.inc"

#define MAPDPP(name, str)                                                                                \
    {                                                                                                    \
        uint32_t v = (uint32_t)lane, o = 999u;                                                           \
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 " str " row_mask:0xf bank_mask:0xf" : "+v"(o) : "v"(v)); \
        out[k * 64 + lane] = o;                                                                          \
        k++;                                                                                             \
    }
__global__ void k_maps(uint32_t* out) {
    const int lane = threadIdx.x;
    int k = 0;
    MAPDPP(0, "quad_perm:[1,0,3,2]");
    MAPDPP(1, "quad_perm:[3,2,1,0]");
    MAPDPP(2, "quad_perm:[2,3,0,1]");
    MAPDPP(3, "row_half_mirror");
    MAPDPP(4, "row_mirror");
    MAPDPP(5, "row_ror:8");
    out[k++ * 64 + lane] = (uint32_t)__builtin_amdgcn_ds_swizzle(lane, 0x1F | (4 << 10));
    out[k++ * 64 + lane] = (uint32_t)__builtin_amdgcn_ds_swizzle(lane, 0x1F | (16 << 10));
    out[k++ * 64 + lane] = (uint32_t)__builtin_amdgcn_ds_swizzle(lane, 0x1F | (31 << 10));
    out[k++ * 64 + lane] = (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 63) << 2, lane);
}
__global__ void k_sort(const uint64_t* f, const uint32_t* x, int n, uint64_t* of, uint32_t* ox, uint32_t* os, int rounds) {
    const int lane = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        fx::RTier R;
        R.f = f[r * 64 + lane];
        R.x = x[r * 64 + lane];
        R.s = (uint32_t)lane;
        R.c = (uint32_t)lane * 3u;
        if (lane >= n) {
            R.f = ~0ull;
            R.x = ~0u;
        }
        fx::r_sort_bitonic(R, lane, n);
        of[r * 64 + lane] = R.f;
        ox[r * 64 + lane] = R.x;
        os[r * 64 + lane] = R.s | (R.c << 8);
    }
}
int main() {
    uint32_t* dm;
    hipMalloc(&dm, 10 * 64 * 4);
    hipLaunchKernelGGL(k_maps, 1, 64, 0, 0, dm);
    std::vector<uint32_t> hm(640);
    hipMemcpy(hm.data(), dm, 640 * 4, hipMemcpyDeviceToHost);
    const char* names[10] = {"quad_perm[1,0,3,2] (want i^1)", "quad_perm[3,2,1,0] (i^3)", "quad_perm[2,3,0,1] (i^2)", "row_half_mirror (i^7)", "row_mirror (i^15)",
                             "row_ror:8 (i^8)", "swizzle xor 4", "swizzle xor 16", "swizzle xor 31", "bpermute xor 63"};
    const int want[10] = {1, 3, 2, 7, 15, 8, 4, 16, 31, 63};
    for (int k = 0; k < 10; k++) {
        int bad = 0;
        for (int i = 0; i < 64; i++) bad += hm[k * 64 + i] != (uint32_t)(i ^ want[k]);
        printf("%-34s %s", names[k], bad ? "DIFFERS:" : "ok\n");
        if (bad) {
            for (int i = 0; i < 16; i++) printf(" %u", hm[k * 64 + i]);
            printf(" ...\n");
        }
    }
    const int rounds = 2000;
    std::mt19937_64 rng(5);
    int total_bad = 0;
    for (int n : {0, 1, 2, 7, 16, 17, 31, 32, 33, 47, 48, 63, 64}) {
        std::vector<uint64_t> f(rounds * 64);
        std::vector<uint32_t> x(rounds * 64);
        for (int i = 0; i < rounds * 64; i++) {
            const int mode = (i / 64) % 4;
            f[i] = mode == 0 ? rng() >> 1 : mode == 1 ? (0x4080000000000000ull + (rng() % 5)) : mode == 2 ? (0x4080000000000000ull + ((rng() % 3) << 32)) : 0x4080000000000000ull;
            x[i] = mode == 3 ? (uint32_t)(rng() % 7) : (uint32_t)(rng() % (mode == 1 ? 4 : 1000000));
        }
        uint64_t *df, *dof;
        uint32_t *dx, *dox, *dos;
        hipMalloc(&df, f.size() * 8); hipMalloc(&dof, f.size() * 8); hipMalloc(&dx, x.size() * 4); hipMalloc(&dox, x.size() * 4); hipMalloc(&dos, x.size() * 4);
        hipMemcpy(df, f.data(), f.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_sort, 1, 64, 0, 0, df, dx, n, dof, dox, dos, rounds);
        std::vector<uint64_t> of(f.size());
        std::vector<uint32_t> ox(x.size()), os(x.size());
        hipMemcpy(of.data(), dof, of.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(ox.data(), dox, ox.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(os.data(), dos, os.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int r = 0; r < rounds; r++) {
            std::vector<std::pair<uint64_t, uint32_t>> w;
            for (int i = 0; i < n; i++) w.push_back({f[r * 64 + i], x[r * 64 + i]});
            std::sort(w.begin(), w.end());
            bool ok = true;
            std::vector<int> seen(64, 0);
            for (int i = 0; i < 64 && ok; i++) {
                if (i < n) {
                    ok = of[r * 64 + i] == w[i].first && ox[r * 64 + i] == w[i].second;
                    const uint32_t src = os[r * 64 + i] & 0xFF;
                    // the payload travelled with its key: the entry that came from lane src
                    ok = ok && src < (uint32_t)n && !seen[src] && f[r * 64 + src] == of[r * 64 + i] && x[r * 64 + src] == ox[r * 64 + i] && (os[r * 64 + i] >> 8) == src * 3u;
                    if (src < 64) seen[src] = 1;
                } else {
                    ok = of[r * 64 + i] == ~0ull && ox[r * 64 + i] == ~0u;
                }
            }
            bad += !ok;
        }
        printf("sort of %2d entries, %d rounds: %d wrong\n", n, rounds, bad);
        total_bad += bad;
        hipFree(df); hipFree(dof); hipFree(dx); hipFree(dox); hipFree(dos);
    }
    printf("TOTAL WRONG %d\n", total_bad);
    return total_bad ? 1 : 0;
}
