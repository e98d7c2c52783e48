// Does touching a region's pages early take the address-translation cost off a later DEPENDENT random access?
// (lat_probe: the first round of random 128-byte reads of a launch costs 2.3-2.5 us in a >= 8 GB footprint against
// 1.1 us in 2 MB; later rounds ~1.0-1.6.)  Every workgroup optionally "warms" first -- one 4-byte load per `stride`
// bytes of the footprint, results unused, by its LAST wave -- then waits `delay` us (the decode kernel's hash phase
// stands between its start and its first table access), then times ONE round of random record reads as lat_probe.
//   hipcc --offload-arch=gfx950 -O3 -o tlb_warm_probe tlb_warm_probe.hip && ./tlb_warm_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// who: 0 = nobody warms, 1 = every workgroup, 2 = one workgroup per XCD (blocks 0..7)
__global__ __launch_bounds__(1024) void probe(const int* __restrict__ buf, unsigned long long bytes,
                                              unsigned long long stride, int who, int delay_ticks, unsigned seed,
                                              unsigned long long* __restrict__ res, int* sink) {
    const int lane = threadIdx.x & 63, half = lane >> 5, sl = lane & 31, wave = threadIdx.x >> 6;
    const unsigned long long t_start = wall_clock64();
    int w = 0;
    if (stride && (who == 1 || (who == 2 && blockIdx.x < 8)) && wave == 15) {
        const unsigned long long pages = bytes / stride;
        for (unsigned long long p = lane; p < pages; p += 64)
            w += __builtin_nontemporal_load(buf + (p * stride) / 4 + 32 * ((blockIdx.x * 37 + p) % (stride / 128 < 1024 ? stride / 128 : 1024)));
    }
    while (wall_clock64() - t_start < (unsigned long long)delay_ticks) {}
    __syncthreads();
    const unsigned long long nrec = bytes / 128;
    unsigned key = seed ^ (blockIdx.x * 7919u + wave * 131u + half * 17u);
    const unsigned long long t0 = wall_clock64();
    int v[6], acc = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const unsigned long long rec = ((unsigned long long)hash32(key + b * 0x9e3779b9u) * nrec) >> 32;
        v[b] = buf[rec * 32 + sl];
    }
#pragma unroll
    for (int b = 0; b < 6; ++b) acc += __builtin_amdgcn_readlane(v[b], 0) + __builtin_amdgcn_readlane(v[b], 32);
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) res[blockIdx.x] = t1 - t0;
    if (acc + w == 12345) *sink = acc;
}

int main() {
    unsigned long long* res; int* sink;
    hipMalloc(&res, 4096 * 8); hipMalloc(&sink, 4);
    // a second large buffer swept between launches so that nothing of `buf` survives in L2 / TLBs from the previous run
    int* other = nullptr; const size_t obytes = 2ull << 30;
    hipMalloc(&other, obytes);
    for (size_t bytes : {1ull << 30, 8ull << 30}) {
        int* buf = nullptr;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); continue; }
        hipMemset(buf, 0, bytes);
        for (int grid : {8, 256})
            for (int who : {0, 1, 2})
                for (unsigned long long stride : {2ull << 20, 64ull << 10, 1ull << 30}) {
                    if (who == 0 && stride != (2ull << 20)) continue;
                    std::vector<double> all;
                    for (int rep = 0; rep < 10; ++rep) {
                        hipMemset(other, rep, obytes);
                        hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, buf, bytes, stride, who, 350, 99u + rep * 31u, res, sink);
                        hipDeviceSynchronize();
                        std::vector<unsigned long long> h(grid);
                        hipMemcpy(h.data(), res, grid * 8, hipMemcpyDeviceToHost);
                        if (rep >= 2) for (auto x : h) all.push_back(x * 0.01);
                    }
                    std::sort(all.begin(), all.end());
                    printf("footprint %5zu MB grid %3d warm-by %s stride %7llu KB: round median %.2f us  p90 %.2f us\n", bytes >> 20, grid,
                           who == 0 ? "nobody  " : who == 1 ? "every WG" : "1 per XCD", stride >> 10, all[all.size() / 2], all[all.size() * 9 / 10]);
                }
        hipFree(buf);
    }
    return 0;
}
