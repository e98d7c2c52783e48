// How long is ONE dependent round of random 128-byte reads, as the decode kernel issues them (half-wave per record,
// six records in flight per wave, 16 waves per workgroup), against the FOOTPRINT the records are spread over?
// If the step from an L2-sized footprint to tens of GB is much more than the ~900-cycle HBM miss, address
// translation (UTCL2 misses, page walks) is part of every dependent access of the decode chain.
//   hipcc --offload-arch=gfx950 -O3 -o lat_probe lat_probe.hip && ./lat_probe
// Prints, per footprint and number of workgroups (1 = idle chip, 256 = every CU at once): us per dependent round
// (median of the workgroups), for chains of 1 and of 4 dependent rounds (the per-round cost is their difference / 3).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// records of 32 ints; a half-wave reads one record; each wave keeps DG records x 2 halves in flight per round; the next
// round's record index depends on the data read (all zeros: the dependency is real, the address stream is the hash)
template <int DG>
__global__ __launch_bounds__(1024) void chase(const int* __restrict__ buf, unsigned long long nrec, int rounds,
                                              unsigned seed, unsigned long long* __restrict__ res, int* sink) {
    const int lane = threadIdx.x & 63, half = lane >> 5, sl = lane & 31, wave = threadIdx.x >> 6;
    unsigned key = seed ^ (blockIdx.x * 7919u + wave * 131u + half * 17u);
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    int acc = 0;
    for (int r = 0; r < rounds; ++r) {
        int v[DG];
#pragma unroll
        for (int b = 0; b < DG; ++b) {
            const unsigned long long rec = ((unsigned long long)hash32(key + b * 0x9e3779b9u + acc) * nrec) >> 32;
            v[b] = buf[rec * 32 + sl];
        }
#pragma unroll
        for (int b = 0; b < DG; ++b) acc += __builtin_amdgcn_readlane(v[b], 0) + __builtin_amdgcn_readlane(v[b], 32);
        key = hash32(key + 0x51ed27u);
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) res[blockIdx.x] = t1 - t0;
    if (acc == 12345) *sink = acc;
}

int main() {
    const size_t sizes[] = {2ull << 20, 64ull << 20, 1ull << 30, 8ull << 30, 40ull << 30};
    unsigned long long* res; int* sink;
    hipMalloc(&res, 4096 * 8); hipMalloc(&sink, 4);
    for (size_t bytes : sizes) {
        int* buf = nullptr;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("%zu MB: alloc failed\n", bytes >> 20); continue; }
        hipMemset(buf, 0, bytes);
        const unsigned long long nrec = bytes / 128;
        for (int grid : {1, 8, 256}) {
            double med[2];
            int ri = 0;
            for (int rounds : {1, 4}) {
                std::vector<double> all;
                for (int rep = 0; rep < 12; ++rep) {
                    hipLaunchKernelGGL(chase<6>, dim3(grid), dim3(1024), 0, 0, buf, nrec, rounds, 1234u + rep * 977u, res, sink);
                    hipDeviceSynchronize();
                    std::vector<unsigned long long> h(grid);
                    hipMemcpy(h.data(), res, grid * 8, hipMemcpyDeviceToHost);
                    if (rep >= 2) for (auto x : h) all.push_back(x * 0.01);
                }
                std::sort(all.begin(), all.end());
                med[ri++] = all[all.size() / 2];
            }
            printf("footprint %6zu MB  workgroups %3d: 1 round %.2f us, 4 rounds %.2f us -> %.2f us per dependent round\n",
                   bytes >> 20, grid, med[0], med[1], (med[1] - med[0]) / 3.0);
        }
        hipFree(buf);
    }
    return 0;
}
