// Can a workgroup poll a word that a workgroup on ANOTHER CU of the same XCD stores (sc0), and with what?
// block 8 waits ~5 us, then stores; block 0 polls.  Variants of the poll: 0 = atomic load, workgroup scope
// (global_load sc0), 1 = atomic load, agent scope (sc1), 2 = returning atomic or (executes in L2).
// The second part measures contention: NP pollers (blocks 0, 16, 24, ... on the same XCD) x 64 lanes polling words of
// the same cache lines.
//   hipcc --offload-arch=gfx950 -O3 -o l2_poll l2_poll.hip && ./l2_poll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void poll_kernel(unsigned long long* word, unsigned long long* res, int nwords, unsigned seq) {
    const int b = blockIdx.x;
    if ((b & 7) != 0) return;                       // XCD 0 only
    if (b == 8) {                                   // the publisher
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 500) {}        // 5 us at 100 MHz
        if (threadIdx.x < nwords) {
            __hip_atomic_store(word + threadIdx.x, ((unsigned long long)seq << 32) | threadIdx.x, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (threadIdx.x == 0) res[1] = wall_clock64();
        return;
    }
    // pollers
    if (threadIdx.x >= nwords) return;
    unsigned long long v = 0;
    int it = 0;
    for (; it < 4000; ++it) {
        if (MODE == 0) v = __hip_atomic_load(word + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) v = __hip_atomic_load(word + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) v = __hip_atomic_fetch_or(word + threadIdx.x, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((unsigned)(v >> 32) == seq) break;
    }
    if (b == 0 && threadIdx.x == 0) { res[0] = wall_clock64(); res[2] = it; }
}

int main() {
    unsigned long long *word, *res;
    hipMalloc(&word, 4096); hipMalloc(&res, 64);
    hipMemset(word, 0, 4096);
    unsigned seq = 1;
    for (int mode = 0; mode < 3; ++mode)
        for (int np : {1, 8, 24})                       // poller workgroups on XCD 0 (blocks 0, 16, 24, ...)
            for (int nwords : {1, 48}) {
                double lat = 0, its = 0; int ok = 0;
                for (int rep = 0; rep < 10; ++rep, ++seq) {
                    hipMemset(res, 0, 64);
                    const int grid = 16 + 8 * (np - 1);
                    if (mode == 0) hipLaunchKernelGGL(poll_kernel<0>, dim3(grid), dim3(64), 0, 0, word, res, nwords, seq);
                    if (mode == 1) hipLaunchKernelGGL(poll_kernel<1>, dim3(grid), dim3(64), 0, 0, word, res, nwords, seq);
                    if (mode == 2) hipLaunchKernelGGL(poll_kernel<2>, dim3(grid), dim3(64), 0, 0, word, res, nwords, seq);
                    hipDeviceSynchronize();
                    unsigned long long h[3]; hipMemcpy(h, res, 24, hipMemcpyDeviceToHost);
                    if (h[2] < 4000) { ++ok; lat += (double)((long long)(h[0] - h[1])) * 0.01; its += h[2]; }
                }
                printf("mode %d pollers %2d words %2d: seen %d/10, store -> seen %.2f us, %.0f polls in ~5 us (%.2f us per poll)\n",
                       mode, np, nwords, ok, ok ? lat / ok : 0., ok ? its / ok : 0., ok ? 5.0 / (its / ok) : 0.);
            }
    return 0;
}
