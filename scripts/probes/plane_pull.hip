// How fast can every workgroup of a 256-block launch pull the same hyperplane block (KL x 128 bf16, 16-byte chunk
// major as lsh_decode_kernel reads it) through its XCD's L2?  Variants: everything in flight at once (NPASS loads
// of CH chunks issued back to back) against "first pass, then the rest" as the decode prologue does.
//   hipcc --offload-arch=gfx950 -O3 -o plane_pull plane_pull.hip && ./plane_pull
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int CH, int MODE>
__global__ __launch_bounds__(1024) void pull(const u32x4* __restrict__ W, int KLpad, int KL, uint32_t* out,
                                             unsigned long long* stamp) {
    const int tid = threadIdx.x;
    unsigned long long t0 = wall_clock64();
    uint32_t acc = 0;
    if (MODE == 0) {                       // pass by pass: loads of pass p+1 issued after pass p is consumed
        for (int c0 = 0; c0 < KL; c0 += 1024) {
            const int c = (c0 + tid) < KLpad ? c0 + tid : KLpad - 1;
            u32x4 w[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) w[i] = W[c + (int64_t)i * KLpad];
#pragma unroll
            for (int i = 0; i < CH; ++i) acc += w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
        }
    } else {                               // all passes in flight at once (KL <= 2048)
        const int ca = tid < KLpad ? tid : KLpad - 1;
        const int cb = (1024 + tid) < KLpad ? 1024 + tid : KLpad - 1;
        u32x4 w[CH], v[CH / 2];
#pragma unroll
        for (int i = 0; i < CH; ++i) w[i] = W[ca + (int64_t)i * KLpad];
        // second pass: the 476 columns spread over all 1024 threads, half the chunks each
        const int half = tid & 1;
        const int col = 1024 + (tid >> 1);
        const int cc = col < KLpad ? col : KLpad - 1;
#pragma unroll
        for (int i = 0; i < CH / 2; ++i) v[i] = W[cc + (int64_t)(i + half * (CH / 2)) * KLpad];
#pragma unroll
        for (int i = 0; i < CH; ++i) acc += w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
#pragma unroll
        for (int i = 0; i < CH / 2; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        (void)cb;
    }
    __syncthreads();
    unsigned long long t1 = wall_clock64();
    out[blockIdx.x * 1024 + tid] = acc;
    if (tid == 0) { stamp[blockIdx.x * 2] = t0; stamp[blockIdx.x * 2 + 1] = t1; }
}

int main() {
    const int KL = 1500, KLpad = 1536, CHK = 16;
    u32x4* W; uint32_t* out; unsigned long long* st;
    hipMalloc(&W, (size_t)KLpad * CHK * 16); hipMemset(W, 1, (size_t)KLpad * CHK * 16);
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 256 * 2 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int grid : {8, 64, 256}) {
            float best = 1e9;
            std::vector<unsigned long long> h(512);
            double dur = 0;
            for (int rep = 0; rep < 20; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL((pull<16, 0>), dim3(grid), dim3(1024), 0, 0, W, KLpad, KL, out, st);
                else hipLaunchKernelGGL((pull<16, 1>), dim3(grid), dim3(1024), 0, 0, W, KLpad, KL, out, st);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost);
                double s = 0; for (int b = 0; b < grid; ++b) s += (double)(h[2 * b + 1] - h[2 * b]);
                dur = s / grid;
            }
            printf("mode %d (%s) grid %3d: launch %.2f us, mean in-kernel %.2f us\n", mode,
                   mode ? "all in flight" : "pass by pass", grid, best * 1e3, dur * 0.01);
        }
    return 0;
}
