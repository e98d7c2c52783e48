// Round 6, VERDICT r05 item 2: what is the BEST CASE of a query SimHash by MFMA inside the decode launch at one workgroup per
// head (cfg 2 / 3: B*H = 256 heads, K*L = 1 700 planes of 128 dims)?
//   valu : what lsh_decode_kernel does today -- every workgroup pulls all planes (435 KB) through its XCD's L2 path and
//          evaluates <= 2 planes per thread by v_dot2c chains (the normalisation, the exact-sign guard and the code cutting
//          are left out on both sides: they are the same work either way).
//   quad : the four workgroups b, b + 8, b + 16, b + 24 of a block of 32 (one XCD) share the work -- member m evaluates the
//          32-plane tiles t = m (mod 4) against the FOUR query rows with v_mfma_f32_32x32x16_bf16 (rows 4 .. 31 of the B
//          operand are zero: the matrix pipe is idle for 7/8 of the tile, which costs nothing here), publishes a tile's 32 sign
//          bits per row as a (launch number << 32 | bits) word to that row's head and polls its own head's 54 words.
//          Planes per workgroup: 109 KB instead of 435.
// Both in a chain of LAUNCHES dependent launches (the bench's graph: layer l + 1 starts when layer l has ended) with a row
// gather of GATHER_LINES random 256-byte reads per workgroup behind the hash (the K / V rows of a layer: they push the
// planes out of the L2 between launches, and a late workgroup's plane loads queue behind its neighbours' rows).
// Prints, per variant: microseconds from a workgroup's start to "sign bits of my head in LDS" (median / p90 / max over
// workgroups, last launch of the chain) and the time per launch.
//   hipcc --offload-arch=gfx950 -O3 -o quad_hash_probe quad_hash_probe.hip && ./quad_hash_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int D = 128, KL = 1700, KLPAD = 1728, TILES = (KL + 31) / 32;   // 54 tiles of 32 planes
constexpr int HEADS = 256, THREADS = 1024, WAVES = 16;
constexpr int GATHER_LINES = 2048;                                        // x 256 B = 512 KB per workgroup (cfg 3: ~510 tokens x 512 B)

__device__ __forceinline__ void dot8(float& acc, const u32x4& a, const u32x4& b) {
    asm("v_dot2c_f32_bf16 %0, %1, %5\n\tv_dot2c_f32_bf16 %0, %2, %6\n\tv_dot2c_f32_bf16 %0, %3, %7\n\tv_dot2c_f32_bf16 %0, %4, %8"
        : "+v"(acc)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

__device__ __forceinline__ void gather_phase(const u32x4* __restrict__ rows, size_t nrows16, uint32_t seed, uint32_t& acc) {
    // every thread: GATHER_LINES / THREADS x 16 lanes-of-a-row reads (a row = 256 B = 16 lanes x 16 B), all in flight
    const int lane16 = threadIdx.x & 15, grp = threadIdx.x >> 4;         // 64 row groups per workgroup
    u32x4 v[GATHER_LINES / 64];
#pragma unroll
    for (int i = 0; i < GATHER_LINES / 64; ++i) {
        uint32_t x = seed * 2654435761u + (uint32_t)(grp * 131 + i * 8191 + blockIdx.x * 524287);
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
        v[i] = __builtin_nontemporal_load(rows + ((size_t)x % nrows16) * 16 + lane16);
    }
#pragma unroll
    for (int i = 0; i < GATHER_LINES / 64; ++i) acc += v[i].x ^ v[i].w;
}

// today's prologue: planes chunk-major Wk[D/8][KLPAD] (16 bytes per (chunk, column))
__global__ __launch_bounds__(THREADS) void hash_valu(const uint16_t* __restrict__ q, const u32x4* __restrict__ Wk,
                                                     const u32x4* __restrict__ rows, size_t nrows16, uint32_t seq,
                                                     uint32_t* __restrict__ out, unsigned long long* __restrict__ stamp) {
    __shared__ u32x4 s_q[16];
    __shared__ uint32_t s_bits[2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long t0 = wall_clock64();
    if (tid < 16) s_q[tid] = reinterpret_cast<const u32x4*>(q + (size_t)blockIdx.x * D)[tid];
    u32x4 w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = Wk[tid + (size_t)i * KLPAD];
    __syncthreads();
    for (int c0 = 0; c0 < KL; c0 += THREADS) {
        const int cn = c0 + THREADS + tid < KLPAD ? c0 + THREADS + tid : KLPAD - 1;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dot8(acc, s_q[i], w[i]);
            if (c0 + THREADS < KL) w[i] = Wk[cn + (size_t)i * KLPAD];
        }
        asm("s_nop 3" : "+v"(acc));
        const unsigned long long bm = __ballot(acc > 0.f);
        if (lane == 0) {
            s_bits[(c0 >> 5) + wave * 2] = (uint32_t)bm;
            s_bits[(c0 >> 5) + wave * 2 + 1] = (uint32_t)(bm >> 32);
        }
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    uint32_t acc = s_bits[tid & 63] + seq;
    gather_phase(rows, nrows16, seq * 977u + s_bits[1], acc);
    out[(size_t)blockIdx.x * THREADS + tid] = acc;
    if (tid == 0) { stamp[blockIdx.x * 2] = t0; stamp[blockIdx.x * 2 + 1] = t1; }
}

// quad MFMA: planes plane-major Wt[KLPAD][D] bf16; xw[HEADS][TILES] tagged words
__global__ __launch_bounds__(THREADS) void hash_quad(const uint16_t* __restrict__ q, const uint16_t* __restrict__ Wt,
                                                     unsigned long long* __restrict__ xw, const u32x4* __restrict__ rows,
                                                     size_t nrows16, uint32_t seq, uint32_t* __restrict__ out,
                                                     unsigned long long* __restrict__ stamp) {
    __shared__ uint32_t s_bits[64];
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long t0 = wall_clock64();
    const int b = blockIdx.x, m = (b & 31) >> 3;                 // member 0 .. 3 of the quad
    const int h0 = (b & ~31) + (b & 7);                          // heads h0 + 8 i, i < 4: one XCD
    if (tid == 0) s_ok = 1;
    // B operand: query row n = lane % 32 (rows 4 .. 31: zero), k-chunk (lane / 32) * 8 of every 16-wide step
    const int n = lane & 31, kh = lane >> 5;
    const int tile = m + 4 * wave;                               // 14 waves have one
    f32x16 c = {0};
    if (tile < TILES) {
        const uint16_t* qrow = q + (size_t)(h0 + 8 * (n & 3)) * D;
        const uint16_t* wrow = Wt + (size_t)(tile * 32 + n) * D;
        u32x4 av[8], bv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {                            // all loads of the tile first: 8 KB of planes per wave
            av[s] = *reinterpret_cast<const u32x4*>(wrow + s * 16 + kh * 8);
            bv[s] = n < 4 ? *reinterpret_cast<const u32x4*>(qrow + s * 16 + kh * 8) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[s]), __builtin_bit_cast(bf16x8, bv[s]), c, 0, 0, 0);
        // lane (n, kh) holds C[plane (i / 4) * 8 + kh * 4 + i % 4][row n], i < 16: 16 sign bits; the other 16 sit in lane n + 32
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) bits |= (c[i] > 0.f ? 1u : 0u) << ((i / 4) * 8 + kh * 4 + (i % 4));
        bits |= (uint32_t)__shfl_xor((int)bits, 32);
        if (lane < 4) {                                          // row `lane`'s word of this tile -> that row's head
            const unsigned long long word = ((unsigned long long)seq << 32) | bits;
            __hip_atomic_store(xw + (size_t)(h0 + 8 * lane) * TILES + tile, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < TILES) {                                           // my head's words: until every one carries this launch's number
        unsigned long long v = 0;
        bool ok = false;
        for (int it = 0; it < 4096 && !ok; ++it) {
            v = __hip_atomic_load(xw + (size_t)b * TILES + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (uint32_t)(v >> 32) == seq;
        }
        s_bits[tid] = (uint32_t)v;
        if (!ok) s_ok = 0;
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    uint32_t acc = s_bits[tid & 31] + (uint32_t)s_ok;
    gather_phase(rows, nrows16, seq * 977u + s_bits[1], acc);
    out[(size_t)blockIdx.x * THREADS + tid] = acc;
    if (tid == 0) { stamp[blockIdx.x * 2] = t0; stamp[blockIdx.x * 2 + 1] = s_ok ? t1 : 0ull; }
}

int main(int argc, char** argv) {
    const int LAUNCHES = argc > 1 ? atoi(argv[1]) : 30, REPS = 20;
    uint16_t *q, *Wt; u32x4 *Wk, *rows; unsigned long long *xw, *st; uint32_t* out;
    const size_t nrows16 = (size_t)1 << 22;                                   // 4 M rows of 256 B = 1 GiB
    hipMalloc(&q, HEADS * D * 2); hipMalloc(&Wt, (size_t)KLPAD * D * 2); hipMalloc(&Wk, (size_t)KLPAD * D * 2);
    hipMalloc(&rows, nrows16 * 256); hipMalloc(&xw, (size_t)HEADS * TILES * 8); hipMalloc(&st, HEADS * 16);
    hipMalloc(&out, (size_t)HEADS * THREADS * 4);
    std::vector<uint16_t> hq(HEADS * D), hw((size_t)KLPAD * D);
    srand(7);
    for (auto& x : hq) x = (uint16_t)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));   // bf16 of magnitude ~0.5 .. 1, random sign
    for (auto& x : hw) x = (uint16_t)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
    hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(Wt, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(Wk, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);           // (the probe does not compare codes: any layout)
    hipMemset(rows, 1, nrows16 * 256); hipMemset(xw, 0, (size_t)HEADS * TILES * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    uint32_t seq = 1;
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<double> med, p90, mx, per;
        for (int rep = 0; rep < REPS; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < LAUNCHES; ++l, ++seq) {
                if (variant == 0) hipLaunchKernelGGL(hash_valu, dim3(HEADS), dim3(THREADS), 0, 0, q, Wk, rows, nrows16, seq, out, st);
                else hipLaunchKernelGGL(hash_quad, dim3(HEADS), dim3(THREADS), 0, 0, q, Wt, xw, rows, nrows16, seq, out, st);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(HEADS * 2);
            hipMemcpy(h.data(), st, HEADS * 16, hipMemcpyDeviceToHost);
            std::vector<double> d;
            int lost = 0;
            for (int b = 0; b < HEADS; ++b) { if (h[2 * b + 1] == 0) ++lost; else d.push_back((double)(h[2 * b + 1] - h[2 * b]) * 0.01); }
            std::sort(d.begin(), d.end());
            if (rep >= 2 && !d.empty()) { med.push_back(d[d.size() / 2]); p90.push_back(d[d.size() * 9 / 10]); mx.push_back(d.back()); per.push_back(ms * 1e3 / LAUNCHES); }
            if (lost) printf("  (variant %d rep %d: %d workgroups timed out)\n", variant, rep, lost);
        }
        auto mid = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("%-5s start -> sign bits of my head in LDS: median %.2f us  p90 %.2f  max %.2f   (last of %d dependent launches; "
               "%d workgroups x %d threads);  %.2f us per launch incl. a %d KB row gather per workgroup\n",
               variant ? "quad" : "valu", mid(med), mid(p90), mid(mx), LAUNCHES, HEADS, THREADS, mid(per), GATHER_LINES / 4);
    }
    return 0;
}
