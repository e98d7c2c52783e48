// simhash.hip -- SimHash projection on the MFMA matrix cores (gfx950).
//
// Replaces models/attnserver.py:264-270 (query side) and :159-168 (key side):
//   S = x @ hash_func   (bf16 x bf16, f32 accumulate)  ->  bit = S > 0  ->  K-bit codes.
// The one dense contraction of the path: v_mfma_f32_32x32x16_bf16, A = 32 rows of x staged
// in LDS, B = pre-transposed hyperplanes Wt[K*L][D] read as one 16-byte load per lane.
//
// Bit-exactness: the sign of an f32-accumulated dot product is order-dependent only when
// |S| is within rounding of zero.  Every |acc| <= 2^-16 * ||x|| * ||w|| (Cauchy-Schwarz bound
// on sum|x_i w_i|, EPS far above the f32 accumulation error) is recomputed exactly in f64
// (products of two bf16 are exact in f64), so the emitted bit is the exact sign, which is
// the order-independent definition the parity tests check against.
#include "common.h"

namespace mp {

constexpr int SH_ROWS = 32;              // MFMA M
constexpr int SH_MAX_TILES = 16;         // max 32-column tiles per workgroup
constexpr float SH_EPS = 1.0f / 65536.0f; // guard band (2^-16) relative to ||x||*||w||: 2x the worst-case
                                          // bound of 128 f32 roundings (2^-17), 270x the measured MFMA
                                          // accumulation error (2^-24.1, tests/test_gpu_parity.py)

// Transpose hash_func [D][KL] -> Wt [KLpad][D] (zero rows beyond KL), the chunk-major copy
// Wk [D/8][KLpad][8] used by the hash fused into the retrieve kernel, and column norms.
__global__ void simhash_prepare_kernel(const uint16_t* __restrict__ W, int D, int KL, int KLpad,
                                       uint16_t* __restrict__ Wt, uint16_t* __restrict__ Wk,
                                       float* __restrict__ wnorm) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= KLpad) return;
    double ss = 0.0;
    for (int d = 0; d < D; ++d) {
        uint16_t v = (n < KL) ? W[(int64_t)d * KL + n] : (uint16_t)0;
        Wt[(int64_t)n * D + d] = v;
        Wk[((int64_t)(d >> 3) * KLpad + n) * 8 + (d & 7)] = v;
        double x = (double)bf16_bits_to_f32(v);
        ss += x * x;
    }
    wnorm[n] = (float)sqrt(ss) * 1.000001f;  // rounded up: it is used as an upper bound
}

// MODE 0: query rows -- L2-normalise in bf16 exactly as torch does (attnserver.py:264-266),
//         codes int32 [R][L], optional qnorm f32 [R].
// MODE 1: key rows   -- no normalisation (attnserver.py:162), codes int16 [L][n] (transposed).
// Block = 64 * max(4, tiles_per_wg) threads: wave w owns column tile w of the workgroup's span
// and prefetches its B fragments (the hyperplanes) before the rows are staged, so the HBM/L2
// latency of the planes overlaps the normalisation.
// MINW: minimum waves per SIMD the register allocation must allow.  The key side is a chain of
// short dependent phases per workgroup (load 8 KB -> 8 MFMAs -> ballots -> pack): its throughput is
// set by how many workgroups a CU can interleave, so it is compiled for 8 waves per SIMD (<= 64 VGPRs).
template <int MODE, int D, int MINW>
__global__ __launch_bounds__(1024, MINW) void simhash_kernel(
    const uint16_t* __restrict__ x,      // [R][D] bf16
    const uint16_t* __restrict__ Wt,     // [KLpad][D] bf16
    const float* __restrict__ wnorm,     // [KLpad]
    int64_t R, int K, int L, int tables_per_wg, int tiles_per_wg, int64_t ld_out,
    void* __restrict__ codes_out, float* __restrict__ qnorm, float* __restrict__ dbg_acc,
    unsigned long long* __restrict__ stamp) {
    constexpr int KSTEPS = D / 16;
    constexpr int STRIDE = D + 8;        // +16 B pad: conflict-free ds_read_b128 across rows
    __shared__ __attribute__((aligned(16))) uint16_t s_x[SH_ROWS * STRIDE];
    __shared__ float s_rn[SH_ROWS];
    __shared__ uint32_t s_bits[SH_ROWS][SH_MAX_TILES + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * SH_ROWS;
    const int table0 = blockIdx.x * tables_per_wg;
    const int col0 = table0 * K;
    const int KL = K * L;
    MP_STAMP(stamp, 0);

    // ---- prefetch this wave's hyperplane fragments (B operand): 16 B per lane per k-step
    const bool has_tile = wave < tiles_per_wg;
    const int n = col0 + wave * 32 + (lane & 31);        // this lane's hyperplane (B column)
    bf16x8 bfrag[KSTEPS];
    float wn = 0.f;
    if (has_tile) {
        const uint16_t* wrow = Wt + (int64_t)n * D + (lane >> 5) * 8;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) bfrag[kk] = *reinterpret_cast<const bf16x8*>(wrow + kk * 16);
        wn = wnorm[n] * SH_EPS;
    }

    // ---- phase A: stage 32 rows (normalised for MODE 0) into LDS; 8 threads per row
    if (tid < SH_ROWS * 8) {
        constexpr int PER = D / 8;       // elements per thread: 8, 16 or 32
        const int row = tid >> 3, part = tid & 7;
        const int64_t gr = r0 + row;
        uint16_t e[PER];
        if (gr < R) {
            const u32x4* src = reinterpret_cast<const u32x4*>(x + gr * D + part * PER);
#pragma unroll
            for (int v = 0; v < PER / 8; ++v) {
                const u32x4 t = src[v];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e[v * 8 + 2 * i] = (uint16_t)(t[i] & 0xffffu);
                    e[v * 8 + 2 * i + 1] = (uint16_t)(t[i] >> 16);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < PER; ++i) e[i] = 0;
        }
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const double v = (double)bf16_bits_to_f32(e[i]);
            ss += v * v;                 // exact: squares of bf16 fit f64, so the sum is order-free
        }
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 4);
        double ssn = ss;                 // sum of squares of what goes to LDS (guard bound)
        if (MODE == 0) {
            const float nrm = (float)sqrt((double)(float)ss);         // == sqrtf(f32 sum), correctly rounded
            const float nb = bf16_bits_to_f32(f32_to_bf16_rne(nrm));  // torch: the norm is a bf16 tensor
            if (qnorm != nullptr && blockIdx.x == 0 && part == 0 && gr < R) qnorm[gr] = nrm;
            ssn = 0.0;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                // IEEE f32 division (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt),
                // then RNE to bf16 -- what torch's bf16 `q / norm` computes
                const float qv = __fdiv_rn(bf16_bits_to_f32(e[i]), nb);
                e[i] = (gr < R) ? f32_to_bf16_rne(qv) : (uint16_t)0;
                const double v = (double)bf16_bits_to_f32(e[i]);
                ssn += v * v;
            }
            ssn += __shfl_xor(ssn, 1);
            ssn += __shfl_xor(ssn, 2);
            ssn += __shfl_xor(ssn, 4);
        }
        u32x4* dst = reinterpret_cast<u32x4*>(s_x + row * STRIDE + part * PER);
#pragma unroll
        for (int v = 0; v < PER / 8; ++v) {
            u32x4 t;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                t[i] = (uint32_t)e[v * 8 + 2 * i] | ((uint32_t)e[v * 8 + 2 * i + 1] << 16);
            dst[v] = t;
        }
        if (part == 0) s_rn[row] = (float)sqrt(ssn) * 1.000001f;
    }
    __syncthreads();
    MP_STAMP(stamp, 1);

    // ---- phase B: one 32x32 MFMA tile per wave, sign bits by ballot
    if (has_tile) {
        const uint16_t* arow = s_x + (lane & 31) * STRIDE + (lane >> 5) * 8;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + kk * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[kk], acc, 0, 0, 0);
        }
        uint32_t nearmask = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);  // C/D layout of the 32x32 MFMA
            const float a = acc[i];
            if (dbg_acc != nullptr && r0 + row < R && n < KL) dbg_acc[(r0 + row) * KL + n] = a;
            // guard band candidates; zero-padded planes / rows (acc == bound == 0) are not
            const bool near = (n < KL) && (r0 + row < R) && fabsf(a) <= wn * s_rn[row];
            nearmask |= (uint32_t)near << i;
            const unsigned long long bm = __ballot(a > 0.f);
            if (lane == 0) {
                s_bits[(i & 3) + 8 * (i >> 2)][wave] = (uint32_t)bm;
                s_bits[(i & 3) + 8 * (i >> 2) + 4][wave] = (uint32_t)(bm >> 32);
            }
        }
        // exact fix-up of the candidates (rare, rolled loops: keeps the hot code small).  Column
        // n's k-values are split between lanes l and l^32, so every flagged element is redone by
        // its own lane pair: f64 products of bf16 pairs are exact, one shuffle joins the halves,
        // and the lane patches its bit in LDS.
        if (__ballot(nearmask != 0)) {
            const uint16_t* wcol = Wt + (int64_t)n * D + (lane >> 5) * 8;
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const bool near = (nearmask >> i) & 1u;
                if (!__ballot(near)) continue;                                   // wave-uniform
                const bool partner_near = __shfl_xor((int)near, 32) != 0;
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const int prow = (i & 3) + 8 * (i >> 2) + 4 * ((lane >> 5) ^ 1);  // partner's row
                double mine = 0.0, forp = 0.0;
                if (near || partner_near) {
                    const uint16_t* xm = s_x + row * STRIDE + (lane >> 5) * 8;
                    const uint16_t* xp = s_x + prow * STRIDE + (lane >> 5) * 8;
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) {
                        const u32x4 w = *reinterpret_cast<const u32x4*>(wcol + kk * 16);
                        const u32x4 a = *reinterpret_cast<const u32x4*>(xm + kk * 16);
                        const u32x4 p = *reinterpret_cast<const u32x4*>(xp + kk * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const double wl = (double)bf16_lo(w[j]), wh = (double)bf16_hi(w[j]);
                            mine += (double)bf16_lo(a[j]) * wl + (double)bf16_hi(a[j]) * wh;
                            forp += (double)bf16_lo(p[j]) * wl + (double)bf16_hi(p[j]) * wh;
                        }
                    }
                }
                const double other = __shfl_xor(forp, 32);                       // partner's half for MY row
                if (near) {
                    uint32_t* wd = &s_bits[row][wave];
                    const uint32_t bmask = 1u << (lane & 31);
                    if (mine + other > 0.0) atomicOr(wd, bmask); else atomicAnd(wd, ~bmask);
                }
            }
        }
    }
    __syncthreads();
    MP_STAMP(stamp, 2);

    // ---- phase C: K-bit pack (bit i of code l <- column l*K + i), coalesced stores
    const uint32_t kmask = (1u << K) - 1u;
    for (int p = tid; p < SH_ROWS * tables_per_wg; p += blockDim.x) {
        int row, tb;
        if (MODE == 0) { row = p / tables_per_wg; tb = p % tables_per_wg; }   // codes[r][l]: l fastest
        else           { tb = p / SH_ROWS;        row = p % SH_ROWS; }        // codes[l][t]: t fastest
        const int l = table0 + tb;
        const int64_t gr = r0 + row;
        if (l >= L || gr >= R) continue;
        const int bp = tb * K, w = bp >> 5, sh = bp & 31;
        uint32_t v = s_bits[row][w] >> sh;
        if (sh + K > 32) v |= s_bits[row][w + 1] << (32 - sh);
        v &= kmask;
        if (MODE == 0) reinterpret_cast<int32_t*>(codes_out)[gr * ld_out + l] = (int32_t)v;
        else           reinterpret_cast<int16_t*>(codes_out)[(int64_t)l * ld_out + gr] = (int16_t)v;
    }
    MP_STAMP(stamp, 3);
}

// ---------------------------------------------------------------- host launchers

// tables per workgroup: the column span tables*K must be a multiple of 32 (whole MFMA tiles)
// and give every wave at least one tile when possible.
static void simhash_geometry(int K, int& tables_per_wg, int& tiles_per_wg) {
    int g = 32, k = K;
    while (k) { int t = g % k; g = k; k = t; }   // g = gcd(32, K)
    int tp = 32 / g;                             // minimal tables so that tp*K % 32 == 0
    int tiles = tp * K / 32;
    while (tiles < 4 && tiles * 2 <= SH_MAX_TILES) { tp *= 2; tiles *= 2; }
    tables_per_wg = tp;
    tiles_per_wg = tiles;
}

int simhash_padded_cols(int K, int L) {
    int tp, tiles;
    simhash_geometry(K, tp, tiles);
    const int wgs = (L + tp - 1) / tp;
    return wgs * tiles * 32;
}

int simhash_supported(int D, int K) {
    int tp, tiles;
    simhash_geometry(K, tp, tiles);
    return (D == 64 || D == 128 || D == 256) && K >= 1 && K <= 15 && tiles <= SH_MAX_TILES;
}

hipError_t launch_simhash_prepare(const uint16_t* W, int D, int K, int L, uint16_t* Wt,
                                  uint16_t* Wk, float* wnorm, hipStream_t st) {
    const int KLpad = simhash_padded_cols(K, L);
    hipLaunchKernelGGL(simhash_prepare_kernel, dim3((KLpad + 255) / 256), dim3(256), 0, st, W, D,
                       K * L, KLpad, Wt, Wk, wnorm);
    return hipGetLastError();
}

unsigned long long* g_stamp = nullptr;   // debug phase-timestamp sink (mp_debug_set_stamp_buffer)

template <int MODE>
static hipError_t launch_simhash_t(const uint16_t* x, const uint16_t* Wt, const float* wnorm,
                                   int64_t R, int D, int K, int L, int64_t ld_out, void* codes,
                                   float* qnorm, float* dbg, unsigned grid_y, hipStream_t st) {
    int tp, tiles;
    simhash_geometry(K, tp, tiles);
    dim3 grid((L + tp - 1) / tp, grid_y);
    dim3 block(64 * (tiles > 4 ? tiles : 4));
    unsigned long long* stamp = (MODE == 0) ? g_stamp : nullptr;
#define MP_SH_CASE(DD)                                                                          \
    if (D == DD) {                                                                              \
        hipLaunchKernelGGL((simhash_kernel<MODE, DD, (MODE == 1 ? 2 : 1)>), grid, block, 0, st, x, Wt, wnorm, R, K, L, \
                           tp, tiles, ld_out, codes, qnorm, dbg, stamp);                        \
        return hipGetLastError();                                                               \
    }
    MP_SH_CASE(128)
    MP_SH_CASE(64)
    MP_SH_CASE(256)
#undef MP_SH_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_simhash_query(const uint16_t* q, const uint16_t* Wt, const float* wnorm, int R,
                                int D, int K, int L, int32_t* codes, float* qnorm, float* dbg,
                                hipStream_t st) {
    return launch_simhash_t<0>(q, Wt, wnorm, R, D, K, L, L, (void*)codes, qnorm, dbg,
                               (unsigned)((R + SH_ROWS - 1) / SH_ROWS), st);
}

// one kv head: keys [n][D] -> codes int16 [L][n]
hipError_t launch_simhash_keys(const uint16_t* keys, const uint16_t* Wt, const float* wnorm,
                               int64_t n, int D, int K, int L, int16_t* codes, hipStream_t st) {
    const int64_t row_tiles = (n + SH_ROWS - 1) / SH_ROWS;
    const int64_t max_y = 65535;   // gridDim.y limit: chunk the token axis
    for (int64_t t0 = 0; t0 < row_tiles; t0 += max_y) {
        const int64_t ny = (row_tiles - t0 < max_y) ? (row_tiles - t0) : max_y;
        const int64_t off = t0 * SH_ROWS;
        // codes are [L][n]: row stride n (ld_out), pointers offset by `off` tokens
        hipError_t e = launch_simhash_t<1>(keys + off * D, Wt, wnorm, n - off, D, K, L, n,
                                           (void*)(codes + off), nullptr, nullptr, (unsigned)ny, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace mp
