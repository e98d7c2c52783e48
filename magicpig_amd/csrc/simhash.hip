// simhash.hip -- SimHash projection on the MFMA matrix cores (gfx950).
//
// Replaces models/attnserver.py:264-270 (query side) and :159-168 (key side):
//   S = x @ hash_func   (bf16 x bf16, f32 accumulate)  ->  bit = S > 0  ->  K-bit codes.
// The one dense contraction of the path: v_mfma_f32_32x32x16_bf16, A = 32 rows of x staged
// in LDS, B = pre-transposed hyperplanes Wt[K*L][D] read as one 16-byte load per lane.
//
// Bit-exactness: the sign of an f32-accumulated dot product is order-dependent only when
// |S| is within rounding of zero.  Every |acc| <= EPS * ||x|| * ||w|| (Cauchy-Schwarz bound
// on sum|x_i w_i|, EPS far above the f32 accumulation error) is recomputed exactly in f64
// (products of two bf16 are exact in f64), so the emitted bit is the exact sign, which is
// the order-independent definition the parity tests check against.
#include "common.h"

namespace mp {

constexpr int SH_THREADS = 256;          // 4 waves
constexpr int SH_ROWS = 32;              // MFMA M
constexpr int SH_MAXD = 256;             // max head_dim (LDS row)
constexpr int SH_LDS_STRIDE = SH_MAXD + 8;  // +16 B pad: conflict-free ds_read_b128 across rows
constexpr int SH_MAX_TILES = 16;         // max 32-column tiles per workgroup
constexpr float SH_EPS = 1.0f / 4096.0f; // guard band (2^-12) relative to ||x||*||w||

// Transpose hash_func [D][KL] -> Wt [KLpad][D] (zero rows beyond KL) and column norms.
__global__ void simhash_prepare_kernel(const uint16_t* __restrict__ W, int D, int KL, int KLpad,
                                       uint16_t* __restrict__ Wt, float* __restrict__ wnorm) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= KLpad) return;
    double ss = 0.0;
    for (int d = 0; d < D; ++d) {
        uint16_t v = (n < KL) ? W[(int64_t)d * KL + n] : (uint16_t)0;
        Wt[(int64_t)n * D + d] = v;
        double x = (double)bf16_bits_to_f32(v);
        ss += x * x;
    }
    wnorm[n] = (float)sqrt(ss) * 1.000001f;  // rounded up: it is used as an upper bound
}

// MODE 0: query rows -- L2-normalise in bf16 exactly as torch does (attnserver.py:264-266),
//         codes int32 [R][L], optional qnorm f32 [R].
// MODE 1: key rows   -- no normalisation (attnserver.py:162), codes int16 [L][n] (transposed).
template <int MODE>
__global__ __launch_bounds__(SH_THREADS) void simhash_kernel(
    const uint16_t* __restrict__ x,      // [R][D] bf16
    const uint16_t* __restrict__ Wt,     // [KLpad][D] bf16
    const float* __restrict__ wnorm,     // [KLpad]
    int64_t R, int D, int K, int L, int tables_per_wg, int tiles_per_wg, int64_t ld_out,
    void* __restrict__ codes_out, float* __restrict__ qnorm, float* __restrict__ dbg_acc) {
    __shared__ __attribute__((aligned(16))) uint16_t s_x[SH_ROWS * SH_LDS_STRIDE];
    __shared__ float s_rn[SH_ROWS];
    __shared__ uint32_t s_bits[SH_ROWS][SH_MAX_TILES + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * SH_ROWS;
    const int table0 = blockIdx.x * tables_per_wg;
    const int col0 = table0 * K;
    const int KL = K * L;

    // ---- phase A: stage 32 rows (normalised for MODE 0) into LDS; 8 threads per row
    {
        const int row = tid >> 3, part = tid & 7;
        const int64_t gr = r0 + row;
        const int per = D >> 3;  // elements per thread (D multiple of 16 -> per multiple of 2)
        const uint16_t* src = x + gr * D + part * per;
        double ss = 0.0;
        if (gr < R)
            for (int i = 0; i < per; ++i) {
                double v = (double)bf16_bits_to_f32(src[i]);
                ss += v * v;  // exact: squares of bf16 fit f64, so the sum is order-free
            }
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 4);
        double ssn = 0.0;  // sum of squares of what goes to LDS (for the guard bound)
        if (MODE == 0) {
            const float nrm = (float)sqrt((double)(float)ss);     // == sqrtf(f32 sum), correctly rounded
            const float nb = bf16_bits_to_f32(f32_to_bf16_rne(nrm));  // torch: bf16 norm tensor
            if (qnorm != nullptr && blockIdx.x == 0 && part == 0 && gr < R) qnorm[gr] = nrm;
            for (int i = 0; i < per; ++i) {
                uint16_t o = 0;
                if (gr < R) {
                    // f32 division, correctly rounded (f64 divide + round: p >= 2q+2), then RNE to bf16
                    const float qv = (float)((double)bf16_bits_to_f32(src[i]) / (double)nb);
                    o = f32_to_bf16_rne(qv);
                }
                s_x[row * SH_LDS_STRIDE + part * per + i] = o;
                double v = (double)bf16_bits_to_f32(o);
                ssn += v * v;
            }
            ssn += __shfl_xor(ssn, 1);
            ssn += __shfl_xor(ssn, 2);
            ssn += __shfl_xor(ssn, 4);
        } else {
            for (int i = 0; i < per; ++i)
                s_x[row * SH_LDS_STRIDE + part * per + i] = (gr < R) ? src[i] : (uint16_t)0;
            ssn = ss;
        }
        if (part == 0) s_rn[row] = (float)sqrt(ssn) * 1.000001f;
    }
    __syncthreads();

    // ---- phase B: MFMA tiles, sign bits by ballot
    const int ksteps = D >> 4;
    for (int ct = wave; ct < tiles_per_wg; ct += SH_THREADS / 64) {
        const int n = col0 + ct * 32 + (lane & 31);       // this lane's hyperplane (B column)
        const uint16_t* wrow = Wt + (int64_t)n * D + (lane >> 5) * 8;
        const uint16_t* arow = s_x + (lane & 31) * SH_LDS_STRIDE + (lane >> 5) * 8;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int kk = 0; kk < ksteps; ++kk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + kk * 16);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(wrow + kk * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        const float wn = wnorm[n] * SH_EPS;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);  // C/D layout of 32x32 MFMA
            const float a = acc[i];
            bool bit = a > 0.f;
            if (dbg_acc != nullptr && r0 + row < R && n < KL) dbg_acc[(r0 + row) * KL + n] = a;
            // guard band: exact recomputation, wave-cooperative (rare)
            // (zero-padded planes / rows give acc == bound == 0: they are not candidates)
            const bool near = (n < KL) && (r0 + row < R) && fabsf(a) <= wn * s_rn[row];
            unsigned long long m = __ballot(near);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int rs = __shfl(row, src);
                const int ns = __shfl(n, src);
                double part = 0.0;
                for (int d = lane; d < D; d += 64)
                    part += (double)bf16_bits_to_f32(s_x[rs * SH_LDS_STRIDE + d]) *
                            (double)bf16_bits_to_f32(Wt[(int64_t)ns * D + d]);
                const double tot = wave_sum(part);
                if (lane == src) bit = tot > 0.0;
            }
            const unsigned long long bm = __ballot(bit);
            if (lane == 0) {
                s_bits[(i & 3) + 8 * (i >> 2)][ct] = (uint32_t)bm;
                s_bits[(i & 3) + 8 * (i >> 2) + 4][ct] = (uint32_t)(bm >> 32);
            }
        }
    }
    __syncthreads();

    // ---- phase C: K-bit pack (bit i of code l <- column l*K + i), coalesced stores
    const uint32_t kmask = (K >= 32) ? 0xffffffffu : ((1u << K) - 1u);
    for (int p = tid; p < SH_ROWS * tables_per_wg; p += SH_THREADS) {
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
    return D >= 16 && D <= SH_MAXD && (D % 16) == 0 && K >= 1 && K <= 15 && tiles <= SH_MAX_TILES;
}

hipError_t launch_simhash_prepare(const uint16_t* W, int D, int K, int L, uint16_t* Wt,
                                  float* wnorm, hipStream_t st) {
    const int KLpad = simhash_padded_cols(K, L);
    hipLaunchKernelGGL(simhash_prepare_kernel, dim3((KLpad + 255) / 256), dim3(256), 0, st, W, D,
                       K * L, KLpad, Wt, wnorm);
    return hipGetLastError();
}

hipError_t launch_simhash_query(const uint16_t* q, const uint16_t* Wt, const float* wnorm, int R,
                                int D, int K, int L, int32_t* codes, float* qnorm, float* dbg,
                                hipStream_t st) {
    int tp, tiles;
    simhash_geometry(K, tp, tiles);
    dim3 grid((L + tp - 1) / tp, (R + SH_ROWS - 1) / SH_ROWS);
    hipLaunchKernelGGL(simhash_kernel<0>, grid, dim3(SH_THREADS), 0, st, q, Wt, wnorm, (int64_t)R,
                       D, K, L, tp, tiles, (int64_t)L, (void*)codes, qnorm, dbg);
    return hipGetLastError();
}

// one kv head: keys [n][D] -> codes int16 [L][n]
hipError_t launch_simhash_keys(const uint16_t* keys, const uint16_t* Wt, const float* wnorm,
                               int64_t n, int D, int K, int L, int16_t* codes, hipStream_t st) {
    int tp, tiles;
    simhash_geometry(K, tp, tiles);
    const int64_t row_tiles = (n + SH_ROWS - 1) / SH_ROWS;
    // gridDim.y is limited to 65535: chunk the token axis
    const int64_t max_y = 65535;
    for (int64_t t0 = 0; t0 < row_tiles; t0 += max_y) {
        const int64_t ny = (row_tiles - t0 < max_y) ? (row_tiles - t0) : max_y;
        const int64_t off = t0 * SH_ROWS;
        dim3 grid((L + tp - 1) / tp, (unsigned)ny);
        // codes are [L][n]: row stride n (ld_out), pointers offset by `off` tokens
        hipLaunchKernelGGL(simhash_kernel<1>, grid, dim3(SH_THREADS), 0, st, keys + off * D, Wt,
                           wnorm, n - off, D, K, L, tp, tiles, n, (void*)(codes + off), (float*)nullptr,
                           (float*)nullptr);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace mp
