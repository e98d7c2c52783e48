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

// Query rows (the standalone form of attnserver.py:264-270; the decode path uses the hash fused
// into the retrieve kernel): L2-normalise in bf16 exactly as torch does, codes int32 [R][L],
// optional qnorm f32 [R].  One workgroup per 32 rows x one span of whole tables.
// Block = 64 * max(4, tiles_per_wg) threads: wave w owns column tile w of the workgroup's span
// and prefetches its B fragments (the hyperplanes) before the rows are staged, so the HBM/L2
// latency of the planes overlaps the normalisation.
template <int D>
__global__ __launch_bounds__(1024) void simhash_query_kernel(
    const uint16_t* __restrict__ x,      // [R][D] bf16
    const uint16_t* __restrict__ Wt,     // [KLpad][D] bf16
    const float* __restrict__ wnorm,     // [KLpad]
    int64_t R, int K, int L, int tables_per_wg, int tiles_per_wg,
    int32_t* __restrict__ codes_out, float* __restrict__ qnorm, float* __restrict__ dbg_acc,
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

    // ---- phase A: stage 32 normalised rows into LDS; 8 threads per row
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
        double ssn = 0.0;                // sum of squares of what goes to LDS (guard bound)
        {
            const float nrm = (float)sqrt((double)(float)ss);         // == sqrtf(f32 sum), correctly rounded
            const float nb = bf16_bits_to_f32(f32_to_bf16_rne(nrm));  // torch: the norm is a bf16 tensor
            if (qnorm != nullptr && blockIdx.x == 0 && part == 0 && gr < R) qnorm[gr] = nrm;
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
        const int row = p / tables_per_wg, tb = p % tables_per_wg;          // codes[r][l]: l fastest
        const int l = table0 + tb;
        const int64_t gr = r0 + row;
        if (l >= L || gr >= R) continue;
        const int bp = tb * K, w = bp >> 5, sh = bp & 31;
        uint32_t v = s_bits[row][w] >> sh;
        if (sh + K > 32) v |= s_bits[row][w + 1] << (32 - sh);
        v &= kmask;
        codes_out[gr * L + l] = (int32_t)v;
    }
    MP_STAMP(stamp, 3);
}

// ---------------------------------------------------------------- key hashing at prefill
// models/attnserver.py:159-168: keys [heads][n][D] -> codes int16 [heads][L][n]; the one place of
// the path where MFMA throughput matters (37.6 GFLOP per kv head at cfg 1).
//
// What bounded it through round 4 (19 % of the dense bf16 peak) was neither the matrix pipe nor, as believed, VALU issue
// alone: every wave read the whole 32-row x D tile out of LDS (1 KB per MFMA) to multiply it with its 32 planes --
// 128 B per clock and CU, which IS the LDS's peak rate; at the ~55 % of it that ds_read_b128 sustains next to the
// staging stores the matrix pipe cannot get past a quarter of its peak.  Round 5:
//   * REGISTER BLOCKING: a wave keeps the fragments of SETS x 32 planes (SETS = 4 at head_dim <= 128: 128 VGPRs) and every
//     fragment of the row tile it reads from LDS feeds two MFMAs on independent accumulators -- 512 B of LDS per MFMA,
//     SETS column tiles per barrier.
//     Workgroups of four waves at two waves per SIMD (256 registers each); a workgroup owns floor(128 SETS / K) tables;
//   * the product is taken TRANSPOSED -- C' = W X^T: lane <-> row of X, accumulator register <-> plane -- so a lane holds
//     16 sign bits of ITS OWN row: they are packed per lane (one v_alignbit per accumulator, a byte permute, one
//     v_permlane32_swap to join the two half-waves) instead of 16 ballots + 32 v_writelane transposing through SGPRs;
//   * the guard band is tested against the lane's own row norm, not the tile's largest; the rows' sums of squares come
//     from v_dot2c (4 instructions per 16-byte chunk instead of 8 converts + 8 FMAs).
// As before: the rows of tile t+2 are requested while tile t is on the matrix pipe and tile t+1 is written to the other
// half of a double-buffered LDS stage (ONE barrier per tile, ordering LDS only); the tile's sign words go into the chunk's
// bit matrix [32 * chunk_tiles rows][tiles + 1 words], from which the K-bit codes are cut at the end and stored as long
// runs of every [L][n] row; guard-band candidates are queued and resolved together at the end (exact f64 dot product,
// one 16-lane group per candidate), patching the bit matrix before the codes are cut out.
constexpr int SK_CH_MAX = 32;             // 32-row tiles per workgroup, at most (keys_chunk_tiles picks; LDS caps it)
constexpr int SK_QCAP = 1024;             // deferred exact-sign candidates per workgroup
constexpr int SK_WAVES = 4;               // waves per workgroup; two workgroups per CU = two waves per SIMD
constexpr int SK_MAX_SETS = 4;            // plane sets (32 planes each) per wave, at most
constexpr int SK_MAX_TABLES = 64;
__host__ __device__ constexpr int sk_sets(int D) { return D <= 128 ? 4 : 2; }   // 128 VGPRs of plane fragments either way

template <int D, bool NORMS = false>     // NORMS: the rows' norms are given (the attention store's kn: bf16-rounded, within 2^-9)
__global__ __launch_bounds__(64 * SK_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void simhash_keys_kernel(
    const uint16_t* __restrict__ x,       // bf16 (centred keys): row r of head h at x + h*head_stride + r*row_stride
    int64_t head_stride, int64_t row_stride,   // elements: n*D, D for keys [heads][n][D]; M*2D, 2D for the K|V store
    const uint16_t* __restrict__ Wt,      // [KLpad][D]
    const float* __restrict__ wnorm,      // [KLpad]
    int64_t n, int K, int L, int tables_per_wg, int tiles_per_wg, int chunk_tiles, int wgs_x, int chunks,
    int heads, int groups,                    // persistent: `groups` groups of wgs_x workgroups per XCD walk the units
    const float* __restrict__ rnorm, int64_t rnorm_stride,   // NORMS: row r of head h has norm rnorm[h * rnorm_stride + r]
    int16_t* __restrict__ codes,              // [heads][L][n]
    unsigned long long* __restrict__ stamp) {
    constexpr int KSTEPS = D / 16;
    constexpr int STRIDE = D + 8;
    constexpr int CPR = D / 8;            // 16-byte chunks per row
    constexpr int SETS = sk_sets(D);
    constexpr int NCH = SH_ROWS * CPR;    // chunks per 32-row tile
    constexpr int NTHR = 64 * SK_WAVES;
    constexpr int QN = NCH / NTHR;        // chunks per thread
    static_assert(NCH % NTHR == 0 && QN >= 1, "a tile's chunks divide over the threads");
    const int CROWS = chunk_tiles * SH_ROWS;
    __shared__ __attribute__((aligned(16))) uint16_t s_x[2][SH_ROWS * STRIDE];
    __shared__ float s_rn[2][SH_ROWS];
    __shared__ __attribute__((aligned(16))) float s_rnall[NORMS ? SK_CH_MAX * SH_ROWS : 1];    // NORMS: the unit's row norms
    constexpr int BW = SK_WAVES * SETS + 1;       // words of the sign matrix per row: one per column tile + one of slack
    __shared__ uint32_t s_queue[SK_QCAP];
    __shared__ int s_qn;
    extern __shared__ uint32_t s_rowbits[];               // sign matrix of the whole chunk: [CROWS][BW] (bit c of
                                                          // word w of row r <- plane col0 + 32 w + c)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nthr = NTHR;
    // 1-D grid, XCD-aware: block b runs on XCD b % 8.  The wgs_x workgroups that hash the SAME rows
    // against different plane spans are given consecutive slots of ONE XCD (b = (q * wgs_x + j) * 8 +
    // xcd for row chunk 8q + xcd), so the rows come from HBM once and from that XCD's L2 wgs_x - 1
    // times; with column-fastest numbering they landed on wgs_x different XCDs and HBM served every
    // copy (PMC: 1.29 GB read per layer at cfg 1 for 0.2 GB of keys).
    // Round 6: the workgroups are PERSISTENT (grid = 8 XCDs x groups x wgs_x <= two per CU): a workgroup keeps its plane span
    // -- 128 VGPRs of fragments per wave, loaded once -- and walks the units (kv head, row chunk) group, group + groups, ...
    // of its XCD; the first two row tiles of the next unit are requested before the current one is flushed.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int colgrp = slot % wgs_x, group = slot / wgs_x;
    const int64_t units = (int64_t)chunks * heads;
    const int64_t unit_step = (int64_t)groups * 8;
    int64_t unit = (int64_t)group * 8 + xcd;                  // (kv head, row chunk), chunk fastest
    if (unit >= units) return;
    const int table0 = colgrp * tables_per_wg;
    const int col0 = table0 * K, KL = K * L;   // col0 is NOT tile aligned: Wt is row-per-plane, any start works
    const uint16_t* xh = x + (int64_t)(unit / chunks) * head_stride;          // the unit whose tiles are being LOADED
    const float* rnh = NORMS ? rnorm + (int64_t)(unit / chunks) * rnorm_stride : nullptr;
    int64_t row_base = (int64_t)(unit % chunks) * CROWS;
    MP_STAMP(stamp, 40);

    // column tile wave * SETS + st of the workgroup's span: planes col0 + 32 (wave SETS + st) + (lane & 31).  Tiles past
    // tiles_per_wg (a K whose span does not fill all sixteen) multiply zero rows of Wt's padding: their words are never read.
    bf16x8 bfrag[SETS][KSTEPS];
    float wcoarse[SETS];                                  // uniform: the widest guard band among the tile's planes
    uint32_t mine[SETS];                                  // per lane: bit i = the plane of accumulator i is this workgroup's
    float wn_s[SETS];
    // every load of the prologue first (as one loop per tile the use of a tile's plane norm put an s_waitcnt vmcnt(0)
    // between the tiles' fragment loads: four dependent round trips, 8 us per workgroup under the stamps)
#pragma unroll
    for (int st = 0; st < SETS; ++st) {
        const int ctile = wave * SETS + st;
        const int ncol = col0 + ctile * 32 + (lane & 31);
        const bool live = ctile < tiles_per_wg;           // uniform
        const uint16_t* wrow = Wt + (int64_t)(live ? ncol : col0) * D + (lane >> 5) * 8;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) bfrag[st][kk] = *reinterpret_cast<const bf16x8*>(wrow + kk * 16);
        wn_s[st] = wnorm[live ? ncol : col0];             // (Wt and wnorm are padded to whole tiles of the last workgroup)
    }
#pragma unroll
    for (int st = 0; st < SETS; ++st) {
        const int ctile = wave * SETS + st;
        const int ncol = col0 + ctile * 32 + (lane & 31);
        const bool live = ctile < tiles_per_wg;           // uniform
        // planes past this workgroup's tables (the tail of its last tile) belong to the next one
        const float wn = (live && ncol < KL && ncol - col0 < tables_per_wg * K) ? wn_s[st] * SH_EPS : -1.f;
        // lane l < 32 knows whether plane l of the tile is ours; a lane's accumulator i is plane (i & 3) + 8 (i >> 2) +
        // 4 (lane >> 5): gather those 16 bits out of the tile's 32-bit validity word
        const uint32_t vw = (uint32_t)__ballot(wn >= 0.f);              // (both half-waves hold the same 32 planes)
        uint32_t mbits = 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) mbits |= ((vw >> ((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5))) & 1u) << i;
        mine[st] = mbits;
        const float w16 = row16_max_nonneg(fmaxf(wn, 0.f));
        const float wmax = __int_as_float(max(__builtin_amdgcn_readlane(__float_as_int(w16), 0),
                                              __builtin_amdgcn_readlane(__float_as_int(w16), 16)));
        wcoarse[st] = live ? wmax : -1.f;
    }

    // Loads are unconditional (row index clamped to the last row): a load under a branch is followed by
    // s_waitcnt vmcnt(0) at the join, which would serialise the prefetch.  Rows past n compute on duplicates
    // whose results are never stored.  Thread tid takes chunk tid % CPR of rows tid / CPR + q * (nthr / CPR).
    const int64_t last_row = n - 1;
    auto tile_load = [&](int t, u32x4 (&sreg)[QN]) {
        // a uniform base (the unit's first row) + a 32-bit element offset per lane: a unit is <= 1 024 rows of <= 512 elements
        const uint16_t* ubase = xh + row_base * row_stride;
        const int lim = (int)(last_row - row_base);                       // last row of the head, relative to the unit
        // (the row of this thread inside a tile, recomputed per call behind an opaque move: hoisted out of the unit loop its
        // variants -- + 16, + 32 per q and tile parity -- lived in registers across the whole tile loop and were spilled)
        int rt = tid / CPR;
        asm volatile("" : "+v"(rt));
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            int r = t * SH_ROWS + rt + q * (NTHR / CPR);
            r = r < lim ? r : lim;
            const uint32_t off = (uint32_t)r * (uint32_t)row_stride + (uint32_t)((tid % CPR) * 8);
            sreg[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ubase + off));
        }
    };
    auto tile_store = [&](int buf, const u32x4 (&sreg)[QN]) {
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int row = (tid / CPR) + q * (NTHR / CPR);
            *reinterpret_cast<u32x4*>(&s_x[buf][row * STRIDE + (tid % CPR) * 8]) = sreg[q];
            if constexpr (NORMS) continue;           // (the unit's norms are in s_rnall: below)
            float ss = 0.f;
            dot8_bf16_chain(ss, sreg[q], sreg[q]);        // sum of squares of the chunk's 8 elements
            dot_settle(ss);
            // the CPR chunks of a row sit in CPR consecutive lanes (nthr is a multiple of 64)
            if (CPR == 16) ss = row16_sum(ss);
            else if (CPR == 32) { ss = row16_sum(ss); ss += __shfl_xor(ss, 16); }
            else {
#pragma unroll
                for (int sft = 1; sft < CPR; sft <<= 1) ss += __shfl_xor(ss, sft);
            }
            if ((tid % CPR) == 0)              // upper bound of ||row|| (v_sqrt_f32 is good to 1 ulp)
                s_rn[buf][row] = __builtin_amdgcn_sqrtf(ss) * 1.0001f;
        }
    };
    // MFMA + signs of tile t (LDS half t & 1).  C' = W X^T: lane l holds, for row (l & 31) of the tile, the 16 planes
    // (i & 3) + 8 (i >> 2) + 4 (l >> 5), i = 0 .. 15, of a column tile's 32.
    // signs + guard band of one column tile's accumulators (the epilogue): ~45 vector instructions and one LDS store
    auto epilogue = [&](const f32x16& acc, int st, int t, float rn) {
        const int ctile = wave * SETS + st;
        // 16 sign bits of this lane's row: b = (b << 1) | sign, last accumulator first, then inverted (bit = acc > 0;
        // an exact zero is inside every guard band and is decided by the exact pass)
        uint32_t b = 0u;
        float amin = 3.0e38f;
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            b = __builtin_amdgcn_alignbit(b, __float_as_uint(acc[i]), 31);
            amin = fminf(amin, fabsf(acc[i]));
        }
        b = ~b;
        // nibble j of b = planes 8 j + 4 (lane >> 5) + (0 .. 3): nibbles to bytes, the upper half-wave 4 bits up
        const uint32_t lo = b & 0x0f0fu, hi = (b >> 4) & 0x0f0fu;
        uint32_t w = __builtin_amdgcn_perm(hi, lo, 0x05010400u) << ((lane >> 5) * 4);
        const auto sw = __builtin_amdgcn_permlane32_swap(w, w, false, false);   // the other half-wave's 16 planes
        w = sw[0] | sw[1];
        if (lane < 32) s_rowbits[(t * SH_ROWS + lane) * BW + ctile] = w;
        // Guard band, tested against the widest band of the tile's planes (plane norms differ by a few per cent: a
        // handful of candidates more, each decided exactly at the end).  NOT rare per wave: one lane in ~450 holds
        // a candidate, so one wave-tile in seven runs this block -- it must not wait for anything: round 4's form
        // read every plane's own band from LDS inside the loop (16 dependent LDS round trips, ~2 500 cycles per
        // execution, half a tile's time on average).  Which of a lane's 16 planes belong to this workgroup at all
        // (the last tile's tail, Wt's zero padding) is a per-lane bit mask made once, `mine`.
        const float thr = wcoarse[st] * rn;
        if (amin <= thr) {
            const int64_t gr = row_base + (int64_t)t * SH_ROWS + (lane & 31);
            const uint32_t ok = gr < n ? mine[st] : 0u;
            // ONE queue entry per (row, column tile, half-wave) with the 16 accumulators' candidate bits: straight-line
            // compares, one LDS atomic and one store per lane that holds any -- the per-candidate form below ran 16 divergent
            // branches with an atomic each, in a path that some wave of the workgroup takes in nine tiles of ten, in front of
            // the tile's barrier
            uint32_t cm = 0u;                      // bit i <- sign(thr - |acc[i]|), shifted in as the sign bits are above
#pragma unroll
            for (int i = 15; i >= 0; --i) cm = __builtin_amdgcn_alignbit(cm, __float_as_uint(thr - fabsf(acc[i])), 31);
            cm = ~cm & ok;
            if (cm) {
                const int slot = atomicAdd(&s_qn, 1);
                if (slot < SK_QCAP)
                    s_queue[slot] = ((uint32_t)(t * SH_ROWS + (lane & 31)) << 21) | ((uint32_t)ctile << 17) |
                                    ((uint32_t)(lane >> 5) << 16) | cm;
            }
        }
    };
    // MFMA + signs of tile t (LDS half t & 1).  C' = W X^T: lane l holds, for row (l & 31) of the tile, the 16 planes
    // (i & 3) + 8 (i >> 2) + 4 (l >> 5), i = 0 .. 15, of a column tile's 32.
    // The row tile's fragments are read from LDS ONCE (KSTEPS x 16 bytes per lane: 32 VGPRs) and the SETS column tiles
    // follow one another on two alternating accumulators: the eight MFMAs of tile st -- a dependent chain, each waiting
    // ~32 cycles for its predecessor -- are interleaved with the epilogue of tile st - 1 (sched_group_barrier: one MFMA,
    // then six vector instructions), so the vector pipe works in the matrix pipe's shadow instead of behind it.
    auto compute = [&](int t) {
        const int buf = t & 1;
        const uint16_t* brow = &s_x[buf][(lane & 31) * STRIDE + (lane >> 5) * 8];
        // (NORMS: an upper bound of ||row|| -- the store's norm is the bf16 rounding of the exact one)
        const float rn = NORMS ? s_rnall[t * SH_ROWS + (lane & 31)] * 1.005f : s_rn[buf][lane & 31];
        constexpr bool KEEP = KSTEPS <= 8;                 // head_dim 256: 64 registers of fragments do not fit next to the planes'
        bf16x8 xb[KEEP ? KSTEPS : 1];
        if constexpr (KEEP) {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) xb[kk] = *reinterpret_cast<const bf16x8*>(brow + kk * 16);
        }
        f32x16 acc[2];
#pragma unroll
        for (int st = 0; st <= SETS; ++st) {
            if (st < SETS) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[st & 1][i] = 0.f;
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const bf16x8 xk = KEEP ? xb[KEEP ? kk : 0] : *reinterpret_cast<const bf16x8*>(brow + kk * 16);
                    acc[st & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfrag[st][kk], xk, acc[st & 1], 0, 0, 0);
                }
            }
            if (st > 0) epilogue(acc[(st - 1) & 1], st - 1, t, rn);
            if (st > 0 && st < SETS) {
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA of tile st ...
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);     // ... then six VALU instructions of tile st - 1's epilogue
                }
            }
        }
    };
    const uint32_t kmask = (1u << K) - 1u;
    // __syncthreads() also drains every outstanding global load (s_waitcnt vmcnt(0)), which would
    // serialise the prefetched tiles behind each barrier; inside the loop only LDS traffic has to be
    // complete before the barrier (HIP guide, "Pipelining across barriers").
    // prologue: tile 0 staged, tile 1 in registers.  Two register sets: the rows of tile t+2 are
    // requested while tile t is on the matrix pipe and written to LDS one phase later.
    u32x4 s0[QN], s1[QN];
    // NORMS: the norms of a unit's rows (<= 1 024 floats) go from the store straight into LDS (buffer_load ... lds: no register
    // holds them) -- requested with the unit's first two tiles, i.e. for the next unit in front of the current unit's flush,
    // behind the tile loop's closing barrier (nobody reads the current unit's norms any more); the barrier in front of the
    // unit's first tile waits for them (vmcnt) like for everything else.  Rows past the head's end read as 0 (out of range).
    auto norms_load = [&]() {
        const int64_t left = n - row_base;
        const int rows = (int)(left < (int64_t)CROWS ? left : (int64_t)CROWS);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rnh + row_base), 0, rows * 4, 0x00020000);
        for (int c = wave; c * WAVE < CROWS; c += SK_WAVES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (__attribute__((address_space(3))) void*)(s_rnall + c * WAVE), 4, lane * 4,
                                                     c * WAVE * 4, 0, 0);
    };
    tile_load(0, s0);
    tile_load(1, s1);
    if constexpr (NORMS) norms_load();
    for (;;) {
    const uint16_t* const xcur = xh;                      // the unit being computed and flushed in this iteration
    const int64_t row_base_cur = row_base;
    int16_t* const codes_cur = codes + (int64_t)(unit / chunks) * L * n;
    int nt = (int)((n - row_base_cur + SH_ROWS - 1) / SH_ROWS);
    if (nt > chunk_tiles) nt = chunk_tiles;
    if (tid == 0) s_qn = 0;                               // (everybody is past the last unit's exact pass: the barrier in front of its cut)
    tile_store(0, s0);
    __syncthreads();                                      // ... and past its cut, which read the sign matrix
    MP_STAMP(stamp, 41);
#define MP_SK_PHASE(T, LOADSET, STORESET)                                  \
    if ((T) < nt) {                                                        \
        tile_load((T) + 2, LOADSET);                                       \
        compute(T);                                                        \
        tile_store(((T) + 1) & 1, STORESET);                               \
        lds_barrier();                                                     \
    }
    for (int t = 0; t < nt; t += 2) {
        MP_SK_PHASE(t, s0, s1)
        MP_SK_PHASE(t + 1, s1, s0)
        if (t == 2) MP_STAMP(stamp, 45);
    }
#undef MP_SK_PHASE
    __syncthreads();
    MP_STAMP(stamp, 42);
    // the next unit's first two row tiles travel while this one is flushed
    unit += unit_step;
    const bool more = unit < units;                       // uniform
    if (more) {
        xh = x + (int64_t)(unit / chunks) * head_stride;
        if constexpr (NORMS) rnh = rnorm + (int64_t)(unit / chunks) * rnorm_stride;
        row_base = (int64_t)(unit % chunks) * CROWS;
        tile_load(0, s0);
        tile_load(1, s1);
        if constexpr (NORMS) norms_load();
    }

    // exact pass over the queued candidates: one 16-lane group per candidate
    const int nq = min(s_qn, SK_QCAP);
    const bool overflow = s_qn > SK_QCAP;
    for (int e = tid >> 4; e < nq; e += nthr >> 4) {
        const uint32_t ent = s_queue[e];
        const int l16 = tid & 15;
        const int crow = ent >> 21;
        const int pbase = (int)((ent >> 17) & 15u) * 32 + 4 * (int)((ent >> 16) & 1u);
        for (uint32_t cm = ent & 0xffffu; cm; cm &= cm - 1u) {           // (uniform over the entry's 16 lanes)
        const int i = __builtin_ctz(cm);
        const int coff = pbase + (i & 3) + 8 * (i >> 2);
        double part = 0.0;
        for (int d8 = l16; d8 < CPR; d8 += 16) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(xcur + (row_base_cur + crow) * row_stride + d8 * 8);
            const u32x4 w = *reinterpret_cast<const u32x4*>(Wt + (int64_t)(col0 + coff) * D + d8 * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                part += (double)bf16_lo(a[j]) * (double)bf16_lo(w[j]) + (double)bf16_hi(a[j]) * (double)bf16_hi(w[j]);
        }
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        part += __shfl_xor(part, 8);
        if (l16 == 0) {
            uint32_t* wd = &s_rowbits[crow * BW + (coff >> 5)];
            const uint32_t m = 1u << (coff & 31);
            if (part > 0.0) atomicOr(wd, m); else atomicAnd(wd, ~m);
        }
        }
    }
    if (overflow) {   // more candidates than the queue holds (adversarial input): every thread re-checks
                      // its share of the bit matrix exactly -- slow, correct
        for (int p = tid; p < tables_per_wg * K * (nt * SH_ROWS); p += nthr) {
            const int coff = p / (nt * SH_ROWS), crow = p % (nt * SH_ROWS);
            if (col0 + coff >= KL || row_base_cur + crow >= n) continue;
            double ex = 0.0;
            for (int d = 0; d < D; ++d)
                ex += (double)bf16_bits_to_f32(xcur[(row_base_cur + crow) * row_stride + d]) *
                      (double)bf16_bits_to_f32(Wt[(int64_t)(col0 + coff) * D + d]);
            uint32_t* wd = &s_rowbits[crow * BW + (coff >> 5)];
            const uint32_t m = 1u << (coff & 31);
            if (ex > 0.0) atomicOr(wd, m); else atomicAnd(wd, ~m);
        }
    }
    __syncthreads();
    MP_STAMP(stamp, 43);

    // Cut the K-bit codes out of the sign matrix: a lane owns a ROW (its words sit 17 apart: no bank conflict), walks the
    // workgroup's tables with a 64-bit window over the row's bits -- position and word index are uniform, so the window's
    // bookkeeping is scalar -- and stores one code per table; for a table, the wave's 64 lanes write 128 contiguous bytes
    // of codes[l][.].  (Through round 5 a lane cut four CONSECUTIVE rows of one table to store 8 bytes: lanes 4 x 17 words
    // apart, a 4- to 8-way bank conflict on every read, 8.5 of a workgroup's 57 us.)
    const int ntok = (int)((n - row_base_cur < (int64_t)nt * SH_ROWS) ? (n - row_base_cur) : (int64_t)nt * SH_ROWS);
    const int tables_here = (L - table0 < tables_per_wg) ? L - table0 : tables_per_wg;
    for (int r0 = wave * WAVE; r0 < ntok; r0 += SK_WAVES * WAVE) {
        const int j = r0 + lane;
        const bool live = j < ntok;
        const uint32_t* rw = s_rowbits + (live ? j : 0) * BW;
        uint32_t lo = rw[0], hi = rw[1];
        int pos = 0, widx = 2;                                            // uniform
        // the store: a descriptor over the table's ntok codes of this unit (uniform) + the lane's 32-bit offset; a lane past
        // the unit's rows is out of range and the hardware drops its store: no branch
        int16_t* tbase = codes_cur + (int64_t)table0 * n + row_base_cur;
        for (int tb = 0; tb < tables_here; ++tb) {
            const uint32_t c = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)pos) & kmask;
            __builtin_amdgcn_raw_buffer_store_b16((short)c, __builtin_amdgcn_make_buffer_rsrc(tbase, 0, ntok * 2, 0x00020000),
                                                  j * 2, 0, 0);
            tbase += n;
            pos += K;
            if (pos >= 32) {
                pos -= 32;
                lo = hi;
                hi = rw[widx < BW ? widx : BW - 1];                       // (the last window may ask one word past the row)
                ++widx;
            }
        }
    }
    MP_STAMP(stamp, 44);
    if (!more) break;
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

static void simhash_keys_geometry(int K, int sets, int& tables_per_wg, int& tiles_per_wg);

// rows of Wt / entries of wnorm: every plane any workgroup of either kernel can touch
int simhash_padded_cols(int K, int L) {
    int tp, tiles;
    simhash_geometry(K, tp, tiles);
    const int wgs = (L + tp - 1) / tp;
    int cols = wgs * tiles * 32;
    for (int sets = 2; sets <= SK_MAX_SETS; sets += 2) {     // the key kernel's span per head_dim class (sk_sets): whole tiles
        simhash_keys_geometry(K, sets, tp, tiles);           // of the LAST workgroup, live or not, must exist in Wt
        const int kcols = ((L + tp - 1) / tp - 1) * tp * K + SK_WAVES * sets * 32;
        cols = cols > kcols ? cols : kcols;
    }
    return cols;
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

hipError_t launch_simhash_query(const uint16_t* q, const uint16_t* Wt, const float* wnorm, int R,
                                int D, int K, int L, int32_t* codes, float* qnorm, float* dbg,
                                hipStream_t st) {
    int tp, tiles;
    simhash_geometry(K, tp, tiles);
    dim3 grid((L + tp - 1) / tp, (unsigned)((R + SH_ROWS - 1) / SH_ROWS));
    dim3 block(64 * (tiles > 4 ? tiles : 4));
#define MP_SH_CASE(DD)                                                                          \
    if (D == DD) {                                                                              \
        hipLaunchKernelGGL((simhash_query_kernel<DD>), grid, block, 0, st, q, Wt, wnorm,        \
                           (int64_t)R, K, L, tp, tiles, codes, qnorm, dbg, g_stamp);            \
        return hipGetLastError();                                                               \
    }
    MP_SH_CASE(128)
    MP_SH_CASE(64)
    MP_SH_CASE(256)
#undef MP_SH_CASE
    return hipErrorInvalidValue;
}

// key-side geometry: a workgroup hashes floor(32 waves sets / K) tables (at most SK_MAX_TABLES), `sets` column tiles of
// 32 planes per wave; its plane span starts at table0 * K whatever the alignment.
static void simhash_keys_geometry(int K, int sets, int& tables_per_wg, int& tiles_per_wg) {
    int tp = (32 * SK_WAVES * sets) / K;
    if (tp > SK_MAX_TABLES) tp = SK_MAX_TABLES;
    tables_per_wg = tp;
    tiles_per_wg = (tp * K + 31) / 32;
}

// Tiles per unit (row chunk) and groups per XCD.  The workgroups are persistent: 8 x groups x wgs_x of them (two per CU at
// most) walk ceil(units / (8 groups)) units each; pick the chunk that minimises rounds x (per-tile time x tiles + the flush's
// fixed cost) -- only the ratio of the two constants matters.  The chunk's sign matrix [32 x tiles][17] words has to fit in
// LDS next to a second workgroup.
static void keys_chunk_tiles(int64_t n, int wgs_x, int heads, int sets, int& chunk_tiles, int& groups) {
    int cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        cus = prop.multiProcessorCount;
    int g = (2 * cus / 8) / wgs_x;                        // groups per XCD
    if (g < 1) g = 1;
    const int64_t tiles = (n + SH_ROWS - 1) / SH_ROWS;
    int best = 1;
    double best_cost = 1e300;
    for (int ch = 1; ch <= SK_CH_MAX; ++ch) {
        // the chunk's sign matrix next to a second workgroup's: 52 KB of the CU's 160 (the static stage, queue and norms
        // of a workgroup are ~24 KB at head_dim 128)
        if ((size_t)ch * SH_ROWS * (SK_WAVES * sets + 1) * sizeof(uint32_t) > 52u * 1024u) break;
        const int64_t units = ((tiles + ch - 1) / ch) * heads;
        const int64_t rounds = (units + 8 * (int64_t)g - 1) / (8 * (int64_t)g);
        const double cost = (double)rounds * (1.9 * ch + 4.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = ch; }
    }
    chunk_tiles = best;
    const int64_t units = ((tiles + best - 1) / best) * heads;
    const int64_t need = (units + 7) / 8;                 // groups per XCD that find a unit at all
    groups = (int)(need < g ? need : g);
}

// keys [heads][n][D] -> codes int16 [heads][L][n]; all kv heads in ONE launch so that the tail
// round of one head is filled by the next
hipError_t launch_simhash_keys_strided(const uint16_t* keys, int64_t head_stride, int64_t row_stride,
                                       const uint16_t* Wt, const float* wnorm, int heads, int64_t n, int D, int K,
                                       int L, int16_t* codes, const float* rnorm, int64_t rnorm_stride, hipStream_t st) {
    int tp, tiles;
    const int sets = sk_sets(D);
    simhash_keys_geometry(K, sets, tp, tiles);
    const int wgs_x = (L + tp - 1) / tp;
    if (heads < 1 || heads > 65535) return hipErrorInvalidValue;
    static thread_local int64_t memo_n = -1;
    static thread_local int memo_x = 0, memo_h = 0, memo_ch = 0, memo_g = 0;
    if (memo_n != n || memo_x != wgs_x || memo_h != heads) {
        keys_chunk_tiles(n, wgs_x, heads, sets, memo_ch, memo_g);
        memo_n = n; memo_x = wgs_x; memo_h = heads;
    }
    const int ch = memo_ch, groups = memo_g;
    const int64_t crows = (int64_t)ch * SH_ROWS;
    const int64_t chunks = (n + crows - 1) / crows;
    const int64_t blocks = (int64_t)groups * wgs_x * 8;
    if (chunks > INT32_MAX || blocks > INT32_MAX) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks);
    dim3 block(64 * SK_WAVES);
    const size_t lds = (size_t)crows * (SK_WAVES * sets + 1) * sizeof(uint32_t);   // the chunk's sign matrix
#define MP_SK_CASE(DD)                                                                              \
    if (D == DD) {                                                                                  \
        if (rnorm != nullptr)                                                                       \
            hipLaunchKernelGGL((simhash_keys_kernel<DD, true>), grid, block, lds, st, keys, head_stride, \
                               row_stride, Wt, wnorm, n, K, L, tp, tiles, ch, wgs_x, (int)chunks, heads, \
                               groups, rnorm, rnorm_stride, codes, g_stamp);                        \
        else                                                                                        \
        hipLaunchKernelGGL((simhash_keys_kernel<DD>), grid, block, lds, st, keys, head_stride,      \
                           row_stride, Wt, wnorm, n, K, L, tp, tiles, ch, wgs_x, (int)chunks, heads, \
                           groups, (const float*)nullptr, (int64_t)0, codes, g_stamp);                                                         \
        return hipGetLastError();                                                                   \
    }
    MP_SK_CASE(128)
    MP_SK_CASE(64)
    MP_SK_CASE(256)
#undef MP_SK_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_simhash_keys(const uint16_t* keys, const uint16_t* Wt, const float* wnorm,
                               int heads, int64_t n, int D, int K, int L, int16_t* codes, hipStream_t st) {
    return launch_simhash_keys_strided(keys, n * D, D, Wt, wnorm, heads, n, D, K, L, codes, nullptr, 0, st);
}

}  // namespace mp
