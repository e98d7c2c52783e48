// lsh.hip -- LSH table build and per-step retrieve on gfx950.
//
// Replaces library/lsh/lsh.cc: LSH::fill (:143-201), LSH::retrieve / batch_retrieve
// (:210-288), get_mask (:308-314).  Integer work, bit-exact as a set + nnz.
//
// HBM layout (per layer):
//   bounds int32 [B*Hkv][L][NB][R+1]  positions in the table row that cut bucket b of table l into R
//                                  token ranges: entry 0 = start, entry R = end of the bucket (the
//                                  reference's table_start / table_end, lsh.h:38-39, interleaved: a probe
//                                  is one 8-byte-wide random read), entry r = first position whose token id
//                                  is >= r * range_len.  R = 1, 2, 4, 8, 16 or 32 = the number of workgroups that
//                                  serve one query head in the decode kernel (fixed at alloc): member r of
//                                  a head's cluster probes entries (r, r+1) and streams, counts, emits and
//                                  gathers ONLY the tokens of its range -- nothing is replicated;
//   table  int32 [B*Hkv][L][M]    token ids in code-sorted order, ASCENDING inside a bucket when R > 1
//                                  (the device build emits them so; mp_lsh_fill checks and re-sorts),
//                                  row stride M (lsh.h:40).
//
// retrieve: the CPU code is serial per head with a byte mask in DRAM (lsh.cc:266-283).  Here
// one workgroup owns a head: the collision state of all M tokens lives in LDS as two bitmaps
// (A = seen once, B = seen twice or more; 2*M/8 bytes = 24.6 KB at M = 98304), bucket ids are
// streamed with coalesced 256-byte wave loads and applied with LDS atomics (ds_or_rtn_b32),
// and the result is emitted by a popcount sweep of B with a block-wide prefix sum, i.e. in
// (byte counters with non-returning ds_add were measured and lost: the 96 KB sweep costs more
// than the returning atomics; the phase is bound by streaming ~64 KB of ids through one CU)
// ASCENDING token order (the reference emits second-hit order; only set + nnz are defined,
// library/lsh/test.py:43-56).
#include <mutex>

#include <hip/hip_runtime.h>

namespace mp {
__device__ int d_stamp_stride = 0;    // debug: > 0 = every workgroup records its phase stamps (set_stamp_stride)
}
#define MP_STAMP_STRIDE ::mp::d_stamp_stride

#include <type_traits>
#include "common.h"

// Phase stamps of THIS translation unit are buffered in LDS and written out when thread 0 leaves the kernel
// (MP_STAMP_FLUSH): a stamp written straight to memory is a store the next `s_waitcnt vmcnt(0)` of its wave waits for,
// which charged every phase boundary ~0.1-0.5 us of the stamp's own round trip (scripts/phase_spread.py).
#if MP_STAMPS
namespace mp {
__shared__ unsigned long long s_stampbuf[64];
}
#undef MP_STAMP
#define MP_STAMP(stamp, slot)                                                                \
    do {                                                                                     \
        if ((stamp) != nullptr && threadIdx.x == 0) ::mp::s_stampbuf[(slot)] = wall_clock64(); \
    } while (0)
#define MP_STAMP_INIT(stamp)                                                                 \
    do {                                                                                     \
        if ((stamp) != nullptr && threadIdx.x == 0)                                          \
            for (int _i = 0; _i < 64; ++_i) ::mp::s_stampbuf[_i] = 0ull;                     \
    } while (0)
#define MP_STAMP_FLUSH(stamp)                                                                \
    do {                                                                                     \
        if ((stamp) != nullptr && threadIdx.x == 0) {                                        \
            const int _ss = MP_STAMP_STRIDE;                                                 \
            if (_ss > 0 || (blockIdx.x | blockIdx.y | blockIdx.z) == 0) {                    \
                unsigned long long* _d = (stamp) + (_ss > 0 ? (size_t)blockIdx.x * _ss : 0); \
                for (int _i = 0; _i < 64; ++_i)                                              \
                    if (::mp::s_stampbuf[_i] != 0ull) _d[_i] = ::mp::s_stampbuf[_i];         \
            }                                                                                \
        }                                                                                    \
    } while (0)
// behind the merge ONE wave runs, whichever drew the last ticket: its lane 0 stamps and flushes
#define MP_STAMP_L(stamp, slot)                                                              \
    do {                                                                                     \
        if ((stamp) != nullptr && (threadIdx.x & 63) == 0) ::mp::s_stampbuf[(slot)] = wall_clock64(); \
    } while (0)
#define MP_STAMP_FLUSH_L(stamp)                                                              \
    do {                                                                                     \
        if ((stamp) != nullptr && (threadIdx.x & 63) == 0) {                                 \
            const int _ss = MP_STAMP_STRIDE;                                                 \
            if (_ss > 0 || (blockIdx.x | blockIdx.y | blockIdx.z) == 0) {                    \
                unsigned long long* _d = (stamp) + (_ss > 0 ? (size_t)blockIdx.x * _ss : 0); \
                for (int _i = 0; _i < 64; ++_i)                                              \
                    if (::mp::s_stampbuf[_i] != 0ull) _d[_i] = ::mp::s_stampbuf[_i];         \
            }                                                                                \
        }                                                                                    \
    } while (0)
#else
#define MP_STAMP_INIT(stamp) do { } while (0)
#define MP_STAMP_FLUSH(stamp) do { } while (0)
#define MP_STAMP_L(stamp, slot) do { } while (0)
#define MP_STAMP_FLUSH_L(stamp) do { } while (0)
#endif

#include "attn_head.h"

namespace mp {

extern unsigned long long* g_stamp;   // simhash.hip

hipError_t set_stamp_stride(int stride) {
    return hipMemcpyToSymbol(HIP_SYMBOL(d_stamp_stride), &stride, sizeof(int));
}

#ifndef MP_RT_THREADS
#define MP_RT_THREADS 1024
#endif
constexpr int RT_THREADS = MP_RT_THREADS;  // 16 waves: one workgroup per query head (A/B builds: 512, two per CU)
constexpr int RT_WAVES = RT_THREADS / 64;
constexpr int RT_GROUP = 12;               // buckets in flight per wave per round (x 2 chunks of 64 ids)
// The barriers of the decode chain that order LDS data only are lds_barrier() (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads()
// also waits for vmcnt(0), i.e. for the acknowledgement of every by-product store in front of it (EXPERIMENTS.md R5-4).
// Table-side loads -- direct slots, bucket records, table ids: every line of them is read once per launch, by one CU -- are
// non-temporal (`nt`: no claim on the L2 the K / V rows and the hyperplanes live in).  Round 4, same instruction schedule with
// and without the bit on these 121 loads: cfg 3 29.77 -> 29.13 us per layer, cfg 1 and cfg 4 within +-0.1 (EXPERIMENTS.md R4-15).
#define MP_TLOAD(p) __builtin_nontemporal_load(p)
constexpr int RT_TAIL_CAP = 2048;          // pooled chunk descriptors for ids beyond 128 per bucket
constexpr int RT_TAIL_UNROLL = 8;
constexpr int CLUSTER_MAX = MP_CLUSTER_MAX; // workgroups per query head of the decode kernel, at most (common.h)

// ---------------------------------------------------------------- LSH::fill
// grid = Hkv*L rows of one request; one workgroup per (kv head, table) row.  RS = R + 1 entries per
// bucket: this kernel writes entries 0 (start) and R (end); lsh_subbounds_kernel fills the rest.
// err bits: 1 = invalid input (unsorted codes, code or id out of range); 4 = ids not ascending inside a
// bucket (only looked at when R > 1; the host then re-sorts the row's buckets on device); 32 = an id >= 2^17 (the
// layer's table words then have no room for a payload: capi.hip lsh_widen).
__global__ __launch_bounds__(256) void lsh_fill_kernel(
    const int16_t* __restrict__ codes,   // [Hkv*L][n] sorted ascending per row
    const int32_t* __restrict__ ids,     // [Hkv*L][n]
    int64_t n, int NB, int64_t M, int RS,
    int32_t* __restrict__ bounds,        // [Hkv*L][NB][RS]   (this request's slice)
    int32_t* __restrict__ table,         // [Hkv*L][M]
    int* __restrict__ err) {
    const int64_t row = blockIdx.x;
    const int16_t* c = codes + row * n;
    const int32_t* src = ids + row * n;
    int32_t* b = bounds + row * NB * RS;
    int32_t* dst = table + row * M;
    // buckets that do not occur keep start = end = 0 (lsh.cc:177-185 on zeroed arrays)
    for (int i = threadIdx.x; i < NB * RS; i += blockDim.x) b[i] = 0;
    __syncthreads();
    bool bad = false, unsorted = false, wide = false;
    for (int64_t k = threadIdx.x; k < n; k += blockDim.x) {
        const int v = c[k];
        const int prev = (k > 0) ? (int)c[k - 1] : -1;
        const int next = (k + 1 < n) ? (int)c[k + 1] : 0x7fffffff;
        const int32_t id = src[k];
        if (v < 0 || v >= NB || v < prev || id < 0 || id >= M) {
            bad = true;
        } else {
            if (v != prev) b[v * RS] = (int)k;                  // first position of value v
            else if (RS > 2 && id <= src[k - 1]) unsorted = true;
            if (v != next) b[v * RS + RS - 1] = (int)(k + 1);   // one past the last
        }
        dst[k] = id;
        if (id >= (1 << 17)) wide = true;
    }
    if (bad) atomicOr(err, 1);
    if (unsorted) atomicOr(err, 4);
    if (wide) atomicOr(err, 32);
}

// codes of a reference-sorted row back in token order (the fallback of mp_lsh_fill when a bucket's ids are
// not ascending -- torch.sort without stable=True, models/attnserver.py:187): tok[row][ids[k]] = codes[k];
// the row is then rebuilt by lsh_build_kernel, which emits ascending ids.  This path needs the ids of a row to be a
// permutation of [0, n) (what a sort returns): an id outside sets err bit 1, a token no id named keeps code -1 and is
// reported by the rebuild.
__global__ void lsh_unsort_kernel(const int16_t* __restrict__ codes, const int32_t* __restrict__ ids,
                                  int64_t n, int16_t* __restrict__ tok, int* __restrict__ err) {
    const int64_t row = blockIdx.y;
    bool bad = false;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = ids[row * n + k];
        if (id >= 0 && id < n) tok[row * n + id] = codes[row * n + k];
        else bad = true;
    }
    if (bad) atomicOr(err, 1);
}

// entries 1 .. R-1 of every bucket: first position of the bucket whose token id is >= r * range_len
// (ids ascend inside a bucket).  One workgroup per (kv head, table) row, one thread per (bucket, cut): a binary
// search over a table row that was just written (L2 hits).
__global__ __launch_bounds__(256) void lsh_subbounds_kernel(
    const int32_t* __restrict__ table, int32_t* __restrict__ bounds, int NB, int R, int range_len, int64_t M,
    uint32_t idmask) {                                   // the words may carry a payload above the id (packed build)
    const int64_t row = blockIdx.x;
    const int RS = R + 1;
    const int32_t* t = table + row * M;
    int32_t* b = bounds + row * NB * RS;
    for (int i = threadIdx.x; i < NB * (R - 1); i += blockDim.x) {
        const int bk = i / (R - 1), r = 1 + i % (R - 1);
        int lo = b[bk * RS], hi = b[bk * RS + R];            // first position in [start, end) with t[p] >= target
        const int target = r * range_len;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int)((uint32_t)t[mid] & idmask) < target) lo = mid + 1;
            else hi = mid;
        }
        b[bk * RS + r] = lo;
    }
}

// direct piece slots [rows][NB][R][SW] (R > 1; SW = 32, 16 or 8 words by the mean piece length, lsh_slot_log2): word 0 =
// length of the piece (bucket, range), word 1 = its position in the table row, words 2 .. SW - 1 = its first SW - 2 ids.  The decode kernel reads a piece with ONE 128-byte access
// straight from the query's code, without the sub-bounds round trip in front of it; the position lets it fetch the
// rest of a longer piece with the next access.  Half a wave writes a slot.
// (Measured and rejected, round 3: the layout [group][R][L][NB][32], in which the slots a cluster member reads are one
// contiguous eighth of its group's 157 MB -- an eighth of the pages to translate: 19.4 / 22.1 / 20.9 us per launch at cfg 1
// randn / clustered / cfg 4 against 19.4 / 22.1 / 21.0 with this one.  Address translation is not what the phase waits for.)
__global__ __launch_bounds__(256) void lsh_slots_kernel(const int32_t* __restrict__ table,
                                                        const int32_t* __restrict__ bounds,
                                                        int32_t* __restrict__ slots, int NB, int R, int64_t M, int swl) {
    const int64_t row = blockIdx.y;                      // (kv head, table) row of this request
    const int SW = 1 << swl;                             // words per slot: 32, 16 or 8 (lsh_slot_log2)
    const int32_t* t = table + row * M;
    const int32_t* b = bounds + row * NB * (R + 1);
    int32_t* s = slots + row * NB * R * SW;
    const int sl = threadIdx.x & (SW - 1);
    const int gpb = 256 >> swl;                          // groups of SW lanes per block: one group writes a slot
    constexpr int U = 8;                                 // pieces in flight per group (round 5: 4 -> 8, stores non-temporal)
    const int total = NB * R;
    // R is a power of two (decode_cluster_size): piece -> (bucket, range) by shift and mask, the bucket's record of R + 1
    // entries starts at piece + bucket; offsets inside a row fit 32 bits (NB R SW <= 2^31: mp_lsh_alloc).  Round 5: the kernel
    // issued 230 M vector instructions per launch at cfg 1 -- 0.37 ms of issue slots for a 0.6-ms launch -- most of them the
    // integer division by R and 64-bit address arithmetic per piece.
    const int clog = 31 - __builtin_clz((unsigned)R);
    for (int p0 = (blockIdx.x * gpb + (threadIdx.x >> swl)) * U; p0 < total; p0 += gridDim.x * gpb * U) {
        int lo[U], hi[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int piece = p0 + u < total ? p0 + u : total - 1;
            const int at = piece + (piece >> clog);
            lo[u] = b[at];
            hi[u] = b[at + 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int len = hi[u] - lo[u];
            v[u] = sl == 0 ? len : lo[u];
            if (sl >= 2) v[u] = sl - 2 < len ? t[lo[u] + sl - 2] : 0;
        }
        int32_t* dst = s + ((uint32_t)p0 << swl) + sl;
#pragma unroll
        for (int u = 0; u < U; ++u)      // 1.26 GB per layer at cfg 1, written once, read by the decode launches much later
            if (p0 + u < total) __builtin_nontemporal_store(v[u], dst + ((uint32_t)u << swl));
    }
}

// ---------------------------------------------------------------- key norms as a payload of the table entries
// table word = token id | (bits 14..0 of the token's bf16 key norm << idbits) (idbits = 17, where max_length <= 2^17).
// The decode kernel's gather then finds a selected token's norm in LDS (scattered there when the token was hit the
// second time) instead of spending one random HBM access per token on it.
// one request's rows [Hkv * L][M]; kn [Hkv][M] (the request's slice of the attention store's norms); bad[kv head] is set
// when a norm the tables reference is not a non-negative bf16 number (the decode kernel then ignores the payload).
// kn == nullptr strips the payloads again (a layer whose ids outgrow 17 bits).
__global__ void lsh_attach_norms_kernel(int32_t* __restrict__ table, const float* __restrict__ kn, int L, int64_t M,
                                        int idbits, int* __restrict__ bad) {
    const int64_t row = blockIdx.y;
    const float* knr = kn ? kn + (row / L) * M : nullptr;
    int32_t* t = table + row * M;
    const uint32_t mask = (1u << idbits) - 1u;
    bool b = false;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < M; p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t w = t[p];
        if (w == -1) continue;                                 // not an entry
        const uint32_t id = (uint32_t)w & mask;
        uint32_t u = (knr != nullptr && id < (uint64_t)M) ? __float_as_uint(knr[id]) : 0u;
        if ((u & 0x8000ffffu) != 0u || (u & 0x7f800000u) == 0x7f800000u) {   // not bf16, negative, inf / nan
            b = true;
            u = 0u;
        }
        t[p] = (int32_t)(id | ((u >> 16) << idbits));
    }
    if (b) atomicOr(bad + row / L, 1);
}
hipError_t launch_lsh_attach_norms(int32_t* table, const float* kn, int Hkv, int L, int64_t M, int idbits, int* bad,
                                   hipStream_t st) {
    int gx = (int)((M + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(lsh_attach_norms_kernel, dim3(gx, Hkv * L), dim3(256), 0, st, table, kn, L, M, idbits, bad);
    return hipGetLastError();
}

// ---------------------------------------------------------------- device-side table build
// (replacement of the half-written LSH::fastfill, lsh.cc:93-142, and of the torch.sort +
// LSH::fill of models/attnserver.py:186-193): stable counting sort of one (kv head, table) row
// of UNSORTED codes -> bounds + ascending ids per bucket.  One workgroup per row.
//
// What it is bound by is the scatter of the ids: written straight to HBM, 4 bytes at a time into
// NB open bucket frontiers per row, it costs 1.6 ms of a 2.1 ms launch at cfg 1 (more open lines
// than L2 holds, so every store becomes its own 32-byte sector write).  So the row is sorted in
// LDS tile by tile (T tokens): every wave owns a contiguous slice of the tile and a private set of
// NB counters, one block-wide scan turns them into per-wave cursors inside the tile's sorted order,
// the waves scatter (id, bucket) into an LDS stage without any barrier or cross-wave ordering --
// inside a wave, 64 consecutive tokens are ranked by a match-any ballot (lane order == token
// order), which keeps the ids of a bucket ascending -- and the stage is written out by position,
// so that adjacent lanes write adjacent ids of the same bucket run.
constexpr int BUILD_LDS_COUNTERS = 32768;   // direct variant: waves = min(16, 32768 / NB)

__device__ __forceinline__ unsigned long long match_any_bits(int v, bool ok, int nbits) {
    unsigned long long m = __ballot(ok);
    for (int bit = 0; bit < nbits; ++bit) {
        const bool one = (v >> bit) & 1;
        const unsigned long long bm = __ballot(one);
        m &= one ? bm : ~bm;
    }
    return m;
}

// row histogram with wave-private counters (cnt [nw][NB], zeroed here), then the exclusive scan
// of the bucket totals: writes bounds, returns each bucket's start through `start_of(i, ex, v)`.
template <typename F>
__device__ __forceinline__ void build_row_histogram(const int16_t* __restrict__ c, int n, int NB,
                                                    int* s_cnt, int* s_tmp, int32_t* __restrict__ b, int RS,
                                                    int* __restrict__ err, F&& start_of, int copies = 0) {
    // copies: sets of counters [copies][NB] the waves are spread over (0 = one per wave)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = copies > 0 ? copies : (int)(blockDim.x >> 6);
    for (int i = tid; i < nw * NB; i += blockDim.x) s_cnt[i] = 0;
    __syncthreads();
    int* mine = s_cnt + (wave % nw) * NB;
    bool bad = false;
    auto count = [&](int v) {
        if (v < 0 || v >= NB) bad = true;
        else atomicAdd(&mine[v], 1);
    };
    // Round 6: the pass over the row's codes was a loop of 2-byte loads, four in flight per lane -- 48 dependent HBM round
    // trips per row: 92 of the 233 us a workgroup spent on a row of cfg 1 (scripts/build_phases.py), 39 % of the kernel.  Now a
    // lane takes EIGHT consecutive codes per 16-byte load (which code a lane counts does not matter to a histogram) and keeps
    // four such loads in flight: 32 codes per lane and round trip.  The row's start is only 2-byte aligned (n is arbitrary):
    // the few codes in front of the first 16-byte boundary and behind the last whole vector are counted one by one.
    const int head_all = (int)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(c) & 15u)) & 15u) >> 1);
    const int head = head_all < n ? head_all : n;
    const int nvec = (n - head) >> 3;
    const u32x4* cv = reinterpret_cast<const u32x4*>(c + head);
    const int step = (int)blockDim.x;
    for (int i0 = tid; i0 < nvec; i0 += 4 * step) {
        u32x4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                       // unconditional (clamped): a load under a branch is waited for at the join
            const int i = i0 + u * step;
            q[u] = cv[i < nvec ? i : nvec - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * step < nvec) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    count((int)(int16_t)(q[u][w] & 0xffffu));
                    count((int)(int16_t)(q[u][w] >> 16));
                }
            }
        }
    }
    for (int k = tid; k < head; k += step) count((int)c[k]);
    for (int k = head + (nvec << 3) + tid; k < n; k += step) count((int)c[k]);
    (void)lane;
    if (bad) atomicOr(err, 1);
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < NB; base += blockDim.x) {
        const int i = base + tid;
        int v = 0;
        if (i < NB) {
#pragma unroll
            for (int w = 0; w < 16; ++w) v += (w < nw) ? s_cnt[w * NB + i] : 0;   // independent LDS reads
        }
        int total;
        __syncthreads();
        const int ex = block_excl_scan(v, s_tmp, total) + carry;
        if (i < NB) {
            b[i * RS] = (v > 0) ? ex : 0;
            b[i * RS + RS - 1] = (v > 0) ? ex + v : 0;
            start_of(i, ex, v);
        }
        carry += total;
    }
    __syncthreads();
}

// LDS: cnt [nw][NB] | gbase [NB] | gdelta [NB] | s_tmp [32] | stage_id [T] | stage_b [T] (u16)
// PACK (round 5, NB <= 2048: the BASELINE configurations, K = 10 and 11): the tile's per-wave counters are 16 bits wide,
// two to a word -- a tile holds at most T = 8 192 tokens, so a count or a cursor never carries into its neighbour -- and
// the row histogram (whole-row counts: 32 bits) uses the same words as nw / 2 sets of counters, two waves to a set: 72 KB
// per workgroup of eight waves (NB = 1024: T = 8 192; NB = 2048: T = 4 096) instead of 123 - 128 KB, so TWO workgroups
// share a CU.  The tile loop is a chain of short phases
// between barriers (count, scan, rank, write out); with one workgroup per CU nothing filled the waits, and a launch of
// 1 200 rows ran in five rounds of 256 (cfg 4: 300 rows in two rounds, the second 17 % full).
// FAST (round 5): the rank of a token inside its bucket's run comes from ONE returning LDS atomic on the wave's cursor --
// the hardware serialises the lanes of an instruction that hit the same word, and on gfx950 it was observed to do so in lane
// order, which is token order.  That order is not architecturally promised, so it is VERIFIED: the write-out checks that
// neighbouring entries of a bucket's run ascend and raises err bit 64 otherwise; the host then rebuilds the request with the
// exact ranking (match-any ballots over the code's bits: ~75 vector instructions per token -- the kernel was bound by VALU
// issue: 319 M wave-instructions per launch at cfg 1 = 0.52 ms of issue slots alone, profiles/archive/r05_pmc_sq_insts_cfg1.md).
// phase stamps of the table build (stamp build only: workgroup 0, thread 0, slots 50 .. 58 of the stamp buffer -- kernel start,
// row histogram done, and for the row's THIRD tile: top, validated + zeroed, counted, scanned, ranked, written out; kernel end)
#if MP_STAMPS
__device__ unsigned long long* d_build_stamp = nullptr;
#define MP_BSTAMP(slot)                                                                                   \
    do {                                                                                                  \
        if (d_build_stamp != nullptr && blockIdx.x == 0 && threadIdx.x == 0) d_build_stamp[slot] = wall_clock64(); \
    } while (0)
#else
#define MP_BSTAMP(slot) do { } while (0)
#endif
template <int TPL, bool PACK = false, bool FAST = false>   // tokens per lane and tile: T = TPL * blockDim.x
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void lsh_build_kernel(   // 16 waves per CU either way: 128 VGPRs
    const int16_t* __restrict__ codes,   // [Hkv*L][n] unsorted
    int n, int NB, int nbits, int64_t M, int RS, int32_t* __restrict__ bounds, int32_t* __restrict__ table,
    int* __restrict__ err,
    // packed build (round 4): with the attention store's norms of this request (kn [Hkv][M], nullptr = plain ids) a word
    // is written as  id | bf16 bits 14..0 of the token's norm << idbits  in the SAME pass -- what lsh_attach_norms_kernel
    // does to a finished table (a second sweep over 472 MB at cfg 1, plus a second build of the direct slots) when the
    // first decode finds plain ids.  bad[kv head] is set for a norm that is not a non-negative bf16 number.
    const float* __restrict__ kn, int L, int idbits, int* __restrict__ bad,
    // round 5: the sub-bounds (entries 1 .. R-1 of a bucket's record: first position whose token id >= r * range_len) are
    // written HERE -- tiles never straddle a range boundary, so when the tile that starts at token r * range_len begins,
    // a bucket's running write position IS its sub-bound r.  lsh_subbounds_kernel did it afterwards by 7 (R - 1) binary
    // searches per bucket over a 472-MB table that had left the L2: 196 dependent HBM reads per thread, 0.19 ms per layer
    // at cfg 1 and 0.30 at cfg 4.  range_len = 0: leave the entries to that kernel.
    int range_len) {
    extern __shared__ int s_mem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int T = TPL * blockDim.x;
    int* s_cnt = s_mem;                                       // PACK: nw * NB / 2 words of two 16-bit counters
    int* s_gbase = s_cnt + (PACK ? (nw * NB) / 2 : nw * NB);
    int* s_gdelta = s_gbase + NB;
    int* s_tmp = s_gdelta + NB;
    int* s_id = s_tmp + 32;
    uint16_t* s_b = reinterpret_cast<uint16_t*>(s_id + T);
    uint16_t* s_cnt16 = reinterpret_cast<uint16_t*>(s_cnt);
    const int64_t row = blockIdx.x;
    const int16_t* c = codes + row * n;
    int32_t* dst = table + row * M;
    MP_BSTAMP(50);
    // (PACK: the whole-row histogram counts up to n per bucket -- 32-bit counters, nw / 2 sets in the nw * NB / 2 words)
    build_row_histogram(c, n, NB, s_cnt, s_tmp, bounds + row * NB * RS, RS, err,
                        [&](int i, int ex, int v) { s_gbase[i] = v > 0 ? ex : -1; },   // -1: a bucket without tokens (it
                        PACK ? nw / 2 : 0);                                             // never moves: every tile adds 0)
    MP_BSTAMP(51);
    int* mine = s_cnt + wave * NB;                            // (not PACK)
    const int wbase = wave * NB;                              // PACK: this wave's counters are cnt16[wbase + v]
    auto count_add = [&](int v, int k) -> int {               // += k on this wave's counter of bucket v; returns the old value
        if (PACK) {
            const int sh = (v & 1) * 16;
            const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(s_cnt) + ((wbase + v) >> 1), (unsigned)k << sh);
            return (int)((old >> sh) & 0xffffu);
        }
        return atomicAdd(&mine[v], k);
    };
    // Round 5: every barrier of the tile loop orders LDS only (lds_barrier): the stores of tile t -- never read back here --
    // drain while tile t + 1 is counted and ranked, and the codes of tile t + 1 are requested before tile t is written out
    // (with __syncthreads() each tile waited for its own 32 KB of scattered stores at the next tile's first barrier).
    auto load_codes = [&](int t0, int (&vq)[TPL]) {
        const int w0 = t0 + wave * (WAVE * TPL);
#pragma unroll
        for (int j = 0; j < TPL; ++j) {
            const int kk = w0 + j * WAVE + lane;
            vq[j] = (int)c[kk < n ? kk : n - 1];              // unconditional (clamped): validated where it is used
        }
    };
    constexpr bool PREFETCH = TPL < 32;                       // (32 codes per lane: no registers for a second set)
    int vq[TPL], vnext[PREFETCH ? TPL : 1];
    if (PREFETCH) load_codes(0, vq);
    const int R = RS - 1;
    int32_t* brow = bounds + row * NB * RS;
    // sub-bound r of every bucket <- its running write position (0 for a bucket without tokens, as the search gives it)
    auto dump_subbound = [&](int r) {
        for (int i = tid; i < NB; i += blockDim.x) {
            const int g = s_gbase[i];
            brow[i * RS + r] = g < 0 ? 0 : g;
        }
    };
    const bool cuts = TPL < 32 && range_len > 0 && R > 1;     // uniform (the 32-codes-per-lane form, K = 12 / 13, has no
                                                              // register to spare: its sub-bounds stay with the search kernel)
    int t0 = 0;
    int tile_no = 0;                                          // (stamp build)
    while (t0 < n) {
        const bool st_tile = tile_no++ == 2;
        if (st_tile) MP_BSTAMP(52);
        // the tile: [t0, tend), at most T tokens.  A range boundary inside it is taken in the tile's stride where it falls
        // between two waves' slices (cfg 1: range_len = 12 288 = 12 slices of 1 024): the cursor the scan hands the first
        // wave behind the boundary IS the bucket's sub-bound -- one store per bucket, no extra tile.  Elsewhere (cfg 4:
        // range_len = 8 224 against slices of 512) the tile ends at the boundary.
        int tend = t0 + T;
        tend = tend < n ? tend : n;
        int cut_wave = -1, cut_r = 0;                           // uniform
        if (cuts) {
            if (t0 > 0 && t0 % range_len == 0 && t0 / range_len <= R - 1) dump_subbound(t0 / range_len);
            const int nb = (t0 / range_len + 1) * range_len;   // the next boundary behind t0
            if (nb < tend) {
                const int off = nb - t0;
                if (off % (WAVE * TPL) == 0 && nb + range_len >= tend && nb / range_len <= R - 1) {
                    cut_wave = off / (WAVE * TPL);
                    cut_r = nb / range_len;
                } else {
                    tend = nb;
                }
            }
        }
        if (!PREFETCH) load_codes(t0, vq);
        // this wave's slice of the tile: tokens w0 + j*64 + lane, j < TPL, kept in registers
        const int w0 = t0 + wave * (WAVE * TPL);
        for (int i = tid; i < (PACK ? (nw * NB) / 2 : nw * NB); i += blockDim.x) s_cnt[i] = 0;
#pragma unroll
        for (int j = 0; j < TPL; ++j) {
            const int kk = w0 + j * WAVE + lane;
            const int v = vq[j];
            vq[j] = (kk < tend && v >= 0 && v < NB) ? v : -1;
        }
        lds_barrier();
        if (st_tile) MP_BSTAMP(53);
#pragma unroll
        for (int j = 0; j < TPL; ++j)
            if (vq[j] >= 0) (void)count_add(vq[j], 1);
        if constexpr (PREFETCH) {
            if (tend < n) load_codes(tend, vnext);             // uniform; consumed by the next iteration
        }
        lds_barrier();
        if (st_tile) MP_BSTAMP(54);
        // per-wave cursors inside the tile's sorted order; where the tile's run of a bucket goes
        int carry = 0;
        for (int base = 0; base < NB; base += blockDim.x) {
            const int i = base + tid;
            int v = 0;
            constexpr int NWX = PACK ? 8 : 16;            // waves per workgroup, at most (PACK is launched with eight)
            int cw[NWX];                                  // the waves' counts of bucket i: ONE batch of LDS reads
#pragma unroll
            for (int w = 0; w < NWX; ++w) {
                cw[w] = (i < NB && w < nw) ? (PACK ? (int)s_cnt16[w * NB + i] : s_cnt[w * NB + i]) : 0;
                v += cw[w];
            }
            int total;
            lds_barrier();
            const int ex = block_excl_scan<true>(v, s_tmp, total) + carry;
            if (i < NB) {
                int run = ex;
                const int g = s_gbase[i];
#pragma unroll
                for (int w = 0; w < NWX; ++w) {
                    if (w < nw) {
                        if (PACK) s_cnt16[w * NB + i] = (uint16_t)run;      // (a 16-bit store: the neighbour's half is another thread's)
                        else s_cnt[w * NB + i] = run;
                        // the boundary in front of wave w's slice: its cursor, as a position of the row, is the sub-bound
                        if (w == cut_wave) brow[i * RS + cut_r] = g < 0 ? 0 : run + (g - ex);
                    }
                    run += cw[w];
                }
                s_gdelta[i] = g - ex;
                s_gbase[i] = g + v;
            }
            carry += total;
        }
        lds_barrier();
        if (st_tile) MP_BSTAMP(55);
        const int tile_count = carry;                   // valid tokens of the tile
        if constexpr (FAST) {
#pragma unroll
            for (int j = 0; j < TPL; ++j) {
                const int v = vq[j];
                if (v >= 0) {
                    const int pos = count_add(v, 1);          // lanes of one instruction on one word: served in lane order (verified below)
                    s_id[pos] = w0 + j * WAVE + lane;
                    s_b[pos] = (uint16_t)v;
                }
            }
        } else
#pragma unroll
        for (int j = 0; j < TPL; ++j) {
            const int v = vq[j];
            const bool ok = v >= 0;
            if (!__ballot(ok)) continue;
            const unsigned long long peers = match_any_bits(v, ok, nbits);
            if (ok) {
                const int rank = __popcll(peers & ((1ull << lane) - 1ull));
                const int leader = __ffsll((long long)peers) - 1;
                int start = 0;
                if (lane == leader) start = count_add(v, __popcll(peers));
                start = __shfl(start, leader);
                s_id[start + rank] = w0 + j * WAVE + lane;
                s_b[start + rank] = (uint16_t)v;
            }
        }
        lds_barrier();
        if (st_tile) MP_BSTAMP(56);
        // Written out by position, TPL entries per thread, in straight-line batches of 8 (round 5): all stage reads, then all
        // norm gathers, then all stores.  As a rolled loop (the trip count is tile_count / blockDim) every entry was a
        // dependent chain of its own -- LDS read -> L2 gather of the norm -> store -- and a tile paid TPL of them in turn.
        if constexpr (TPL >= 32) {        // (K = 12, 13: 32 codes per lane live in registers -- the rolled form, no batch arrays)
            const float* knr = kn ? kn + (row / L) * M : nullptr;
            bool refused = false;
            for (int p = tid; p < tile_count; p += blockDim.x) {
                const int id = s_id[p];
                uint32_t u = knr ? __float_as_uint(knr[id]) : 0u;
                if ((u & 0x8000ffffu) != 0u || (u & 0x7f800000u) == 0x7f800000u) {   // not bf16, negative, inf / nan
                    refused = true;
                    u = 0u;
                }
                dst[p + s_gdelta[s_b[p]]] = knr ? (int32_t)((uint32_t)id | ((u >> 16) << idbits)) : id;
            }
            if (refused) atomicOr(bad + row / L, 1);
        } else {
            {
            constexpr int OB = TPL >= 16 ? 4 : 8;                // entries per batch (TPL is 8 or 16; 16 keeps two sets of codes in registers)
            const float* knr = kn ? kn + (row / L) * M : nullptr;    // a tile's 8 192 norms: 32 KB, read once per table row
            bool refused = false, misordered = false;
#pragma unroll
            for (int j0 = 0; j0 < TPL; j0 += OB) {
                if (tid + j0 * (int)blockDim.x >= tile_count) break;     // (not uniform: the usual per-lane exit)
                int idv[OB], pos[OB];
#pragma unroll
                for (int j = 0; j < OB; ++j) {
                    const int p = tid + (j0 + j) * (int)blockDim.x;
                    const int pc = p < tile_count ? p : tid;            // (tid < tile_count here: a valid entry)
                    idv[j] = s_id[pc];
                    const int bk = s_b[pc];
                    pos[j] = pc + s_gdelta[bk];
                    if (FAST) {    // the run of a bucket must ascend: this entry against its right neighbour
                        const int pn = pc + 1 < tile_count ? pc + 1 : pc;
                        if (pn != pc && (int)s_b[pn] == bk && s_id[pn] <= idv[j]) misordered = true;
                    }
                }
                if (kn == nullptr) {
#pragma unroll
                    for (int j = 0; j < OB; ++j)
                        if (tid + (j0 + j) * (int)blockDim.x < tile_count) dst[pos[j]] = idv[j];
                } else {
                    uint32_t nu[OB];
#pragma unroll
                    for (int j = 0; j < OB; ++j) nu[j] = __float_as_uint(knr[idv[j]]);
#pragma unroll
                    for (int j = 0; j < OB; ++j) {
                        if (tid + (j0 + j) * (int)blockDim.x < tile_count) {
                            uint32_t u = nu[j];
                            if ((u & 0x8000ffffu) != 0u || (u & 0x7f800000u) == 0x7f800000u) {   // not bf16, negative, inf / nan
                                refused = true;
                                u = 0u;
                            }
                            dst[pos[j]] = (int32_t)((uint32_t)idv[j] | ((u >> 16) << idbits));
                        }
                    }
                }
            }
            if (refused) atomicOr(bad + row / L, 1);
            if (FAST && misordered) atomicOr(err, 64);
            }
        }
        // the next tile's zeroing of s_cnt is ordered after this loop's LDS reads by its barrier (lgkmcnt(0) in front of
        // it: the reads have landed in registers); s_id / s_b / s_gdelta are rewritten only after two more barriers
        if constexpr (PREFETCH) {
#pragma unroll
            for (int j = 0; j < TPL; ++j) vq[j] = vnext[j];
        }
        if (st_tile) MP_BSTAMP(57);
        t0 = tend;
    }
    if (cuts) {   // boundaries at or behind the last token: the bucket's end (its final write position)
        lds_barrier();                                         // (the last tile's s_gbase)
        for (int r = (n + range_len - 1) / range_len; r <= R - 1; ++r)
            if (r >= 1) dump_subbound(r);
    }
    MP_BSTAMP(58);
}

// Direct variant for NB too large for the staged layout (K >= 14): the row is cut into one
// contiguous segment per wave and the waves scatter straight to HBM from private cursors.
__global__ __launch_bounds__(1024) void lsh_build_direct_kernel(
    const int16_t* __restrict__ codes, int n, int NB, int nbits, int64_t M, int RS,
    int32_t* __restrict__ bounds, int32_t* __restrict__ table, int* __restrict__ err) {
    extern __shared__ int s_mem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    int* s_cnt = s_mem;                  // [nw][NB] per-wave histogram, then per-wave cursors
    int* s_tmp = s_mem + nw * NB;
    const int64_t row = blockIdx.x;
    const int16_t* c = codes + row * n;
    int32_t* b = bounds + row * NB * RS;
    int32_t* dst = table + row * M;
    for (int i = tid; i < nw * NB; i += blockDim.x) s_cnt[i] = 0;
    __syncthreads();
    const int seg = (((n + nw - 1) / nw) + 63) & ~63;     // tokens per wave
    const int k0 = wave * seg, k1 = (k0 + seg < n) ? k0 + seg : n;
    int* mine = s_cnt + wave * NB;
    bool bad = false;
#pragma unroll 4
    for (int k = k0 + lane; k < k1; k += WAVE) {
        const int v = c[k];
        if (v < 0 || v >= NB) bad = true;
        else atomicAdd(&mine[v], 1);
    }
    if (bad) atomicOr(err, 1);
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < NB; base += blockDim.x) {
        const int i = base + tid;
        int v = 0;
        if (i < NB)
            for (int w = 0; w < nw; ++w) v += s_cnt[w * NB + i];
        int total;
        __syncthreads();
        const int ex = block_excl_scan(v, s_tmp, total) + carry;
        if (i < NB) {
            b[i * RS] = (v > 0) ? ex : 0;
            b[i * RS + RS - 1] = (v > 0) ? ex + v : 0;
            int run = ex;
            for (int w = 0; w < nw; ++w) {
                const int t = s_cnt[w * NB + i];
                s_cnt[w * NB + i] = run;
                run += t;
            }
        }
        carry += total;
    }
    __syncthreads();
    for (int k = k0; k < k1; k += WAVE) {
        const int kk = k + lane;
        const int v = (kk < k1) ? (int)c[kk] : -1;
        const bool ok = v >= 0 && v < NB;
        const unsigned long long peers = match_any_bits(v, ok, nbits);
        if (ok) {
            const int rank = __popcll(peers & ((1ull << lane) - 1ull));
            const int leader = __ffsll((long long)peers) - 1;
            int start = 0;
            if (lane == leader) start = atomicAdd(&mine[v], __popcll(peers));
            start = __shfl(start, leader);
            dst[start + rank] = kk;
        }
    }
}

// ---------------------------------------------------------------- LSH::batch_retrieve
// block = 1024, dynamic LDS:
//   A[words] | B[words] | s_start[Lpad] | s_len[Lpad] | s_tail[RT_TAIL_CAP] | s_tmp[32] | s_ntail
// `words` = words per collision bitmap = ceil(tokens the workgroup owns / 32): all M tokens of the head in
// the stand-alone retrieve (grid = B*H), one RANGE of M / R tokens in the decode kernel (grid = R * B*H).
// HASH = 1 fuses the query SimHash of models/attnserver.py:264-270 into the prologue: the head's
// query row is normalised (bf16 semantics as simhash.hip), every thread evaluates <= 2 hyperplanes
// from the chunk-major plane copy Wk[D/8][KLpad][8] (coalesced 16-byte loads, 128 f32 FMAs each,
// exact-sign guard in f64 by the whole wave), the sign bits meet in LDS by ballot and are packed into codes --
// the codes never travel through HBM between two kernels (they are still written for get_mask).
struct HashArgs {
    const uint16_t* q;       // [BH][D] bf16 queries
    const uint16_t* Wk;      // [D/8][KLpad][8] bf16 hyperplanes, chunk-major
    const float* wnorm;      // [KLpad] column norms (upper bounds)
    int32_t* codes_out;      // [BH][L]
    float* qnorm_out;        // [BH]
    int D, K, KLpad;
    int exact_norm;          // 1: the query row is normalised by the exact (f64) sequence always (A/B, tests)
};

static int g_exact_norm = 0;   // mp_debug_set_option("simhash_exact_norm"): read at launch
void set_exact_norm(int v) { g_exact_norm = v; }

// AD > 0 (with HASH) appends the sparse attention of the head to the same launch (attn_head.h): the
// selected ids never leave the workgroup's LDS.  A head is served by a CLUSTER of R = 1, 2, 4, 8, 16 or 32
// workgroups (grid = R * BH, block b -> head b % BH, rank b / BH).  The head's TOKENS are partitioned over
// the members: member r owns tokens [r * range_len, (r + 1) * range_len).  Every bucket's ids ascend, so
// the part of a probed bucket that falls into the range is the contiguous piece between the bucket's
// sub-bounds r and r + 1: the member reads that pair, streams only that piece, counts collisions in a bitmap
// of range_len bits, emits and gathers its own selected tokens.  Only the query hash is repeated by the
// members; with R > 1 their softmax states meet through one partial each and an arrival ticket.
struct AttnArgs {
    const uint16_t* kv;      // [B*Hkv][M][2][D]
    const float* kn;         // [B*Hkv][M]
    float* part_o;           // [BH][maxs][D]
    float2* part_ml;         // [BH][maxs]
    int* part_cnt;           // [BH][CLUSTER_MAX] selected tokens of every member (R > 1), bits 0..23; bits 24..27 its XCC_ID
    int* head_cnt;           // [BH] arrival tickets, zero between launches
    uint16_t* out;           // [BH][D] bf16
    float* mve;              // [2][BH]
    float2* head_mz;         // [BH]
    const int32_t* slots;    // [B*Hkv][L][NB][R][2^slot_log2] direct piece slots (R > 1, short pieces) or nullptr
    int slot_log2;           // words per slot, log2 (5, 4 or 3: lsh_slot_log2)
    float* score;            // [BH][M] (nullable); member r's logits start at column r * range_len
    int* err;                // device flag: bit 4 = a cluster member ran on another XCD than observed
    int BH, BHp, maxs, cap, cluster_log2;   // BHp: heads per rank in the grid (BH padded to 8 when R > 1); cap: ids of the LDS stage (multiple of AH_SLICE)
    int same_xcd;            // the members of a cluster share one XCD (and its L2): verified by the host
    unsigned long long* xw;  // split hash: [BH][xwords] (launch sequence << 32 | 32 sign bits) exchanged by a cluster
    unsigned int* xseq;      // split hash: [BH] sequence number of the current launch (the merger advances it)
    int xwords, xmode;       // words per head; xmode 2 = nobody publishes (test: every member takes the fallback)
    const int* pay_bad;      // [B*Hkv] or nullptr: the table entries of this layer carry their tokens' key norms,
                             // unless the flag of the head's KV group says one of them could not be packed ...
    const int* idbits_dev;         // the layer's id width as the device knows it (a fill that widens the layer clears it
                                   // in stream order): overrides the launch argument, which a replayed graph froze
    const unsigned int* att_ver;   // [B*Hkv] ... or the version of the norms the group's rows carry (0: plain ids)
    const unsigned int* kn_ver;    // [B*Hkv] is not the version of the norms the attention store holds now: both are
                                   // device words written in stream order by the fills / the packing, so a replayed
                                   // graph sees the state of its replay, not of its capture
    // optional static window (models/attnserver.py:281-308): exact attention over the first win_len[h]
    // rows of a second KV store joins the same softmax, which IS flashinfer.merge_state of the two parts
    const uint16_t* win_kv;  // [B*Hkv][win_M][2][D] or nullptr
    const int32_t* win_len;  // [BH]
    int64_t win_M;
    // stand-alone retrieve, host-buffer mode (capi.hip: HostRetrieve): a second copy of the emitted rows in HBM and a
    // checksum per row (two u32 sums: row_mix of (entry, position), entry x position), by which the attention entry
    // recognises rows the kernel wrote straight into the caller's pinned memory
    int32_t* rows2;          // [BH][M] or nullptr  (HASH = 5: the hyperplanes' plane-major copy Wt [KLpad][D] bf16 instead)
    uint32_t* rowsum;        // [BH][2] or nullptr
    // (with rows2 the stand-alone retrieve also leaves a second copy of its counts in HBM -- nnz is pinned host memory then --
    // through part_cnt, which only the decode uses otherwise: a field of its own grew the decode kernel's argument block and
    // cost cfg 1 0.1 us per launch)
};

// ---- the lean decode's gather (EXPERIMENTS.md R6-1): a wave that is done counting CLAIMS slices of the workgroup's list of selected
// table words -- SHORT_L entries, one step of the fold, by fetch_add on the list's head -- waits until its slice is reserved in
// full or nobody counts any more, reads it (the read is the written-yet check: an entry still says -1) and folds its rows into the
// wave's running softmax; again, until the list is used up.  Returns the entries this wave folded.
template <int ADL, int SHORT_L>
__device__ __forceinline__ int lean_claim_and_gather(AhState& st_own, int* s_done, int* s_res, int* s_head, const int32_t* s_ids, int lcap,
                                                     const uint16_t* kv_l, const float* kn_l, const u32x4& qv_l, const float* qn_lds, int64_t M,
                                                     int K, int L, uint32_t idmask, int idbits, bool pay,
                                                     unsigned long long* __restrict__ stamp) {
    const int lane = threadIdx.x & 63;
    // entries per claim = one step of the fold (claims of 12 -- what a wave adds on average -- were +0.2 us: a step's cost is
    // per step, R6-1)
    const int CL = SHORT_L;
    int folded = 0;
    for (;;) {
        int start = 0, nh = 0;
        // (the first look at the two counters travels with the claim: one LDS round trip, not two)
        int done = __hip_atomic_load(s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int res = __hip_atomic_load(s_res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0) start = __hip_atomic_fetch_add(s_head, CL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        start = __builtin_amdgcn_readfirstlane(start);
        if (start >= lcap) break;                                    // (beyond the stage: the spill list, below)
        // until the slice is reserved in full, or nobody counts any more (then `res` is final: a wave's last
        // reservation precedes its "done", and `done` is read first)
        for (;;) {
            if (res >= start + CL || done >= RT_WAVES) break;
            // a poll is an instruction the counting waves of this SIMD do not issue: far from its turn a wave sleeps longer
            if (start + CL - res > 2 * CL) __builtin_amdgcn_s_sleep(8);
            else __builtin_amdgcn_s_sleep(1);
            done = __hip_atomic_load(s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            res = __hip_atomic_load(s_res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        res = __builtin_amdgcn_readfirstlane(res);
        nh = (res < lcap ? res : lcap) - start;
        if (nh <= 0) break;
        nh = nh < CL ? nh : CL;
        // reserved is not written: the reserving wave stores its entries right behind its atomic.  The gather's own
        // read of the slice is the check -- row group r of the step reads entries r UPS .. r UPS + UPS - 1 (attn_head_fold_lean);
        // an entry of the slice that still says "not written" (-1) means: read again.  The fold gets the registers.
        constexpr int LPRc = ADL / 8, RPLc = 64 / LPRc;
        u32x4 pre0 = {0u, 0u, 0u, 0u}, pre1 = {0u, 0u, 0u, 0u};      // (two named registers: an array here went to scratch)
        auto read_slice = [&](auto ups_tag) {
            constexpr int UPSc = decltype(ups_tag)::value;
            const int e0 = (lane / LPRc) * UPSc;
            for (;;) {
                bool missing = false;
                pre0 = *reinterpret_cast<const volatile u32x4*>(s_ids + start + e0);
#pragma unroll
                for (int e = 0; e < 4; ++e) missing = missing || (e0 + e < nh && pre0[e] == 0xffffffffu);
                if constexpr (UPSc > 4) {
                    pre1 = *reinterpret_cast<const volatile u32x4*>(s_ids + start + e0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) missing = missing || (e0 + 4 + e < nh && pre1[e] == 0xffffffffu);
                }
                if (__ballot(missing) == 0ull) break;            // wave-uniform
            }
        };
        constexpr int UPS_S = SHORT_L / RPLc;                        // entries a row group reads: 4 (or 8: two registers)
        const int e0s = (lane / LPRc) * UPS_S;
        auto slice = [&](int j) {                                    // (the fold asks for entries e0 and, UPS = 8, e0 + 4)
            if constexpr (UPS_S > 4) return (j - e0s) >= 4 ? pre1 : pre0;
            else return pre0;
        };
        read_slice(std::integral_constant<int, UPS_S>{});
        attn_head_fold_lean<ADL, SHORT_L>(st_own, kv_l, kn_l, qv_l, *qn_lds, nh, M, K, L, 0, 1, slice, idmask, idbits, pay,
                                          stamp);
        folded += nh;
    }
    return folded;
}

// ---- emission of a workgroup's selected tokens in ASCENDING order (everything but the lean decode): every thread counts the set
// bits of its contiguous words of bitmap B, a block-wide exclusive scan gives it its place, and it writes its tokens' ids to the
// head's row in HBM (`out`: the stand-alone retrieve's result, the decode's by-product), to the gather's stage in LDS (AD > 0:
// the first `cap` of them) and -- host-buffer mode of the stand-alone retrieve -- to a second row in HBM with two checksums.
// Returns the number of tokens selected.
template <int AD>
__device__ __forceinline__ int emit_ordered(const uint32_t* bmB, int words, int T0, int32_t* __restrict__ out, int32_t* s_ids, int cap,
                                            int32_t* __restrict__ out2, int* s_tmp, uint32_t& cs1, uint32_t& cs2,
                                            unsigned long long* __restrict__ stamp) {
    const int tid = threadIdx.x;
    const int nsw = words;
    const int wpt = (nsw + RT_THREADS - 1) / RT_THREADS;
    const int w0 = tid * wpt;
    int cnt = 0, total = 0;
    for (int k = 0; k < wpt; ++k)
        if (w0 + k < nsw) cnt += __popc(bmB[w0 + k]);
    int off = block_excl_scan<true>(cnt, s_tmp, total);
    MP_STAMP(stamp, 20);
    for (int k = 0; k < wpt; ++k) {
        if (w0 + k >= nsw) break;
        uint32_t bits = bmB[w0 + k];
        const int base = T0 + ((w0 + k) << 5);
        while (bits) {
            const int p = __ffs((int)bits) - 1;
            bits &= bits - 1;
            out[off] = base + p;
            if (AD > 0 && off < cap) s_ids[off] = base + p;
            if (AD == 0 && out2 != nullptr) {
                out2[off] = base + p;
                cs1 += row_mix((uint32_t)(base + p + 1), (uint32_t)(off + 1));
                cs2 += (uint32_t)(base + p + 1) * (uint32_t)(off + 1);
            }
            ++off;
        }
    }
    return total;
}

// ---- last phase of a head that is served by a cluster: ONE wave of every member -- the one that merged its workgroup's waves --
// arrives here with the member's softmax state (m, Z, this lane's elements o0 | o1 of the partial output) and its count.
// (Round 6, VERDICT r05 item 8: split out of lsh_head_body -- one instruction of difference in the kernels' ISA.  The query-hash
// phase was split out the same way and put back: as a function of its own the compiler issued the kernel-argument loads in front
// of the query row's request again -- +0.1 us per launch on every configuration; scripts/experiments/r06_hash_phase_as_function.patch,
// profiles/r06_ab_refactor.txt.)
template <int ADD>
__device__ __forceinline__ void cluster_handoff(const AttnArgs& aa, int64_t h, int rank, int clog, float m, float Z, float o0, float o1,
                                                int total, uint16_t* out_h, int32_t* __restrict__ nnz,
                                                unsigned long long* __restrict__ stamp) {
    const int lane = threadIdx.x & 63;
    // ---- cluster > 1: publish this member's state (a member without tokens publishes m = -inf, Z = 0), drain,
    // take an arrival ticket; the member that draws the last ticket merges (hand-off recipe:
    // cdna_hip_programming.md G16, as attn_sparse_kernel).  Block b runs on XCD b % 8, so when B*H is a multiple
    // of 8 the members of a cluster (blocks h, h + BH, ...) share ONE XCD and its L2: the hand-off then only has
    // to bypass the per-CU L1 (sc0, the workgroup scope of a split workgroup) instead of writing through to the
    // memory side (sc1), which takes three ~0.6 us L2 round trips instead of three ~1.8 us ones.  The
    // host enables it only after xcd_round_robin_verified() has seen the placement on this device, and every
    // launch re-checks that the members of a cluster really ran on ONE XCD: each publishes its XCC_ID next to its
    // count and the merger compares (err bit 4 otherwise: mp_attn_check reports it).  That comparison needs a merger:
    // members spread over several XCDs draw their tickets from different L2s, nobody draws the last one, nothing is
    // merged -- the counters left standing are what mp_attn_check looks for (attn_ticket_check_kernel, err bit 8) and
    // resets.  WHICH XCD a residue lands on
    // is not fixed -- under graph replay the round robin was observed to start elsewhere than in the probe
    // launches -- only that blocks b and b + 8k share one matters.
    constexpr int VPL = ADD / 64;
    const int nmem = 1 << clog;
    const int64_t pre = h * aa.maxs;
    int ticket = 0;
    uint32_t my_xcc = 0;
    if (aa.same_xcd) {
        // every member publishes the XCD it ran on next to its count; the merger compares them with its own
        my_xcc = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;   // HW_REG_XCC_ID[3:0]
        constexpr int kSc0 = 1;   // aux bit 0 = sc0 on gfx940+
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            aa.part_o + pre * ADD, 0, (int)(aa.maxs * ADD * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<float*>(aa.part_ml + pre), 0, (int)(aa.maxs * 8), 0x00020000);
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(aa.part_cnt + h * CLUSTER_MAX, 0,
                                                                            CLUSTER_MAX * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0), ro, (rank * ADD + lane * VPL) * 4, 0, kSc0);
        if (VPL == 2)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o1), ro, (rank * ADD + lane * 2 + 1) * 4, 0, kSc0);
        if (lane == 0) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m), rm, rank * 8, 0, kSc0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Z), rm, rank * 8 + 4, 0, kSc0);
            __builtin_amdgcn_raw_buffer_store_b32((uint32_t)total | (my_xcc << 24), rc, rank * 4, 0, kSc0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing lane of this wave drained
        MP_STAMP_L(stamp, 41);
        if (lane == 0)
            ticket = __hip_atomic_fetch_add(aa.head_cnt + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        MP_STAMP_L(stamp, 38);
        if (ticket != nmem - 1) {
            MP_STAMP_FLUSH_L(stamp);
            return;
        }
        if (lane == 0) {
            __hip_atomic_store(aa.head_cnt + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // split hash: every member is past the exchange -- the next launch gets a new sequence number
            if (aa.xseq != nullptr) __hip_atomic_fetch_add(aa.xseq + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        __hip_atomic_store(reinterpret_cast<unsigned int*>(aa.part_o + (pre + rank) * ADD + lane * VPL),
                           __float_as_uint(o0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (VPL == 2)
            __hip_atomic_store(reinterpret_cast<unsigned int*>(aa.part_o + (pre + rank) * ADD + lane * 2 + 1),
                               __float_as_uint(o1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(aa.part_ml + pre + rank),
                               (unsigned long long)__float_as_uint(m) | ((unsigned long long)__float_as_uint(Z) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(aa.part_cnt + h * CLUSTER_MAX + rank, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            ticket = __hip_atomic_fetch_add(aa.head_cnt + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        MP_STAMP_L(stamp, 38);
        if (ticket != nmem - 1) {
            MP_STAMP_FLUSH_L(stamp);
            return;
        }
        if (lane == 0) __hip_atomic_store(aa.head_cnt + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the last arriver merges the R records in rank order (bit-identical whoever merges).  Lane u < R reads member
    // u's (m, Z, count) -- one round of loads for any R up to 32 -- and every lane reads its two elements of all R
    // partial outputs; everything is requested before anything is used (a use inside the loading loop made every
    // member's loads wait for the previous member's: R dependent L2 round trips).  The scales exp(m_u - max) are computed
    // once, by lane u, and broadcast with v_readlane.  (Measured and rejected, round 4: the wave's two halves taking half
    // of the members each with 16-byte loads -- R / 2 load instructions instead of 2 R -- and one cross-half add: cfg 1
    // 19.42 against 19.27 us per layer, cfg 4 at R = 16 17.23 against 16.98; EXPERIMENTS.md R4-4.)
    float mm = -INFINITY, ZZ = 0.f, q0 = 0.f, q1 = 0.f;
    int csum = 0;
    auto merge_records = [&](auto n_tag) {
        constexpr int NM = decltype(n_tag)::value;                    // 8, 16 or 32 >= nmem: loads past nmem re-read member 0
        const int lu = lane < nmem ? lane : 0;
        float m_u, z_u;
        int c_u;
        float oa[NM], ob[NM];
        if (aa.same_xcd) {
            constexpr int kSc0 = 1;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                aa.part_o + pre * ADD, 0, (int)(aa.maxs * ADD * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<float*>(aa.part_ml + pre), 0, (int)(aa.maxs * 8), 0x00020000);
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(aa.part_cnt + h * CLUSTER_MAX, 0,
                                                                                CLUSTER_MAX * 4, 0x00020000);
            m_u = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rm, lu * 8, 0, kSc0));
            z_u = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rm, lu * 8 + 4, 0, kSc0));
            c_u = (int)__builtin_amdgcn_raw_buffer_load_b32(rc, lu * 4, 0, kSc0);
#pragma unroll
            for (int u = 0; u < NM; ++u) {
                const int uu = u < nmem ? u : 0;
                oa[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, (uu * ADD + lane * VPL) * 4, 0, kSc0));
                ob[u] = VPL == 2 ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, (uu * ADD + lane * 2 + 1) * 4, 0, kSc0))
                                 : 0.f;
            }
        } else {
            const unsigned long long pk = __hip_atomic_load(
                reinterpret_cast<unsigned long long*>(aa.part_ml + pre + lu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m_u = __uint_as_float((uint32_t)pk);
            z_u = __uint_as_float((uint32_t)(pk >> 32));
            c_u = __hip_atomic_load(aa.part_cnt + h * CLUSTER_MAX + lu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < NM; ++u) {
                const int uu = u < nmem ? u : 0;
                oa[u] = __uint_as_float(__hip_atomic_load(
                    reinterpret_cast<unsigned int*>(aa.part_o + (pre + uu) * ADD + lane * VPL), __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT));
                ob[u] = VPL == 2 ? __uint_as_float(__hip_atomic_load(
                                       reinterpret_cast<unsigned int*>(aa.part_o + (pre + uu) * ADD + lane * 2 + 1),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                 : 0.f;
            }
        }
        if (lane >= nmem) {
            m_u = -INFINITY;
            z_u = 0.f;
            c_u = 0;
        }
        // a member on another XCD (a placement the host did not observe): its partial may be stale in this L2
        if (aa.same_xcd) {
            const bool off = lane < nmem && ((uint32_t)c_u >> 24) != my_xcc;
            if (__ballot(off) != 0ull && lane == 0) atomicOr(aa.err, 4);
        }
        c_u &= 0xffffff;
        mm = wave_max(m_u);
        const float e_u = (m_u != -INFINITY) ? __expf(m_u - mm) : 0.f;   // a member without tokens: weight 0
        const float ez_u = e_u * z_u;
#pragma unroll
        for (int u = 0; u < NM; ++u) {
            const float e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e_u), u));     // 0 past nmem
            ZZ += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ez_u), u));
            csum += __builtin_amdgcn_readlane(c_u, u);
            q0 = fmaf(e, oa[u], q0);
            q1 = fmaf(e, ob[u], q1);
        }
    };
    if (nmem <= 8) merge_records(std::integral_constant<int, 8>{});
    else if (nmem <= 16) merge_records(std::integral_constant<int, 16>{});
    else merge_records(std::integral_constant<int, 32>{});
    attn_head_finalize<ADD>(mm, ZZ, q0, q1, out_h, aa.mve, aa.BH, (int)h, aa.head_mz);   // ZZ = 0: no member had a token
    if (lane == 0) nnz[h] = csum;
    MP_STAMP_L(stamp, 39);
    MP_STAMP_FLUSH_L(stamp);
}

// HASH: 0 = the codes are given (`query`), 1 = fused SimHash prologue, 3 = the same with the planes split over the
// members of the head's cluster (decode only, clusters on one XCD), 2 = (decode only) codes + ||q|| were written
// to ha.codes_out / ha.qnorm_out by simhash_query_kernel (the MFMA kernel) in a launch of its own: the A/B
// variant of the decode entry behind the decode_mfma_hash option.
// LEAN (decode only; mp_decode_*_ex with MP_DECODE_NO_BYPRODUCTS): the launch leaves no by-products -- no codes, no ||q||,
// no result rows, no logits -- and nothing between "counted" and the first row request: no workgroup barrier, no popcount
// sweep, no block scan, no ordered emission, no staging.  A token joins the workgroup's list in LDS the moment its SECOND
// collision is counted (exactly one lane sees its counter go 1 -> 2, whatever the interleaving: a reservation by one LDS
// atomic per wave and batch); a wave that is done counting CLAIMS the next 16 (32) entries of that list with another atomic
// and gathers them at once -- while other waves still count -- and comes back for more until production is over.  Balanced
// by construction (every step is a full slice, whoever takes it) and as early as it can be: what bounds the gather is the
// row requests a CU turns over (EXPERIMENTS.md R3-10; measured per wave here: a step's time grows with its valid rows), so
// they should start early and be spread evenly.  (Round 6 first built wave-OWNED lists -- no reservation, a Poisson share per
// wave: the workgroup then waits for its longest list, R6-1.)  Entries beyond the 4 096-entry stage (nearly every token
// selected: K = 1 tests) and the chunk pool's finds go through a spill list in HBM, folded by the wave that draws the last
// ticket.
template <int HASH, int CH, int AD, bool WIN, bool LEAN = false>    // CH = min(16, D / 8): plane chunks kept in registers (HASH only);
                                                 // WIN: fold the static window in (its own instantiation, so the
                                                 // plain decode kernel carries none of its code)
__device__ __forceinline__ void lsh_head_body(
    const int32_t* __restrict__ bounds, const int32_t* __restrict__ table,
    const int32_t* __restrict__ query, int32_t* __restrict__ results, int32_t* __restrict__ nnz,
    int G, int L, int NB, int64_t M, int R, int range_len, int words, int Lpad, int idbits, const HashArgs& ha,
    const AttnArgs& aa, unsigned long long* __restrict__ stamp) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_u32[];
    // the counting area: two bitmaps of `words` words (A = seen once, B = seen twice or more) -- or, LEAN, one COUNTER per
    // token of the range (8 bits while L <= 255, else 16; a word holds 4 / 2 of them, incremented by ONE returning 32-bit
    // atomic add whose result says "this was the token's second hit": no carry can cross a counter, a token is hit at most
    // L times): 8 / 16 x `words` words
    const int csh = (L <= 255) ? 3 : 4;                        // LEAN: log2 of the bits per counter
    const int cw = LEAN ? (words << csh) : 2 * words;          // (32 tokens per bitmap word: 8 / 16 counter words)
    uint32_t* bmA = s_u32;
    uint32_t* bmB = s_u32 + words;
    int* s_start = reinterpret_cast<int*>(s_u32 + cw);
    int* s_len = s_start + Lpad;
    uint32_t* s_tail = reinterpret_cast<uint32_t*>(s_len + Lpad);   // (table << 16 | chunk) of ids beyond 128
    int* s_tmp = reinterpret_cast<int*>(s_tail + RT_TAIL_CAP);
    int* s_ntail = s_tmp + 32;
    // HASH only: s_q (normalised query, bf16 pairs) | s_rn | s_bits (sign bits of the K*L planes)
    // (placed by INDEX, rounded up to 16 bytes for ds_read_b128: a uintptr_t round trip would lose the
    // LDS address space and turn every read of the query into a flat_load that waits on vmcnt too)
    uint32_t* s_q = s_u32 + ((cw + 2 * Lpad + RT_TAIL_CAP + 32 + 4 + 3) & ~3);
    float* s_rn = reinterpret_cast<float*>(s_q + 128);       // [0] guard bound of ||nq||, [1] ||q||
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_rn + 4);
    // AD only: s_ids [cap] (ids of this workgroup's slices) | s_merge | s_tk
    int32_t* s_ids = reinterpret_cast<int32_t*>(s_bits + ((((16 * L + 31) >> 5) + 2 * RT_WAVES + 4 + 3) & ~3));
    float* s_merge = reinterpret_cast<float*>(s_ids + (AD > 0 ? aa.cap : 0));
    int* s_tk = reinterpret_cast<int*>(s_merge + attn_head_lds_floats(RT_WAVES, AD > 0 ? AD : 2));
    uint32_t* s_qraw = reinterpret_cast<uint32_t*>(s_tk + 4);   // AD: the raw query row (bf16 pairs), 16-byte aligned
    uint16_t* s_kn = reinterpret_cast<uint16_t*>(s_qraw + 64);  // AD, table entries with a key-norm payload: bf16 [range_len]
    // Table entries may carry their token's key norm in the bits above `idbits` (lsh_attach_norms_kernel; max_length <=
    // 2^17: a 17-bit id + the 15 bits of a non-negative bf16).  The id is always masked; with `pay` the decode scatters
    // the norm of a token into LDS when it is hit the second time -- the gather then reads it from there instead of
    // spending one HBM line request in five on a 4-byte value (EXPERIMENTS.md R3-10).
    // (read BEHIND the query row's request, below: in front of it the four device words were three dependent scalar round
    // trips at the head of the kernel -- EXPERIMENTS.md R3-14)
    uint32_t idmask = idbits ? ((1u << idbits) - 1u) : 0xffffffffu;
    bool pay = false;                                           // uniform

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // decode: the grid is R x BHp blocks, BHp = B*H rounded up to a multiple of 8 when R > 1, so that the members of
    // a cluster (blocks h, h + BHp, ...) share the residue b % 8 -- one XCD -- whatever B*H is; the padding blocks leave
    const int BHp = (AD > 0) ? aa.BHp : 1;
    const int64_t h = (AD > 0) ? (int64_t)(blockIdx.x % BHp) : (int64_t)blockIdx.x;
    const int rank = (AD > 0) ? (int)(blockIdx.x / BHp) : 0;
    // (a padding block -- h >= B*H, only where B*H is not a multiple of 8 -- leaves BELOW, behind the query row's request:
    // in front of it the test was a kernel-argument round trip of its own ahead of the row's address)
    const bool padding_block = AD > 0 && h >= aa.BH;
    // -- the query row FIRST (its round trip heads the dependent chain: q -> norm -> LDS -> dots): wave 0, D/64
    //    elements per lane (D = 64, 128 or 256).  Requested before anything else touches the kernel arguments: they
    //    arrive in half a dozen dependent scalar loads, and behind them the row was requested ~1 us into the kernel.
    uint32_t e01 = 0u, e23 = 0u;                                 // elements 0,1 | 2,3 of this lane (bf16 pairs)
    // HASH = 5 (one workgroup per head, B*H a multiple of 32, head_dim 128): the heads h0 + 8 i, i < 4 -- four workgroups of
    // one XCD -- hash TOGETHER: member qm evaluates the 32-plane tiles t = qm (mod 4) against all FOUR query rows on the
    // matrix pipe and hands every row's sign bits to its head through the XCD's L2 (below).  Waves 0 .. 3 fetch a row each.
    constexpr bool QUAD = HASH == 5;
    const int64_t qh0 = (h & ~(int64_t)31) + (h & 7);
    const int qm = (int)((h & 31) >> 3);
    if ((HASH == 1 || HASH == 3 || HASH == 5) && wave < (QUAD ? 4 : 1)) {   // ONE load per lane, no loop: nothing to wait for here
        const int per0 = ha.D >> 6;
        const uint16_t* src = ha.q + (padding_block ? 0 : (QUAD ? qh0 + 8 * wave : h)) * ha.D + lane * per0;
        if (per0 == 2) e01 = *reinterpret_cast<const uint32_t*>(src);
        else if (per0 == 1) e01 = *src;
        else {
            const uint2 t = *reinterpret_cast<const uint2*>(src);
            e01 = t.x;
            e23 = t.y;
        }
    }
    MP_STAMP_INIT(stamp);
    const int clog = (AD > 0) ? aa.cluster_log2 : 0;
    const bool lead = rank == 0;                              // the member that writes codes / ||q||
    const int64_t g = h / G;
    const int RS = R + 1;
    const int32_t* bnd = bounds + g * L * NB * RS;
    const int32_t* tab = table + g * L * M;
    // the tokens this workgroup owns: [t0, t0 + tlen); sub-bound entries (e_lo, e_hi) of every probed bucket
    const int e_lo = (AD > 0) ? rank : 0, e_hi = (AD > 0) ? rank + 1 : R;
    const int64_t t0 = (AD > 0) ? (int64_t)rank * range_len : 0;
    const int64_t trem = M - t0;
    const uint32_t tlen = (AD > 0) ? (uint32_t)(trem <= 0 ? 0 : (trem < range_len ? trem : range_len)) : (uint32_t)M;

    MP_STAMP(stamp, 16);
    if (padding_block) return;
    if (tid == 0) {
        *s_ntail = 0;
        if (AD > 0) s_tk[0] = 0;                                  // (the waves' LDS ticket)
        s_tmp[30] = 0;                                            // pieces that overflow their direct slot
        s_tmp[29] = 1;                                            // split hash: every word of the head arrived
        s_tmp[27] = 0;                                            // LEAN: entries reserved in the workgroup's list (= tokens selected)
        s_tmp[25] = 0;                                            // LEAN: entries claimed
        s_tmp[24] = 0;                                            // LEAN: waves done counting
    }
    // collision bitmaps and piece lengths start at zero: done here, under the query row's round trip
    if constexpr (LEAN) {                                     // (cw is a multiple of 8: 16-byte stores)
        for (int i = tid * 4; i < cw; i += RT_THREADS * 4) *reinterpret_cast<u32x4*>(s_u32 + i) = u32x4{0u, 0u, 0u, 0u};
        // the list: every entry "not written yet" (-1 is no table word: the packing refuses a NaN norm, an id is < M)
        for (int i = tid * 4; i < (AD > 0 ? aa.cap : 0); i += RT_THREADS * 4)
            *reinterpret_cast<u32x4*>(s_ids + i) = u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    } else {
        for (int i = tid; i < 2 * words; i += RT_THREADS) s_u32[i] = 0u;
    }
    for (int l = tid; l < Lpad; l += RT_THREADS) s_len[l] = 0;
    // -- the layer's id width and the state of this KV group's payloads, as the device knows them: four words, requested
    //    HERE -- behind the row's request, in front of the hyperplanes -- as VECTOR loads (buffer loads of a uniform
    //    address) and looked at only where table words are first counted, behind the hash.  As scalar loads they came
    //    back through lgkmcnt, which the compiler waits on as a whole: it issued two of them, waited, issued the
    //    third, waited (round 3: three dependent round trips at the head of the kernel; round 4's first form: two, in
    //    front of the plane loads and of wave 0's look at the query row -- "q row in" 0.4 us late under the stamps).
    //    Vector loads return in order behind the row and are counted exactly.
    uint32_t st_ib = 0u, st_bad = 1u, st_av = 0u, st_kv = 1u;
    const bool st_live = AD > 0 && aa.idbits_dev != nullptr;       // uniform
    if (st_live) {
        st_ib = __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(aa.idbits_dev), 0, 4, 0x00020000), 0, 0, 0);
        if (aa.pay_bad != nullptr) {
            st_bad = __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(aa.pay_bad + g), 0, 4, 0x00020000), 0, 0, 0);
            st_av = __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(aa.att_ver + g), 0, 4, 0x00020000), 0, 0, 0);
            st_kv = __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(aa.kn_ver + g), 0, 4, 0x00020000), 0, 0, 0);
        }
    }
    bool bits_exchanged = false;                                 // uniform: the split hash delivered every sign word
    if (HASH == 2) {   // the hash ran as its own launch: only the raw query row (for q.k) and ||q|| are fetched here
        const int per = ha.D >> 6;
        if (wave == 0) {
            const uint16_t* src = ha.q + h * ha.D + lane * per;
            uint16_t* raw = reinterpret_cast<uint16_t*>(s_qraw) + lane * per;
            for (int i = 0; i < per; ++i) raw[i] = src[i];
            if (lane == 0) s_rn[1] = ha.qnorm_out[h];
        }
    }
    if (HASH == 1 || HASH == 3 || HASH == 5) {
        const int D = ha.D, KL = ha.K * L;
        const int chunks = D >> 3;                              // 16-byte plane chunks per hyperplane
        const u32x4* Wk4 = reinterpret_cast<const u32x4*>(ha.Wk);
        const int per = D >> 6;                                  // (the query row was requested at the top of the kernel)
        // -- the first pass's hyperplane chunks do not depend on q: put them in flight next.  All
        //    plane loads of this prologue are UNCONDITIONAL on clamped indices (columns >= KL of Wk are
        //    zero): with loads under divergent branches the compiler cannot count what is outstanding
        //    and waits vmcnt(0) before every use, which serialised the second pass's prefetch.
        // -- SPLIT hash (decode, clusters that share an XCD): every member of the head's cluster evaluates 1/R of the
        //    planes -- units of 64, one wave each: unit u belongs to member u % R, wave u / R -- and the sign bits are
        //    exchanged through the XCD's L2.  Every CU pulls its planes through a ~50 bytes/clock path
        //    (scripts/probes/plane_pull.hip): 384 KB at cfg 1, 845 KB at cfg 4 when a member hashes alone.
        const int U = (KL + 63) >> 6;                            // units of 64 planes
        const bool split = HASH == 3 && AD > 0 && aa.xw != nullptr && clog > 0 && U <= (RT_WAVES << clog);   // uniform
        const int unit = rank + (wave << clog);
        uint32_t seq = 0;
        if (split) {   // this launch's sequence number: written by the previous launch's merger (kernel boundary)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(aa.xseq + h, 0, 4, 0x00020000);
            seq = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 1);
        }
        u32x4 w[CH];
        auto load_planes = [&](int col) {
            const int cc = col < ha.KLpad ? col : ha.KLpad - 1;
#pragma unroll
            for (int i = 0; i < CH; ++i) w[i] = Wk4[cc + (int64_t)i * ha.KLpad];
        };
        if (!QUAD) load_planes(split ? (unit << 6) + lane : tid);   // (the quad's members pull a quarter of the planes, as MFMA tiles)
        // QUAD: this launch's number = the quad's arrival counter / 4, drawn HERE by an L2 atomic (launches of a stream do not
        // overlap: the four members of launch k draw 4 k .. 4 k + 3 before any member of launch k + 1 exists); a word tagged
        // with it proves by itself that it belongs to this launch.  Wave 15 has no tile: it draws.
        if (QUAD && wave == RT_WAVES - 1 && lane == 0)
            s_tmp[28] = (int)(__hip_atomic_fetch_add(aa.xseq + qh0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 2);
        uint32_t* s_q4 = s_tail;                                 // QUAD: the four normalised rows [4][64 words] (the pool's LDS, unused
        float* s_rn4 = reinterpret_cast<float*>(s_tail + 256);   // until the probe) and their guard bounds
        // the column norm of the first column this thread decides: fetched here, under the query row's round trip (as a
        // load behind the dot chain its L2 latency sat between the dots and the sign)
        const int col_first = split ? (unit << 6) + lane : tid;
        const float wn_first = ha.wnorm[col_first < ha.KLpad ? col_first : ha.KLpad - 1];
        // -- normalise the query row (QUAD: waves 0 .. 3 one row of the quad each; wave qm's is this head's own)
        const bool own_row = !QUAD || wave == qm;
        if (wave < (QUAD ? 4 : 1)) {
            const uint16_t e[4] = {(uint16_t)(e01 & 0xffffu), (uint16_t)(e01 >> 16),
                                   (uint16_t)(e23 & 0xffffu), (uint16_t)(e23 >> 16)};
            // The definition (what torch computes on bf16 tensors, pinned by the qhash_* fixtures): nrm = f32 sqrt of the
            // f32-rounded EXACT sum of squares, nb = bf16(nrm), element = bf16(IEEE f32 quotient).  The exact sequence -- f64
            // squares, an f64 wave sum, an f64 sqrt, two IEEE divisions -- is ~130 dependent instructions on the one wave
            // everybody waits for (~0.5 us).  Round 4: a FAST form first -- f32 sum (squares of bf16 numbers are exact in
            // f32; <= 7 roundings), v_sqrt_f32 (1 ulp), v_rcp_f32 + multiply (<= 2.5 ulp from the IEEE quotient) -- which
            // gives the same bf16 numbers unless a value lies within a few f32 ulps of a bf16 rounding boundary (low 16 bits
            // at 0x8000): there, and only there (0.1 % of the rows for the norm, 3 % for some element), the wave takes the
            // exact sequence.  Both tests are wave-uniform; the guards (32 / 16 ulps) are several times the error bounds.
            float xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = bf16_bits_to_f32(e[i]);   // elements past `per` are zero
            MP_STAMP(stamp, 28);                               // the query row has arrived
            float nrm, nb;
            bool fast = ha.exact_norm == 0;
            if (fast) {
                float sf = fmaf(xf[3], xf[3], fmaf(xf[2], xf[2], fmaf(xf[1], xf[1], xf[0] * xf[0])));
                sf = wave_sum(sf);
                float s1 = __builtin_amdgcn_sqrtf(sf);
                s1 = fmaf(fmaf(-s1, s1, sf), 0.5f * __builtin_amdgcn_rcpf(s1), s1);   // one Newton step: within ~1 ulp of sqrt(sf)
                const uint32_t low = __float_as_uint(s1) & 0xffffu;
                const bool edge = (low >= 0x8000u - 32u && low <= 0x8000u + 32u) || !(sf > 1e-30f && sf < 1e30f);
                fast = !edge;                                  // uniform: sf is the same on every lane
                nrm = s1;
            }
            if (!fast) {
                double ss = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) ss += (double)xf[i] * (double)xf[i];   // exact, order-free
                ss = wave_sum(ss);
                nrm = (float)sqrt((double)(float)ss);          // (__fsqrt_rn measured no faster here, and not bit-identical)
            }
            nb = bf16_bits_to_f32(f32_to_bf16_rne(nrm));
            if (!LEAN && lane == 0 && lead && own_row && ha.qnorm_out != nullptr) ha.qnorm_out[h] = nrm;
            uint16_t* dst = reinterpret_cast<uint16_t*>(s_q) + lane * per;
            uint16_t* dst4 = reinterpret_cast<uint16_t*>(s_q4 + wave * 64) + lane * per;
            uint16_t* raw = reinterpret_cast<uint16_t*>(s_qraw) + lane * per;
            float tq[4];
            bool amb = false;
            if (fast) {
                const float rc = __builtin_amdgcn_rcpf(nb);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    tq[i] = xf[i] * rc;
                    const uint32_t tb = __float_as_uint(tq[i]);
                    const uint32_t lo16 = tb & 0xffffu;
                    // near a rounding boundary, or so small that the reciprocal form may flush where the quotient does not
                    amb = amb || (i < per && xf[i] != 0.f &&
                                  ((lo16 >= 0x8000u - 16u && lo16 <= 0x8000u + 16u) || (tb & 0x7f800000u) < (32u << 23)));
                }
                fast = __ballot(amb) == 0ull;                  // uniform
            }
            if (!fast) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tq[i] = __fdiv_rn(xf[i], nb);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < per) {
                    const uint16_t nq = f32_to_bf16_rne(tq[i]);
                    if (QUAD) dst4[i] = nq;
                    if (own_row) {
                        dst[i] = nq;
                        if (AD > 0) raw[i] = e[i];
                    }
                }
            // ||nq|| <= ||q||/nb * (1 + 2^-8): every element is rounded to bf16 once (guard bound)
            if (lane == 0) {
                if (QUAD) s_rn4[wave] = (nrm / nb) * 1.005f;
                if (own_row) {
                    s_rn[0] = (nrm / nb) * 1.005f;
                    s_rn[1] = nrm;
                }
            }
            MP_STAMP(stamp, 29);                               // normalised row written to LDS
        }
        __syncthreads();
        MP_STAMP(stamp, 22);
        // -- one hyperplane per thread per pass: f32 FMA chain over D, exact-sign guard; the next
        //    pass's chunks are fetched into the registers this pass has just consumed
        const float rn = *s_rn;
        const u32x4* q4 = reinterpret_cast<const u32x4*>(s_q);
        // exact sign of the columns inside the guard band, one column at a time by the WHOLE wave: lane j
        // takes elements j*per .. of the dot product (exact f64 products of bf16 pairs), a DPP wave sum adds
        // them (exact and order-free).  One parallel round of loads: ~0.6 us.  The same loop run by the owning
        // thread alone (16 dependent 16-byte loads) took ~2.5 us, and with eight members per head almost every
        // cluster had such a member somewhere -- it was the tail of the whole launch.
        auto exact_signs = [&](bool near, int col0, bool& bit) {            // col0: column of this wave's lane 0
            for (unsigned long long fm = __ballot(near); fm; fm &= fm - 1) {   // wave-uniform
                const int src = __ffsll((long long)fm) - 1;
                const int cs = col0 + src;
                const int d0 = lane * per;
                const uint16_t* wp = ha.Wk + ((int64_t)(d0 >> 3) * ha.KLpad + cs) * 8 + (d0 & 7);
                const uint16_t* qp = reinterpret_cast<const uint16_t*>(s_q) + d0;
                double part = 0.0;
                for (int i = 0; i < per; ++i)
                    part += (double)bf16_bits_to_f32(qp[i]) * (double)bf16_bits_to_f32(wp[i]);
                const double ex = wave_sum(part);
                if (lane == src) bit = ex > 0.0;
            }
        };
        auto finish_column = [&](int c, float& acc, bool& bit, bool& near) {
            if (c < KL) {
                for (int kc = CH; kc < chunks; ++kc)                // head_dim 256: remaining chunks
                    dot8_bf16_chain(acc, q4[kc], Wk4[c + (int64_t)kc * ha.KLpad]);
                dot_settle(acc);
                bit = acc > 0.f;
                const float wn = (c == col_first) ? wn_first : ha.wnorm[c];
                near = fabsf(acc) <= (1.0f / 65536.0f) * rn * wn;    // 2^-16 guard band (simhash.hip SH_EPS)
            }
        };
        bool have_bits = false;                                  // (every workgroup barrier below orders LDS only)
        if constexpr (QUAD) {
            // ---- tile t = qm + 4 wave (32 planes) x the quad's four rows: C' = W X^T by v_mfma_f32_32x32x16_bf16 -- A = the
            // planes (plane-major copy Wt: lane (n, kh) holds plane n's k-chunk kh of every 16-wide step), B = the rows
            // (row n < 4 from LDS, rows 4 .. 31 zero: 7/8 of the tile is idle, which costs nothing: the pipe has nothing else
            // to do); lane (n, kh) ends up with C[plane (i / 4) 8 + kh 4 + i % 4][row n], i < 16.  Guard band as everywhere
            // (2^-16 ||nq|| ||w||, here against the widest plane of the tile), exact f64 recomputation inside it.
            // (the plane-major copy travels in AttnArgs::rows2, which only the stand-alone retrieve uses otherwise: a field of
            // its own would grow the argument block of every instantiation -- measured +0.1 us at cfg 1 in round 5)
            const uint16_t* Wt5 = reinterpret_cast<const uint16_t*>(aa.rows2);
            const int T2 = 2 * U;                                    // 32-bit words (= tiles) the head's workgroup polls
            const int tile = qm + 4 * wave;
            const int n = lane & 31, kh = lane >> 5;
            const uint32_t kq = (uint32_t)s_tmp[28];                 // (drawn in front of the barrier behind the normalisation)
            if (tile < T2 && aa.xmode != 2) {
                const uint16_t* wrow = Wt5 + (int64_t)(tile * 32 + n) * D + kh * 8;
                u32x4 av[8], bv[8];
#pragma unroll
                for (int st8 = 0; st8 < 8; ++st8) av[st8] = *reinterpret_cast<const u32x4*>(wrow + st8 * 16);
                const float wn_n = ha.wnorm[tile * 32 + n];
#pragma unroll
                for (int st8 = 0; st8 < 8; ++st8)
                    bv[st8] = n < 4 ? *reinterpret_cast<const u32x4*>(s_q4 + n * 64 + st8 * 8 + kh * 4) : u32x4{0u, 0u, 0u, 0u};
                f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st8 = 0; st8 < 8; ++st8)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[st8]), __builtin_bit_cast(bf16x8, bv[st8]),
                                                                c, 0, 0, 0);
                const float band = (1.0f / 65536.0f) * s_rn4[n & 3] * wave_max(wn_n);
                uint32_t bits = 0u, nearm = 0u;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int pb = (i / 4) * 8 + kh * 4 + (i % 4);
                    bits |= (c[i] > 0.f ? 1u : 0u) << pb;
                    nearm |= (fabsf(c[i]) <= band ? 1u : 0u) << pb;
                }
                // planes past K L are padding (zero columns: their exact product IS zero, bit 0 -- nothing to recompute; left
                // in, the all-padding tile at the end of the row cost its wave 128 recomputations and its peers a time-out)
                const int live = KL - tile * 32;
                const uint32_t livem = live >= 32 ? 0xffffffffu : (live <= 0 ? 0u : (1u << live) - 1u);
                bits &= livem;
                nearm &= livem;
                if (n >= 4) nearm = 0u;
                for (unsigned long long fm = __ballot(nearm != 0u); fm; fm &= fm - 1) {   // wave-uniform: rare
                    const int src = __ffsll((long long)fm) - 1;
                    uint32_t left = (uint32_t)__builtin_amdgcn_readlane((int)nearm, src);
                    const int row = src & 3;
                    while (left) {
                        const int pb = __ffs((int)left) - 1;
                        left &= left - 1;
                        const uint32_t qw = s_q4[row * 64 + lane];                            // elements 2 lane, 2 lane + 1
                        const uint32_t ww = *reinterpret_cast<const uint32_t*>(Wt5 + (int64_t)(tile * 32 + pb) * D + 2 * lane);
                        const double part = (double)bf16_lo(qw) * (double)bf16_lo(ww) + (double)bf16_hi(qw) * (double)bf16_hi(ww);
                        const double ex = wave_sum(part);
                        if (lane == src) bits = (bits & ~(1u << pb)) | ((ex > 0.0 ? 1u : 0u) << pb);
                    }
                }
                bits |= (uint32_t)__shfl_xor((int)bits, 32);         // the two k-halves hold disjoint planes of the tile
                MP_STAMP(stamp, 23);                                 // own tile evaluated
                if (lane < 4) {                                      // row `lane`'s 32 sign bits of this tile -> that row's head
                    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(       // (one descriptor over the block of 32 heads)
                        aa.xw + (h & ~(int64_t)31) * aa.xwords, 0, 32 * aa.xwords * 8, 0x00020000);
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 pk = {bits, kq};
                    __builtin_amdgcn_raw_buffer_store_b64(pk, rx, (((int)(h & 7) + 8 * lane) * aa.xwords + tile) * 8, 0, 1);   // sc0: to the XCD's L2
                }
            }
            unsigned long long* xh = aa.xw + h * aa.xwords;
            if (tid < T2) {
                unsigned long long v = 0;
                bool ok = false;
                for (int it = 0; it < 96 && !ok; ++it) {             // bounded: a member may only wait for peers that are running
                    v = __hip_atomic_load(xh + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (uint32_t)(v >> 32) == kq;
                }
                s_bits[tid] = (uint32_t)v;
                if (!ok) s_tmp[29] = 0;
            }
            __syncthreads();
            MP_STAMP(stamp, 24);
            have_bits = s_tmp[29] != 0;                              // uniform
            bits_exchanged = have_bits;
            if (!have_bits) load_planes(tid);                        // a peer is not running: this head hashes alone
        }
        if (split) {
            // ---- this wave's unit (if it has one), published as two 64-bit words (sequence << 32 | 32 sign bits):
            // a word proves by itself that it belongs to THIS launch, so there is no counter, no acknowledgement and
            // no ticket -- a store, then returning L2 atomics that poll the head's 2 U words until every one carries
            // the sequence number.  A workgroup may only wait for peers that are running: the wait is bounded, and a
            // member that gives up hashes all planes itself (always correct, never blocked).
            bool bit = false, near = false;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) dot8_bf16_chain(acc, q4[i], w[i]);
            if (unit < U) finish_column((unit << 6) + lane, acc, bit, near);
            exact_signs(unit < U && near, unit << 6, bit);
            const unsigned long long bm = __ballot(bit);
            MP_STAMP(stamp, 23);                                     // own unit evaluated
            unsigned long long* xh = aa.xw + h * aa.xwords;
            if (unit < U && aa.xmode != 2 && lane == 0) {
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xh, 0, aa.xwords * 8, 0x00020000);
                const u32x4 pk = {(uint32_t)bm, seq, (uint32_t)(bm >> 32), seq};
                __builtin_amdgcn_raw_buffer_store_b128(pk, rx, unit * 16, 0, 1);          // sc0: to the XCD's L2
            }
            if (tid < 2 * U) {
                unsigned long long v = 0;
                bool ok = false;
                for (int it = 0; it < 96 && !ok; ++it) {             // ~0.13 us per poll
                    // polled with AGENT-scope loads (sc1): they bypass this CU's L1 and are served by the XCD's L2,
                    // where the peers' stores land 0.45 us after they were issued, however many workgroups poll the
                    // same lines (scripts/probes/l2_poll.hip).  A workgroup-scope load (sc0) is served by the L1 from
                    // the first poll's line for ever, and so is an atomic "or 0", which the compiler turns into one.
                    v = __hip_atomic_load(xh + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (uint32_t)(v >> 32) == seq;
                }
                s_bits[tid] = (uint32_t)v;
                if (!ok) s_tmp[29] = 0;                              // set to 1 with the other LDS state at the top
            }
            __syncthreads();
            MP_STAMP(stamp, 24);                                     // every word of the head is there (or timed out)
            have_bits = s_tmp[29] != 0;                              // uniform
            bits_exchanged = have_bits;
            if (!have_bits) load_planes(tid);                        // on its own after all: pass 0's chunks
        }
        if (!have_bits)
        for (int c0 = 0; c0 < KL; c0 += RT_THREADS) {
            const int c = c0 + tid, cn = c + RT_THREADS;
            const int cnc = cn < ha.KLpad ? cn : ha.KLpad - 1;
            bool bit = false, near = false;
            float acc = 0.f;
            if (c0 + RT_THREADS < KL) {                              // uniform: another pass follows
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    dot8_bf16_chain(acc, q4[i], w[i]);              // q4[i]: LDS broadcast
                    w[i] = Wk4[cnc + (int64_t)i * ha.KLpad];
                }
            } else {
#pragma unroll
                for (int i = 0; i < CH; ++i) dot8_bf16_chain(acc, q4[i], w[i]);
            }
            finish_column(c, acc, bit, near);
            exact_signs(near, c0 + (wave << 6), bit);
            const unsigned long long bm = __ballot(bit);
            if (lane == 0) {
                s_bits[(c0 >> 5) + wave * 2] = (uint32_t)bm;
                s_bits[(c0 >> 5) + wave * 2 + 1] = (uint32_t)(bm >> 32);
            }
            MP_STAMP(stamp, 23 + (c0 >> 10));
        }
    }
    // (split hash, every word arrived: the barrier behind the exchange already ordered the sign bits -- and the one behind
    // the normalisation the zero-filled bitmaps -- in front of everything below; one barrier less on the chain)
    if (!bits_exchanged) __syncthreads();
    MP_STAMP(stamp, 27);
    const uint32_t T0 = (uint32_t)t0;                                   // M <= 2^22 (mp_lsh_alloc)
    auto apply = [&](int32_t t) {
        const uint32_t u = ((uint32_t)t & idmask) - T0;                 // token index inside the range
        if (t != -1 && u < tlen) {                                      // (-1: a lane without an id) t0 <= id < t0 + tlen
            const uint32_t bit = 1u << (u & 31);
            const uint32_t old = atomicOr(&bmA[u >> 5], bit);           // first hit: 0 -> 1
            if (old & bit) {
                atomicOr(&bmB[u >> 5], bit);                            // any later hit: -> 2
                if (pay) s_kn[u] = (uint16_t)((uint32_t)t >> idbits);   // (every hit of a token carries the same norm)
            }
        }
    };
    // ---- LEAN counting: per-token COUNTERS instead of the two bitmaps, a batch of ids at a time.  Stage 1 (lean_first):
    // every id's counter is incremented by one returning LDS atomic, issued as the id's load arrives, results left in
    // registers.  Stage 2 (lean_rest): the lanes whose atomic returned 1 -- this was the token's SECOND hit: exactly one
    // lane per selected token, whatever the interleaving -- append the token's table word (id | key norm) to the WAVE's own
    // list: position = the wave's cursor (a scalar) + mbcnt of the ballot; no atomic, nothing shared.
    // The stage-1 atomics are issued by inline asm under an explicit lane mask: (a) only the lanes that hold an id take
    // part -- a first form with unconditional atomics (idle lanes ORing 0 into words of their own) cost cfg 3 +2 us per
    // launch: LDS atomics are paid per active lane; (b) an LDS instruction the COMPILER sees under a branch makes it wait
    // for every outstanding one at the join, which would serialise the batch again.  The compiler does not count them, so
    // lean_rest starts with an explicit s_waitcnt; its own waits stay correct (LDS operations return in order: waiting
    // for its k youngest covers everything older, counted or not).
    int* s_res = s_tmp + 27;                                                 // entries reserved in the list (incl. those beyond the stage)
    int* s_head = s_tmp + 25;                                                // entries claimed by gathering waves
    int* s_done = s_tmp + 24;                                                // waves that have finished counting
    const uint32_t lds0 = (uint32_t)(uintptr_t)s_u32;                        // LDS byte offset of the counting area
    const uint32_t cmask = (1u << (1u << csh)) - 1u;                         // 0xff / 0xffff
    const int lcap = AD > 0 ? aa.cap : 0;                                    // entries of the stage (4 096)
    int32_t* spill_rows = results + h * M + t0;                              // the member's columns of the head's result row: >= its tokens
    int nsp = 0;                                                             // the last wave alone: entries it has put on the spill list
    bool spilled = false;                                                    // uniform: this wave has stores to the spill list in flight
    auto lean_first = [&](int32_t t, uint32_t& u, uint32_t& old) {
        const uint32_t uu = ((uint32_t)t & idmask) - T0;
        const bool ok = t != -1 && uu < tlen;
        u = uu;
        old = 0u;
        const unsigned long long m = __ballot(ok);
        const uint32_t addr = lds0 + ((uu >> (5 - csh)) << 2);
        const uint32_t one = 1u << ((uu & ((32u >> csh) - 1u)) << csh);
        unsigned long long sv;
        // (m is a subset of exec: plain moves, which leave SCC alone -- s_and_saveexec would clobber a condition the
        // compiler may hold across the asm)
        asm volatile("s_mov_b64 %1, exec\n\t"
                     "s_mov_b64 exec, %2\n\t"
                     "ds_add_rtn_u32 %0, %3, %4\n\t"
                     "s_mov_b64 exec, %1"
                     : "+v"(old), "=&s"(sv)
                     : "s"(m), "v"(addr), "v"(one)
                     : "memory");
    };
    // alone = false: reserve in the workgroup's list; alone = true (the wave that drew the last ticket, nobody left to claim):
    // straight to the spill list, positions from a register
    auto lean_rest = [&](auto& t, auto& u, auto& old, bool alone) {
        constexpr int N = (int)std::extent<typename std::remove_reference<decltype(old)>::type>::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // every counter of the batch has answered
        int run = 0;                                                         // wave-uniform
        int off[N];
#pragma unroll
        for (int b = 0; b < N; ++b) {
            // the counter stood at 1: this lane's id is its token's second hit (a lane without an id kept old = 0)
            const bool second = ((old[b] >> ((u[b] & ((32u >> csh) - 1u)) << csh)) & cmask) == 1u;
            const unsigned long long bal = __ballot(second);
            off[b] = second ? run + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u)) : -1;
            run += __popcll(bal);
        }
        if (run > 0) {                                                       // uniform
            int base = lcap + nsp;
            if (!alone) {
                if (lane == 0) base = atomicAdd(s_res, run);
                base = __builtin_amdgcn_readfirstlane(base);
            } else {
                nsp += run;
            }
            spilled = spilled || base + run > lcap;
#pragma unroll
            for (int b = 0; b < N; ++b) {
                if (off[b] >= 0) {
                    const int pos = base + off[b];
                    if (pos < lcap) s_ids[pos] = t[b];
                    else spill_rows[pos - lcap] = t[b];                      // (rare: K = 1-like data, the chunk pool)
                }
            }
        }
    };
    auto lean_apply1 = [&](int32_t t, bool alone) {                          // a batch of one (the rare sweeps)
        int32_t tt[1] = {t};
        uint32_t uu[1], oo[1];
        lean_first(t, uu[0], oo[0]);
        lean_rest(tt, uu, oo, alone);
    };
    auto code_of = [&](int l) {   // HASH 1: bit i of code l <- plane l*K + i
        if (HASH == 2) return (int)ha.codes_out[h * L + l];
        const int bp = l * ha.K, w = bp >> 5, sh = bp & 31;
        uint32_t v = s_bits[w] >> sh;
        if (sh + ha.K > 32) v |= s_bits[w + 1] << (32 - sh);
        return (int)(v & ((1u << ha.K) - 1u));
    };
    auto code_fast = [&](int l) {   // code_of without a branch: both words, one funnel shift (s_bits has slack words)
        if (HASH == 2) return (int)ha.codes_out[h * L + l];
        const uint32_t bp = (uint32_t)l * (uint32_t)ha.K, w = bp >> 5;
        return (int)(__funnelshift_r(s_bits[w], s_bits[w + 1], bp & 31u) & ((1u << ha.K) - 1u));
    };
    if (AD == 0 && HASH == 1 && bounds == nullptr) {
        // hash-only launch (mp_simhash_query on a handful of rows: one workgroup per row on the vector pipes is ~3.5 us
        // in-kernel where the 32-row MFMA tiles of simhash_query_kernel, ten workgroups for 32 rows, take ~8)
        for (int l = tid; l < L; l += RT_THREADS) ha.codes_out[h * L + l] = code_of(l);
        MP_STAMP_FLUSH(stamp);
        return;
    }
    if (st_live) {                                                  // (the four state words requested at the head of the kernel)
        const int ib = __builtin_amdgcn_readfirstlane((int)st_ib);
        const int bad = __builtin_amdgcn_readfirstlane((int)st_bad);
        const uint32_t av = (uint32_t)__builtin_amdgcn_readfirstlane((int)st_av);
        const uint32_t kv = (uint32_t)__builtin_amdgcn_readfirstlane((int)st_kv);
        idbits = ib;
        pay = (ib != 0) & (bad == 0) & (av == kv);
        idmask = idbits ? ((1u << idbits) - 1u) : 0xffffffffu;
    }
    const int32_t* slots = (AD > 0) ? aa.slots : nullptr;
    // direct pieces: the barrier behind the pass, and the chunk pool of what is longer than slot + follow-up (skewed data)
    // (LEAN: run by ONE wave -- the one that drew the last ticket, behind everybody's counting -- without barriers)
    auto direct_pool_build = [&]() {
        if constexpr (!LEAN) lds_barrier();
        MP_STAMP(stamp, 17);
        if (s_tmp[30] > 0) {                                            // uniform; pieces longer than 30 + 96 ids
            for (int l = LEAN ? lane : tid; l < L; l += LEAN ? WAVE : RT_THREADS) {
                const int len = s_len[l];                               // start and length were checked in the pass
                if (len > 0) {
                    const int nch = (len + 63) >> 6;
                    const int base = atomicAdd(s_ntail, nch);
                    for (int c = 0; c < nch && base + c < RT_TAIL_CAP; ++c)
                        s_tail[base + c] = ((uint32_t)l << 16) | (uint32_t)c;
                }
            }
            if constexpr (!LEAN) __syncthreads();
        }
        MP_STAMP(stamp, 18);
    };
    const bool direct_path = AD > 0 && HASH != 0 && slots != nullptr;   // uniform
    if (direct_path) {
        // ---- DIRECT pieces: the piece (table l, bucket code, range rank) has a 128-byte slot holding its length,
        // position and first 30 ids, so ONE dependent round trip (hash -> slot) replaces two (hash -> sub-bounds ->
        // ids).  Half a wave reads a slot; a wave keeps DG loads = 2 DG pieces in flight.  A piece longer than 30
        // ids is finished by its half-wave with one more access (below); the slots exist only where the mean piece
        // is <= 12.5 ids.  (256-byte slots holding 63 ids need no second access and measured slower: 23.8 against
        // 22.9 us per layer at cfg 1, 28.6 against 27.3 at cfg 4 -- twice the bytes of random 128-byte reads.)
        // A slot is SW = 32, 16 or 8 words (aa.slot_log2; chosen at alloc from the mean piece length M / (2^K R)): its
        // first two words are the piece's length and position, the rest its first SW - 2 ids.  A GROUP of SW lanes reads a
        // slot, so one load instruction fetches 64 / SW slots: with clusters of 16 / 32 workgroups per head (cfg 4: mean
        // piece 4 / 2 ids, L = 300) a wave's 19 tables are 5 / 3 loads of 64- / 32-byte slots instead of 10 of 128 bytes.
        auto direct_pass = [&](auto dg_tag, auto sw_tag) {
        constexpr int DG = decltype(dg_tag)::value;
        constexpr int SWL = decltype(sw_tag)::value;                    // log2 of the slot width in words
        constexpr int SW = 1 << SWL, GPW = 64 / SW;                     // groups (= slots) per load instruction
        constexpr int SLOT_IDS = SW - 2, SLOT_MORE = 3 * SW;            // ids in the slot / fetched by the group itself behind it
        const int grp = lane >> SWL, sl = lane & (SW - 1);
        const int32_t* sg = slots + ((int64_t)g * L * NB * R + rank) * SW;
        // the value lane (group start + k) holds, k = 0 (length) or 1 (position), for every lane of the group
        auto group_word = [&](int32_t v, auto k_tag) -> int {
            constexpr int KK = decltype(k_tag)::value;
            if constexpr (SW == 32) {
                const int a0 = __builtin_amdgcn_readlane(v, KK), a1 = __builtin_amdgcn_readlane(v, 32 + KK);
                return grp ? a1 : a0;
            } else if constexpr (SW == 16) {                            // a group is one DPP row
                return __builtin_amdgcn_update_dpp(0, v, 0x150 + KK, 0xf, 0xf, false);            // row_newbcast:KK
            } else {                                                    // two groups per DPP row
                const int lo = __builtin_amdgcn_update_dpp(0, v, 0x150 + KK, 0xf, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(0, v, 0x150 + 8 + KK, 0xf, 0xf, false);
                return (lane & 8) ? hi : lo;
            }
        };
        for (int l0 = 0; l0 < L; l0 += RT_WAVES * GPW * DG) {
            int32_t v[DG];
            int cd[DG];
            uint32_t at[DG];
            // three straight-line rounds -- codes (two LDS words + one funnel shift each), slot offsets (shifts only:
            // NB = 2^K, R = 2^clog, 2^SWL-word slots; the host keeps a group's slots under 2^31 words), loads -- and the
            // stores of the codes after them.  As one loop body per piece (a branch around the second LDS word, 64-bit
            // multiplies, a branch around the store) the compiler emitted six dependent LDS round trips in front of
            // the loads: 1.5 us between "sign bits in LDS" and "loads issued" (scripts/phase_times.py), now 0.3.
#pragma unroll
            for (int b = 0; b < DG; ++b) {
                const int l = l0 + (b * RT_WAVES + wave) * GPW + grp;
                const int lc = l < L ? l : L - 1;                       // loads stay unconditional
                cd[b] = code_fast(lc);
                at[b] = (((((uint32_t)lc << ha.K) + (uint32_t)cd[b]) << clog) << SWL) + (uint32_t)sl;
            }
#pragma unroll
            for (int b = 0; b < DG; ++b) v[b] = MP_TLOAD(sg + at[b]);
            MP_STAMP(stamp, 42);                                     // slot loads issued
            if (!LEAN && (HASH == 1 || HASH == 3 || HASH == 5) && lead && sl == 0) {
#pragma unroll
                for (int b = 0; b < DG; ++b) {
                    const int l = l0 + (b * RT_WAVES + wave) * GPW + grp;
                    if (l < L) ha.codes_out[h * L + l] = cd[b];
                }
            }
            // A piece longer than its slot.  Not rare: SimHash buckets are far from equally likely (ten random planes in
            // 128 dimensions: bucket sizes at cfg 1 run from 32 (p1) to 209 (p99) around a mean of 96) and a query lands
            // in the heavy ones more often -- 1.5 % of the probed pieces on isotropic keys, 5.9 % on the clustered
            // workload (bench.py --data clustered), i.e. two to three per wave somewhere in almost every cluster.  Its
            // group fetches up to 3 SW more ids itself (the slot carries the position): ONE more dependent access for
            // the whole wave -- the follow-up loads of ALL its long pieces are issued while the slots are still being
            // counted and are waited for together (round 2 waited for each long piece's loads on the spot: a wave
            // with three long pieces paid three dependent round trips, and the launch waits for its slowest wave).
            // What is longer still (skewed data) goes to the chunk pool.
            int32_t e0[DG], e1[DG], e2[DG];
            int r1s[DG];
            uint32_t more = 0u, wide = 0u;                                  // wave-uniform bit masks over b
            uint32_t lu[LEAN ? DG : 1], lo[LEAN ? DG : 1];                  // LEAN: token index / first-hit result per slot id
#pragma unroll
            for (int b = 0; b < DG; ++b) {
                const int l = l0 + (b * RT_WAVES + wave) * GPW + grp;
                const int pl = group_word(v[b], std::integral_constant<int, 0>{});   // length of the piece
                const int pp = group_word(v[b], std::integral_constant<int, 1>{});   // its position in the table row
                if (b == 0) MP_STAMP(stamp, 43);                        // the first slot has arrived
                if (b == DG - 1) MP_STAMP(stamp, 44);                   // the last one has
                int rest = 0;
                if (l < L) {
                    rest = pl - SLOT_IDS;
                    if (pp < 0 || (int64_t)pp + pl > M) rest = 0;       // never outside the row
                }
                const int r1 = rest < SLOT_MORE ? rest : SLOT_MORE;
                r1s[b] = r1;
                e0[b] = e1[b] = e2[b] = -1;
                if (__ballot(rest > 0)) {                               // wave-uniform: the follow-up goes out NOW
                    more |= 1u << b;
                    const int lc = l < L ? l : L - 1;
                    const int32_t* row = tab + (int64_t)lc * M;
                    const int at0 = pp + SLOT_IDS + sl;
                    e0[b] = MP_TLOAD(row + (sl < r1 ? at0 : 0));
                    if (__ballot(r1 > SW)) {
                        wide |= 1u << b;
                        e1[b] = MP_TLOAD(row + (sl + SW < r1 ? at0 + SW : 0));
                        e2[b] = MP_TLOAD(row + (sl + 2 * SW < r1 ? at0 + 2 * SW : 0));
                    }
                    if (sl == 0 && rest > SLOT_MORE) {                  // skewed data: the chunk pool takes the rest
                        s_start[l] = pp + SLOT_IDS + SLOT_MORE;
                        s_len[l] = rest - SLOT_MORE;
                        atomicAdd(&s_tmp[30], 1);
                    }
                }
                if constexpr (LEAN) {
                    if (!(l < L && sl >= 2 && sl - 2 < pl)) v[b] = -1;
                    lean_first(v[b], lu[b], lo[b]);                     // the slot's own ids: stage 1 as the slot arrives
                } else {
                    if (l < L && sl >= 2 && sl - 2 < pl) apply(v[b]);   // the slot's own ids
                }
            }
            if constexpr (LEAN) lean_rest(v, lu, lo, false);             // stage 2 for the DG slots together
            if (more) {                                                 // wave-uniform; one wait for all follow-ups
                if constexpr (LEAN) {
                    // e0 of every b as ONE batch (lanes without a follow-up id idle), then -- rare -- e1 and e2 as another
#pragma unroll
                    for (int b = 0; b < DG; ++b) {
                        e0[b] = ((more >> b) & 1u) && sl < r1s[b] ? e0[b] : -1;
                        lean_first(e0[b], lu[b], lo[b]);
                    }
                    lean_rest(e0, lu, lo, false);
                    if (wide) {                                         // uniform
#pragma unroll
                        for (int b = 0; b < DG; ++b) {
                            e1[b] = ((wide >> b) & 1u) && sl + SW < r1s[b] ? e1[b] : -1;
                            lean_first(e1[b], lu[b], lo[b]);
                        }
                        lean_rest(e1, lu, lo, false);
#pragma unroll
                        for (int b = 0; b < DG; ++b) {
                            e2[b] = ((wide >> b) & 1u) && sl + 2 * SW < r1s[b] ? e2[b] : -1;
                            lean_first(e2[b], lu[b], lo[b]);
                        }
                        lean_rest(e2, lu, lo, false);
                    }
                } else {
#pragma unroll
                for (int b = 0; b < DG; ++b) {
                    if (more & (1u << b)) {
                        apply(sl < r1s[b] ? e0[b] : -1);
                        if (wide & (1u << b)) {
                            apply(sl + SW < r1s[b] ? e1[b] : -1);
                            apply(sl + 2 * SW < r1s[b] ? e2[b] : -1);
                        }
                    }
                }
                }
            }
        }
        };
        // loads in flight per wave: enough to cover L tables in ONE round where it fits (a second round is a second
        // dependent round trip): 128-byte slots 6 (L <= 192) or 10; 64-byte slots 5 (L <= 320); 32-byte slots 3 (L <= 384)
        if (aa.slot_log2 == 3) direct_pass(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});
        else if (aa.slot_log2 == 4) direct_pass(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
        else if (L > RT_WAVES * 2 * 6) direct_pass(std::integral_constant<int, 10>{}, std::integral_constant<int, 5>{});
        else direct_pass(std::integral_constant<int, 6>{}, std::integral_constant<int, 5>{});
        MP_STAMP(stamp, 45);                                            // this wave's pieces counted
        if constexpr (!LEAN) direct_pool_build();                       // (LEAN: behind the wave's own gather, below)
    } else {
    // probe: the two sub-bounds of this workgroup's token range, adjacent 4-byte words of the bucket's record
    // (issued first: longest latency)
    for (int l = tid; l < Lpad; l += RT_THREADS) {
        int st = 0, len = 0;
        if (l < L) {
            int code;
            if (HASH != 0) {
                code = code_of(l);
                if (!LEAN && (HASH == 1 || HASH == 3 || HASH == 5) && lead) ha.codes_out[h * L + l] = code;
            } else {
                code = query[h * L + l];
            }
            if (code >= 0 && code < NB) {
                const int32_t* rec = bnd + ((int64_t)l * NB + code) * RS;
                const int lo = MP_TLOAD(rec + e_lo), hi = MP_TLOAD(rec + e_hi);
                st = lo;
                len = hi - lo;
                if (st < 0 || len < 0 || (int64_t)st + len > M) len = 0;
            }
        }
        s_start[l] = st;
        s_len[l] = len;
        if (len > 128) {   // 64-id chunks beyond the first two go to a shared pool (skewed buckets)
            const int nch = (len - 128 + 63) >> 6;
            const int base = atomicAdd(s_ntail, nch);
            for (int c = 0; c < nch && base + c < RT_TAIL_CAP; ++c)
                s_tail[base + c] = ((uint32_t)l << 16) | (uint32_t)(c + 2);
        }
    }
    lds_barrier();
    MP_STAMP(stamp, 17);
    MP_STAMP(stamp, 18);

    // stream the probed pieces: wave w owns tables w, w+16, ...; per round it keeps RT_GROUP pieces x 2 chunks of 64 ids in
    // flight, then applies them.  Straight-line rounds: all (start, length) pairs out of LDS, then all loads, then all
    // applies.  As one loop body per piece with the loads under lane-divergent branches the compiler put an
    // s_waitcnt vmcnt(0) between the pieces: two, not twelve, were in flight.
    // Round 5 -- both chunks of every piece in ONE round trip.  Until round 4 the second 64 ids of a
    // piece were requested only behind the first chunks' applies: SimHash buckets are wide (p99 of a probed piece = 2.3 x
    // its mean; mean 32 ids at cfg 2 / 3), so ~4 % of the pieces are longer than 64 ids, almost every workgroup holds a
    // wave with such a piece, and the launch waits for the workgroup that paid the second dependent round trip.  Now both
    // chunks of all of a wave's pieces are requested before anything is applied: cfg 3 29.06 -> 28.43 us per layer, cfg 2
    // 32.36 -> 31.71, cfg 2 on clustered keys 41.42 -> 40.70 (same box, alternating regions: profiles/archive/r05_ab_stream_variants.txt).
    // The loads go through ONE buffer descriptor over the KV group's table rows: a lane past its piece gets an offset
    // beyond num_records, for which the hardware returns 0 WITHOUT a memory request -- the second-chunk loads of the 96 %
    // short pieces cost an instruction slot and no line (clamped addresses, the form used where a pointer is needed,
    // would re-request the piece's first line 12 times per wave).  Measured and not taken (= 1): the pool's first round
    // (ids beyond 128) in the same batch -- its descriptors come out of LDS in three dependent reads per chunk, in front of
    // the applies of everybody's first chunk: cfg 3 28.90, cfg 2 32.07, clustered 41.46 -- no better than round 4's form.
    {
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const uint64_t tab_bytes = (uint64_t)L * (uint64_t)M * 4ull;
    if (tab_bytes < (1ull << 32)) {                                          // uniform: 32-bit offsets cover the group's rows
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(tab), 0,
                                                                            (int)(uint32_t)tab_bytes, 0x00020000);
        constexpr int kNt = 2;                                               // aux bit 1 = nt on gfx940+
        constexpr uint32_t kOut = 0xfffffff0u;                               // beyond any num_records: no request, returns 0
        for (int l0 = wave_s; l0 < L; l0 += RT_WAVES * RT_GROUP) {
            int32_t id0[RT_GROUP], id1[RT_GROUP];
            int ln[RT_GROUP];
            uint32_t wb[RT_GROUP];
#pragma unroll
            for (int b = 0; b < RT_GROUP; ++b) {
                const int l = l0 + b * RT_WAVES;
                const int lc = l < L ? l : L - 1;
                ln[b] = s_len[lc];
                wb[b] = (uint32_t)s_start[lc];
            }
#pragma unroll
            for (int b = 0; b < RT_GROUP; ++b) {
                const int l = l0 + b * RT_WAVES;
                const int lc = l < L ? l : L - 1;
                ln[b] = l < L ? __builtin_amdgcn_readfirstlane(ln[b]) : 0;   // wave-uniform: one table per wave
                wb[b] = (uint32_t)lc * (uint32_t)M + (uint32_t)__builtin_amdgcn_readfirstlane((int)wb[b]);
                id0[b] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(
                    rt, lane < ln[b] ? (wb[b] + (uint32_t)lane) << 2 : kOut, 0, kNt);
            }
#pragma unroll
            for (int b = 0; b < RT_GROUP; ++b)
                id1[b] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(
                    rt, lane + 64 < ln[b] ? (wb[b] + (uint32_t)lane + 64u) << 2 : kOut, 0, kNt);
            // (the pool behind them as before: with its first round in this batch too -- form 1 of R5-1 -- its descriptors,
            // three dependent LDS reads per chunk, stood in front of everybody's first chunk and cost what the trip saved)
            // applied in issue order: the first piece's ids are counted while the last piece's are still on their way
            bool longer = false;
#pragma unroll
            for (int b = 0; b < RT_GROUP; ++b) longer = longer || ln[b] > 64;
            if constexpr (LEAN) {
                uint32_t lu[RT_GROUP], lo[RT_GROUP];
#pragma unroll
                for (int b = 0; b < RT_GROUP; ++b) {
                    if (!(lane < ln[b])) id0[b] = -1;
                    lean_first(id0[b], lu[b], lo[b]);
                }
                lean_rest(id0, lu, lo, false);
                if (longer) {                                                // wave-uniform (the loads are out already)
#pragma unroll
                    for (int b = 0; b < RT_GROUP; ++b) {
                        if (!(lane + 64 < ln[b])) id1[b] = -1;
                        lean_first(id1[b], lu[b], lo[b]);
                    }
                    lean_rest(id1, lu, lo, false);
                }
            } else {
#pragma unroll
            for (int b = 0; b < RT_GROUP; ++b) apply(lane < ln[b] ? id0[b] : -1);
            if (longer) {                                                    // wave-uniform (the loads are out already)
#pragma unroll
                for (int b = 0; b < RT_GROUP; ++b) apply(lane + 64 < ln[b] ? id1[b] : -1);
            }
            }
        }
    } else
    for (int l0 = wave_s; l0 < L; l0 += RT_WAVES * RT_GROUP) {
        int32_t id0[RT_GROUP], id1[RT_GROUP];
        int ln[RT_GROUP], sa[RT_GROUP];
#pragma unroll
        for (int b = 0; b < RT_GROUP; ++b) {
            const int l = l0 + b * RT_WAVES;
            const int lc = l < L ? l : L - 1;
            ln[b] = s_len[lc];
            sa[b] = s_start[lc];
        }
#pragma unroll
        for (int b = 0; b < RT_GROUP; ++b) {
            const int l = l0 + b * RT_WAVES;
            const int lc = l < L ? l : L - 1;
            ln[b] = l < L ? __builtin_amdgcn_readfirstlane(ln[b]) : 0;       // wave-uniform: one table per wave
            sa[b] = __builtin_amdgcn_readfirstlane(sa[b]);
            const int32_t* row = tab + (int64_t)lc * M + sa[b];
            id0[b] = MP_TLOAD(row + (lane < ln[b] ? lane : 0));
        }
        // applied in issue order: the first piece's ids are counted while the last piece's are still on their way
#pragma unroll
        for (int b = 0; b < RT_GROUP; ++b) {
            if constexpr (LEAN) lean_apply1(lane < ln[b] ? id0[b] : -1, false);   // (tables beyond 4 GB per KV group: not a tuned path)
            else apply(lane < ln[b] ? id0[b] : -1);
        }
        bool longer = false;
#pragma unroll
        for (int b = 0; b < RT_GROUP; ++b) longer = longer || ln[b] > 64;
        if (longer) {                                                        // wave-uniform
#pragma unroll
            for (int b = 0; b < RT_GROUP; ++b) {
                const int l = l0 + b * RT_WAVES;
                const int lc = l < L ? l : L - 1;
                const int32_t* row = tab + (int64_t)lc * M + sa[b];
                id1[b] = MP_TLOAD(row + (lane + 64 < ln[b] ? lane + 64 : 0));
            }
#pragma unroll
            for (int b = 0; b < RT_GROUP; ++b) {
                if constexpr (LEAN) lean_apply1(lane + 64 < ln[b] ? id1[b] : -1, false);
                else apply(lane + 64 < ln[b] ? id1[b] : -1);
            }
        }
    }
    }
    }
    // ---- LEAN: this wave is done counting -- it has nothing more to add to the list.  It now CLAIMS slices of the list
    // (16 entries where a head is split over a cluster, 32 otherwise) and gathers them, while other waves still count, until
    // every wave is done and the list is used up; then the window's share, then its state goes to LDS behind a ticket.  No
    // workgroup barrier from here on: the wave that draws the LAST ticket does what is left alone -- normally nothing but
    // the merge; with skewed keys the chunk pool (pieces beyond slot + follow-up, the sub-bounds path's ids beyond 128) and
    // the spill list.
    constexpr int ADL = AD > 0 ? AD : 64;
    AhState st_own = ah_state_init(lane, ADL / 8);
    int pre_total = 0;
    u32x4 qv_own = {0u, 0u, 0u, 0u};            // LEAN: this lane's eight query elements (read once, used again by the last wave)
    bool rare = true;                           // LEAN, uniform: the last wave has pooled chunks / a spill list to see to
    float m = 0.f, Z = 0.f, o0 = 0.f, o1 = 0.f;
    if constexpr (LEAN) {
        MP_STAMP(stamp, 33);
        if (lane == 0) (void)__hip_atomic_fetch_add(s_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32x4 qv_l = *reinterpret_cast<const u32x4*>(s_qraw + (lane % (ADL / 8)) * 4);
        qv_own = qv_l;
        const uint16_t* kv_l = aa.kv + g * M * 2 * ADL;
        const float* kn_l = aa.kn + g * M;
        constexpr int SHORT_L = (ADL == 128) ? 16 : AH_SLICE;
        // (the launcher uses this form only where a head is a cluster: the step is the short one -- 16 tokens at head_dim 128 --
        // and no 32-token instantiation of the fold sits in this kernel: its registers were the kernel's spills)
        const int folded = lean_claim_and_gather<ADL, SHORT_L>(st_own, s_done, s_res, s_head, s_ids, lcap, kv_l, kn_l, qv_l, s_rn + 1, M,
                                                               ha.K, L, idmask, idbits, pay, stamp);
        if (WIN && aa.win_kv != nullptr) {      // the static window: dense slices rank, rank + R, ... over the waves
            int wl = aa.win_len[h];
            wl = wl < 0 ? 0 : (wl > aa.win_M ? (int)aa.win_M : wl);
            auto none = [](int) { return u32x4{0u, 0u, 0u, 0u}; };
            if (wl > 0)
                attn_head_fold<ADL, RT_WAVES, true, AH_SLICE>(st_own, aa.win_kv + g * aa.win_M * 2 * ADL, nullptr, qv_l, 1.f,
                                                              wl, aa.win_M, 0, 0, rank, 1 << clog, none, nullptr, stamp);
        }
#if MP_STAMPS
        // per-wave: gathers done (slots 48 + wave; bits 0..9 of the 100 MHz stamp replaced by the entries the wave folded)
        if (stamp != nullptr && lane == 0)
            ::mp::s_stampbuf[48 + wave] = (wall_clock64() << 10) | (unsigned long long)(folded < 1023 ? folded : 1023);
#endif
        (void)folded;
        attn_head_publish<ADL>(st_own, s_merge);
        if (spilled) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's spill stores are acknowledged
        int tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(s_tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk != RT_WAVES - 1) {
            MP_STAMP_FLUSH(stamp);
            return;
        }
        MP_STAMP_L(stamp, 40);                                           // every wave's state is in LDS
        // ONE round of LDS reads for all the last wave normally needs: the sixteen states, the count, and the words that
        // say whether anything is left to do
        const int v30 = s_tmp[30], vnt = *s_ntail;
        pre_total = *s_res;
        attn_head_merge_read<ADL, RT_WAVES>(s_merge, m, Z, o0, o1);
        rare = (direct_path && v30 > 0) || vnt > 0 || pre_total > lcap;
        spilled = false;
        nsp = pre_total > lcap ? pre_total - lcap : 0;                   // the spill list so far: what did not fit the stage
        if (rare && direct_path) direct_pool_build();
    }
    // (LEAN: what follows is run by one wave)
    const int pw = LEAN ? 0 : wave;
    constexpr int PN = LEAN ? 1 : RT_WAVES, PT = LEAN ? WAVE : RT_THREADS;
    const int ptid = LEAN ? lane : tid;
    // ids beyond the first 128 of a piece (skewed data): pooled 64-id chunks, waves take them
    // round-robin with RT_TAIL_UNROLL loads in flight
    const int ntail = (LEAN && !rare) ? 0 : __builtin_amdgcn_readfirstlane(*s_ntail);
    if (ntail <= RT_TAIL_CAP) {
        for (int c0 = pw; c0 < ntail; c0 += PN * RT_TAIL_UNROLL) {
            int32_t idt[RT_TAIL_UNROLL];
#pragma unroll
            for (int u = 0; u < RT_TAIL_UNROLL; ++u) {
                const int c = c0 + u * PN;
                idt[u] = -1;
                if (c < ntail) {
                    const uint32_t d = s_tail[c];
                    const int l = d >> 16, j = ((d & 0xffffu) << 6) + lane;
                    if (j < s_len[l]) idt[u] = tab[(int64_t)l * M + s_start[l] + j];
                }
            }
            if constexpr (LEAN) {
                uint32_t lu[RT_TAIL_UNROLL], lo[RT_TAIL_UNROLL];
#pragma unroll
                for (int u = 0; u < RT_TAIL_UNROLL; ++u) lean_first(idt[u], lu[u], lo[u]);
                lean_rest(idt, lu, lo, true);                 // (the pool is the last wave's: spill list)
            } else {
#pragma unroll
            for (int u = 0; u < RT_TAIL_UNROLL; ++u) apply(idt[u]);
            }
        }
    } else {   // pool overflow (> 128 K extra ids per head): plain strided sweep of every long piece
        const int first = (AD > 0 && slots != nullptr) ? 0 : 128;      // direct mode: s_start is already past the slot's ids
        for (int l = 0; l < L; ++l) {
            const int len = s_len[l];
            if (len <= first) continue;
            const int32_t* row = tab + (int64_t)l * M + s_start[l];
            for (int j0 = first; j0 < len; j0 += PT) {                       // (uniform trip count: the LEAN form ballots)
                const int j = j0 + ptid;
                const int32_t t = j < len ? row[j] : -1;
                if constexpr (LEAN) lean_apply1(t, true);
                else apply(t);
            }
        }
    }
    if constexpr (LEAN) {
        if (spilled) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the pool's finds: this wave's own stores)
    } else {
    // (direct pass without pooled chunks: nothing was counted since the barrier behind the pass)
    if (!(direct_path && ntail == 0)) lds_barrier();
    }
    MP_STAMP(stamp, 19);

    // the stand-alone retrieve writes the head's list; a decode member writes ITS list at column t0 of the
    // head's row (a by-product: get_score's order, the spill path below), nnz is summed at the hand-off
    int32_t* out = results + h * M + t0;
    // LEAN: there is no list to emit -- the waves have gathered it slice by slice; the count is what was reserved
    int total = 0;
    int nspill = 0;                                                  // uniform
    uint32_t cs1 = 0u, cs2 = 0u;            // stand-alone retrieve: checksum of this thread's entries (host-buffer mode)
    int32_t* out2 = (AD == 0 && aa.rows2 != nullptr) ? aa.rows2 + h * M : nullptr;
    if constexpr (LEAN) {
        total = pre_total + nsp - (pre_total > lcap ? pre_total - lcap : 0);    // + the pool's finds
        nspill = nsp;
        MP_STAMP(stamp, 20);
    } else {
        total = emit_ordered<AD>(bmB, words, (int)T0, out, s_ids, AD > 0 ? aa.cap : 0, out2, s_tmp, cs1, cs2, stamp);
    }
    // AD: the member's ids stay in LDS; a list longer than the stage (cap ids) is read back from HBM
    const bool spill = !LEAN && AD > 0 && total > aa.cap;
    if ((LEAN ? lane : tid) == 0 && (AD == 0 || clog == 0)) {
        nnz[h] = total;
        if (AD == 0 && aa.rows2 != nullptr && aa.part_cnt != nullptr) aa.part_cnt[h] = total;
    }
    MP_STAMP(stamp, 21);
    if (AD == 0) {
        if (aa.rowsum != nullptr) {         // block sum of the two u32 checksums (wrap-around arithmetic: order-free)
            __syncthreads();                // (s_tmp is the scan's scratch)
            if (tid < 2) s_tmp[tid] = 0;
            __syncthreads();
            cs1 = wave_sum_u32(cs1);
            cs2 = wave_sum_u32(cs2);
            if (lane == 0) {
                atomicAdd(reinterpret_cast<unsigned int*>(&s_tmp[0]), cs1);
                atomicAdd(reinterpret_cast<unsigned int*>(&s_tmp[1]), cs2);
            }
            __syncthreads();
            if (tid < 2) aa.rowsum[h * 2 + tid] = (uint32_t)s_tmp[tid];
        }
        MP_STAMP_FLUSH(stamp);
        return;
    }

    // ------------------------------------------------------------ fused sparse attention of head h
    constexpr int ADD = AD > 0 ? AD : 64;
    uint16_t* out_h = aa.out + h * ADD;
    int wlen = 0;
    if (WIN && aa.win_kv != nullptr) {
        wlen = aa.win_len[h];
        wlen = wlen < 0 ? 0 : (wlen > aa.win_M ? (int)aa.win_M : wlen);
    }
    if (!LEAN && clog == 0 && total == 0 && wlen == 0) {
        attn_head_empty<ADD>(out_h, aa.mve, aa.BH, (int)h, aa.head_mz);
        MP_STAMP_FLUSH(stamp);
        return;
    }
    // s_ids complete.  Only a SPILLED list is read back from HBM by other waves (ids_hbm): only then must the stores have
    // drained (vmcnt) and be visible in L2; otherwise the barrier orders LDS alone -- the by-product rows' stores are
    // acknowledged ~0.5 us after they were issued, and this barrier sits on the path to the first row request
    // (LEAN, list in the stage: the counting barrier already ordered the appended ids in front of everything below)
    if constexpr (!LEAN) {
        if (spill) __syncthreads();
        else lds_barrier();
    }
    MP_STAMP(stamp, 33);
    // two instantiations of the sparse fold: the LDS path carries no global load ahead of its gathers
    const u32x4 qv = LEAN ? qv_own : *reinterpret_cast<const u32x4*>(s_qraw + (lane % (ADD / 8)) * 4);
    float* score_h = (!LEAN && aa.score) ? aa.score + h * M + t0 : nullptr;
    const uint16_t* kv_g = aa.kv + g * M * 2 * ADD;
    const float* kn_g = aa.kn + g * M;
    auto ids_lds = [&](int j) { return *reinterpret_cast<const u32x4*>(s_ids + j); };
    auto ids_hbm = [&](int j) {
        u32x4 v = {0u, 0u, 0u, 0u};
        for (int e = 0; e < 4; ++e)
            v[e] = ((uint32_t)(j + e) < tlen) ? (uint32_t)__builtin_nontemporal_load(out + j + e) : 0u;
        return v;
    };
    // lists that one round of 16-token steps covers (a member's ~190 ids at cfg 1) take those: twice the waves,
    // half the rows per wave (head_dim 128; at 64 a 16-token step would be two load instructions).  Measured and
    // rejected: 16-token steps for the last partial round of a long list (cfg 2: 641 ids = one round of 32-token
    // steps + 129 ids) -- 37.2 us per layer against 35.6 with HBM saturated.
    constexpr int SHORT = (ADD == 128) ? 16 : AH_SLICE;
    const bool short_list = SHORT < AH_SLICE && total <= SHORT * RT_WAVES && !spill;
    AhState st = LEAN ? st_own : ah_state_init(lane, ADD / 8);
    const uint16_t* kn_lds = pay ? s_kn : nullptr;
    if constexpr (LEAN) {
        // the waves folded their own finds above; what is left is the spill list (raw table words in this member's columns
        // of the result row; every wave waited for its stores before it drew its ticket), folded by this one wave
        if (nspill > 0)                                                  // uniform
            attn_head_fold_lean<ADD, (ADD == 128 ? 16 : AH_SLICE)>(st, kv_g, kn_g, qv, s_rn[1], nspill, M, ha.K, L, 0, 1, ids_hbm,
                                                                   idmask, idbits, pay, stamp);
    } else
    if (short_list)
        attn_head_fold<ADD, RT_WAVES, false, SHORT>(st, kv_g, kn_g, qv, s_rn[1], total, M, ha.K, L, 0, 1, ids_lds,
                                                    score_h, stamp, 0, kn_lds, (int)T0);
    else if (!spill)
        attn_head_fold<ADD, RT_WAVES, false, AH_SLICE>(st, kv_g, kn_g, qv, s_rn[1], total, M, ha.K, L, 0, 1,
                                                       ids_lds, score_h, stamp, 0, kn_lds, (int)T0);
    else
        attn_head_fold<ADD, RT_WAVES, false, AH_SLICE>(st, kv_g, kn_g, qv, s_rn[1], total, M, ha.K, L, 0, 1,
                                                       ids_hbm, score_h, stamp, 0, kn_lds, (int)T0);
    if (!LEAN && WIN && wlen > 0) {         // the static window: dense slices rank, rank + R, ... (k runs from
                                            // `wave` again, so the waves that got no sparse slice are served first)
        auto none = [](int) { return u32x4{0u, 0u, 0u, 0u}; };
        attn_head_fold<ADD, RT_WAVES, true, AH_SLICE>(st, aa.win_kv + g * aa.win_M * 2 * ADD, nullptr, qv, 1.f, wlen,
                                                      aa.win_M, 0, 0, rank, 1 << clog, none, nullptr, stamp);
    }
    if constexpr (LEAN) {                   // (this wave drew the last ticket above; its own slot again if it folded more)
        if (rare) {
            if (nspill > 0) attn_head_publish<ADD>(st, s_merge);
            attn_head_merge_read<ADD, RT_WAVES>(s_merge, m, Z, o0, o1);
        }
    } else {
    // no workgroup barrier: the wave that draws the last LDS ticket merges and goes on to the hand-off -- ONE wave from
    // here on, whichever it is (round 5; EXPERIMENTS.md R5-2)
    if (!attn_head_merge_ticket<ADD, RT_WAVES>(st, s_merge, s_tk, m, Z, o0, o1)) {
        MP_STAMP_FLUSH(stamp);
        return;
    }
    MP_STAMP_L(stamp, 40);
    }
    if (clog == 0) {
        attn_head_finalize<ADD>(m, Z, o0, o1, out_h, aa.mve, aa.BH, (int)h, aa.head_mz);
        MP_STAMP_L(stamp, 39);
        MP_STAMP_FLUSH_L(stamp);
        return;
    }
    cluster_handoff<ADD>(aa, h, rank, clog, m, Z, o0, o1, total, out_h, nnz, stamp);
}

// LSH::batch_retrieve (optionally with the query hash as its prologue)
template <int HASH, int CH>
__global__ __launch_bounds__(RT_THREADS, 4) void lsh_retrieve_kernel(
    const int32_t* __restrict__ bounds, const int32_t* __restrict__ table,
    const int32_t* __restrict__ query, int32_t* __restrict__ results, int32_t* __restrict__ nnz,
    int G, int L, int NB, int64_t M, int R, int words, int Lpad, const int* __restrict__ idbits_dev, HashArgs ha,
    int32_t* __restrict__ rows2, int32_t* __restrict__ nnz2, uint32_t* __restrict__ rowsum,
    unsigned long long* __restrict__ stamp) {
    AttnArgs aa = {};
    aa.rows2 = rows2;
    aa.part_cnt = nnz2;
    aa.rowsum = rowsum;
    // the layer's id width from the device word (written in stream order by a fill that widens the layer): a launch
    // argument would be frozen in a captured graph
    const int idbits = idbits_dev ? *idbits_dev : 0;
    lsh_head_body<HASH, CH, 0, false>(bounds, table, query, results, nnz, G, L, NB, M, R, 0, words, Lpad, idbits, ha, aa, stamp);
}

// the whole sparse layer of models/attnserver.py:264-300: hash -> retrieve -> attention
template <int CH, int AD, bool WIN, int HASH = 1, bool LEAN = false>
__global__ __launch_bounds__(RT_THREADS, 4) void lsh_decode_kernel(
    const int32_t* __restrict__ bounds, const int32_t* __restrict__ table,
    int32_t* __restrict__ results, int32_t* __restrict__ nnz,
    int G, int L, int NB, int64_t M, int R, int range_len, int words, int Lpad, int idbits, HashArgs ha, AttnArgs aa,
    unsigned long long* __restrict__ stamp) {
    lsh_head_body<HASH, CH, AD, WIN, LEAN>(bounds, table, nullptr, results, nnz, G, L, NB, M, R, range_len, words, Lpad,
                                           idbits, ha, aa, stamp);
}

// The decode kernel leaves member r's selected ids / logits at column r * range_len of the head's row; this
// moves the R segments of every row down into one contiguous list (get_score's `ind` order).  One workgroup
// per head; a segment only ever moves to lower addresses, chunk by chunk (read, barrier, write, barrier).
__global__ __launch_bounds__(1024) void lsh_compact_segments_kernel(uint32_t* __restrict__ rows,
                                                                   const int* __restrict__ part_cnt, int R,
                                                                   int range_len, int64_t M) {
    const int64_t h = blockIdx.x;
    uint32_t* row = rows + h * M;
    int off = part_cnt[h * CLUSTER_MAX] & 0xffffff;   // bits 24+ : the XCD a member reported (same-XCD hand-off)
    for (int r = 1; r < R; ++r) {
        const int cnt = part_cnt[h * CLUSTER_MAX + r] & 0xffffff;
        const int64_t src = (int64_t)r * range_len;
        if (off != src) {
            for (int base = 0; base < cnt; base += blockDim.x) {
                const int j = base + threadIdx.x;
                const uint32_t v = (j < cnt) ? row[src + j] : 0u;
                __syncthreads();
                if (j < cnt) row[off + j] = v;
                __syncthreads();
            }
        }
        off += cnt;
    }
}

// ---------------------------------------------------------------- LSH::get_mask (debug view)
// Recomputes min(count, 2) per token for the last query codes; byte counters in global memory,
// one workgroup per head: each workgroup zeroes its row, then walks the probed buckets table by table.
__global__ __launch_bounds__(256) void lsh_mask_kernel(
    const int32_t* __restrict__ bounds, const int32_t* __restrict__ table,
    const int32_t* __restrict__ query, int8_t* __restrict__ mask, int G, int L, int NB, int64_t M, int R,
    uint32_t idmask) {
    const int64_t h = blockIdx.x;
    const int64_t g = h / G;
    const int RS = R + 1;
    int8_t* row = mask + h * M;
    for (int64_t i = threadIdx.x; i < M; i += blockDim.x) row[i] = 0;
    __syncthreads();
    // tables are processed one after another by the whole block: within one table a token id
    // occurs at most once, so plain read-modify-write is race-free.
    for (int l = 0; l < L; ++l) {
        const int code = query[h * L + l];
        if (code < 0 || code >= NB) continue;
        const int32_t* rec = bounds + ((g * L + l) * NB + code) * RS;
        const int bx = rec[0], by = rec[R];
        const int32_t* src = table + (g * L + l) * M;
        for (int j = bx + threadIdx.x; j < by; j += blockDim.x) {
            const int32_t w = src[j];
            if (w == -1) continue;                                   // not an entry (as the retrieve's apply())
            const int64_t t = (int64_t)((uint32_t)w & idmask);
            if (t < M && row[t] < 2) row[t] = row[t] + 1;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- host launchers
// run `fn` once per (process, current device), serialised: function attributes are per device
struct DeviceOnce {
    std::mutex mu;
    bool done[64] = {};
    template <typename F>
    hipError_t run(F&& fn) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lock(mu);
        if (done[dev]) return hipSuccess;
        e = fn();
        if (e == hipSuccess) done[dev] = true;
        return e;
    }
};

// dynamic LDS a retrieve / decode workgroup may ask for: the CU's 160 KiB -- minus, in the -DMP_STAMPS=1 measurement
// build only, the static 512-byte stamp buffer (+ slack)
#if MP_STAMPS
constexpr size_t RT_LDS_DYN_MAX = 160u * 1024u - 1024u;
#else
constexpr size_t RT_LDS_DYN_MAX = 160u * 1024u;
#endif
size_t lsh_lds_limit() { return RT_LDS_DYN_MAX; }

// dynamic LDS of the retrieve body for a workgroup that owns `tokens` tokens
static size_t body_lds_bytes(int64_t tokens, int L) {
    const int Lpad = (L + 63) & ~63;
    // + fused-hash scratch: 128 words of query, 4 of norm, sign bits of up to 16*L planes (+ ballot slack)
    return (size_t)(2 * ((tokens + 31) / 32) + 2 * Lpad + RT_TAIL_CAP + 64 + 140 + (16 * L + 31) / 32 + 2 * RT_WAVES + 4) * 4;
}
size_t retrieve_lds_bytes(int64_t M, int L) { return body_lds_bytes(M, L); }
// the hash-only launch (mp_simhash_query on a handful of rows) keeps the sign bits of all K*L planes in LDS
bool lsh_hash_only_supported(int L) { return body_lds_bytes(0, L) <= RT_LDS_DYN_MAX; }

// tokens per range for a head split over R workgroups (multiple of 32: whole bitmap words)
int lsh_range_len(int64_t M, int R) { return (int)((((M + R - 1) / R) + 31) & ~(int64_t)31); }

hipError_t launch_lsh_subbounds(const int32_t* table, int32_t* bounds, int rows, int NB, int R, int64_t M,
                                int idbits, hipStream_t st) {
    if (R <= 1) return hipSuccess;
    hipLaunchKernelGGL(lsh_subbounds_kernel, dim3(rows), dim3(256), 0, st, table, bounds, NB, R, lsh_range_len(M, R), M,
                       idbits ? ((1u << idbits) - 1u) : 0xffffffffu);
    return hipGetLastError();
}

// words per direct slot (log2) for a head split over R token ranges: 128-byte slots (30 ids) while the mean piece
// M / (NB R) is above 5 ids, 64-byte slots (14 ids) above 2.5, 32-byte slots (6 ids) below -- a slot should hold a piece
// of ~2.5x the mean (SimHash buckets are wider than Poisson: p99 of a probed piece at cfg 1 is 2.3x its mean)
static int g_slot_log2 = 0;   // mp_debug_set_option("decode_slot_log2"): 3 / 4 / 5 forces 32- / 64- / 128-byte slots (A/B), 0 = by the mean piece
void set_slot_log2(int v) { g_slot_log2 = (v >= 3 && v <= 5) ? v : 0; }
int get_slot_log2() { return g_slot_log2; }
int lsh_slot_log2(int64_t M, int NB, int R) {
    if (g_slot_log2) return g_slot_log2;
    const double mean = (double)M / ((double)NB * (double)R);
    return mean > 5.0 ? 5 : (mean > 2.5 ? 4 : 3);
}

// swl: log2 of the words per slot THE HANDLE was allocated with (the option behind lsh_slot_log2 is process-wide and may
// have changed since: the buffer's size, the builder and the reader must agree -- ADVICE r05)
hipError_t launch_lsh_slots(const int32_t* table, const int32_t* bounds, int32_t* slots, int rows, int NB, int R,
                            int64_t M, int swl, hipStream_t st) {
    if (slots == nullptr) return hipSuccess;          // (R = 1 with slots: the decode_direct = 2 experiment)
    int gx = (NB * R + 31) / 32;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(lsh_slots_kernel, dim3(gx, rows), dim3(256), 0, st, table, bounds, slots, NB, R, M, swl);
    return hipGetLastError();
}

hipError_t launch_lsh_fill(const int16_t* codes, const int32_t* ids, int rows, int64_t n, int NB,
                           int64_t M, int R, int32_t* bounds, int32_t* table, int* err, hipStream_t st) {
    hipLaunchKernelGGL(lsh_fill_kernel, dim3(rows), dim3(256), 0, st, codes, ids, n, NB, M, R + 1, bounds,
                       table, err);
    return hipGetLastError();
}

hipError_t launch_lsh_unsort(const int16_t* codes, const int32_t* ids, int rows, int64_t n, int16_t* tok,
                             int* err, hipStream_t st) {
    int gx = (int)((n + 255) / 256);
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(lsh_unsort_kernel, dim3(gx, rows), dim3(256), 0, st, codes, ids, n, tok, err);
    return hipGetLastError();
}

// staged layout per NB (LDS <= 160 KB): waves per row, tokens per lane and tile
// tiles a row of n tokens is sorted in: tiles never straddle a range boundary unless it falls between two waves' slices
static int64_t build_tiles(int64_t n, int R, int range_len, int tpl, int nw) {
    const int64_t T = (int64_t)tpl * 64 * nw, slice = (int64_t)tpl * 64;
    if (R <= 1 || range_len <= 0 || range_len % slice == 0) return (n + T - 1) / T;
    int64_t tiles = 0;
    for (int r = 0; r < R; ++r) {
        int64_t len = n - (int64_t)r * range_len;
        len = len < 0 ? 0 : (len > range_len ? range_len : len);
        tiles += (len + T - 1) / T;
    }
    return tiles;
}

static bool build_staged_geometry(int NB, int64_t n, int R, int range_len, int& nw, int& tpl, size_t& lds, bool& pack) {
    pack = false;
    if (NB <= 1024)      { nw = 8;  tpl = 16; pack = true; }      // 72 KB: two workgroups per CU (16-bit tile counters)
    else if (NB <= 2048) {                                        // 72 - 80 KB: tiles of 4 096, 4 608 or 5 120 tokens -- the one that
        nw = 8; tpl = 8; pack = true;                             // cuts the row's token ranges into the fewest tiles (cfg 4:
        for (int cand = 9; cand <= 10; ++cand)                    // range_len 8 224 = 4 096 + 4 096 + 32, but 4 608 + 3 616)
            if (build_tiles(n, R, range_len, cand, nw) < build_tiles(n, R, range_len, tpl, nw)) tpl = cand;
    }
    else if (NB <= 4096) { nw = 4;  tpl = 32; }
    else if (NB <= 8192) { nw = 2;  tpl = 32; }
    else return false;
    const size_t T = (size_t)tpl * 64 * nw;
    // PACK: nw * NB 16-bit counters (the row histogram's 32-bit counters: nw / 2 sets in the same words)
    lds = ((pack ? (size_t)nw * NB / 2 : (size_t)nw * NB) + 2 * (size_t)NB + 32 + T) * 4 + T * 2;
    return lds <= 160u * 1024u;
}

// bounds entries 0 and R + the table; launch_lsh_subbounds fills entries 1 .. R-1 afterwards
// kn != nullptr: the packed build (only the staged kernels pack; *packed says whether they ran)
hipError_t launch_lsh_build(const int16_t* codes, int rows, int64_t n, int NB, int64_t M, int R,
                            int32_t* bounds, int32_t* table, int* err, const float* kn, int L, int idbits, int* bad,
                            bool* packed, bool* subbounds_done, bool exact_rank, hipStream_t st) {
    if (packed) *packed = false;
    if (subbounds_done) *subbounds_done = false;
    static DeviceOnce once;
    const hipError_t attr_err = once.run([] {
        const void* fns[] = {reinterpret_cast<const void*>(lsh_build_kernel<16, true>),
                             reinterpret_cast<const void*>(lsh_build_kernel<8, true>),
                             reinterpret_cast<const void*>(lsh_build_kernel<9, true>),
                             reinterpret_cast<const void*>(lsh_build_kernel<10, true>),
                             reinterpret_cast<const void*>(lsh_build_kernel<16, true, true>),
                             reinterpret_cast<const void*>(lsh_build_kernel<8, true, true>),
                             reinterpret_cast<const void*>(lsh_build_kernel<9, true, true>),
                             reinterpret_cast<const void*>(lsh_build_kernel<10, true, true>),
                             reinterpret_cast<const void*>(lsh_build_kernel<32>),
                             reinterpret_cast<const void*>(lsh_build_direct_kernel)};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    });
    if (attr_err != hipSuccess) return attr_err;
    if (n > INT32_MAX || NB > BUILD_LDS_COUNTERS) return hipErrorInvalidValue;
    if (n <= 0) {         // an empty request: every bucket empty (all entries of a record 0), no table word; the kernels
                          // prefetch codes[min(k, n - 1)] and must not be launched on nothing
        if (subbounds_done) *subbounds_done = true;
        return hipMemsetAsync(bounds, 0, (size_t)rows * NB * (R + 1) * sizeof(int32_t), st);
    }
    int nbits = 0;
    while ((1 << nbits) < NB) ++nbits;
    int nw, tpl;
    size_t lds;
    bool pack;
    const int RS = R + 1;
    if (build_staged_geometry(NB, n, R, R > 1 ? lsh_range_len(M, R) : 0, nw, tpl, lds, pack)) {
        const bool fast = pack && !exact_rank;                 // (the 32-codes-per-lane form keeps the exact ranking)
#if MP_STAMPS
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(d_build_stamp), &g_stamp, sizeof(g_stamp), 0, hipMemcpyHostToDevice, st);
#endif
#define MP_BUILD_CASE(TPL, PK, FA)                                                                         \
        if (tpl == TPL && pack == PK && fast == FA)                                                        \
            hipLaunchKernelGGL((lsh_build_kernel<TPL, PK, FA>), dim3(rows), dim3(64 * nw), lds, st, codes, \
                               (int)n, NB, nbits, M, RS, bounds, table, err, kn, L, idbits, bad,           \
                               R > 1 ? lsh_range_len(M, R) : 0);
        MP_BUILD_CASE(16, true, true) MP_BUILD_CASE(8, true, true) MP_BUILD_CASE(9, true, true) MP_BUILD_CASE(10, true, true)
        MP_BUILD_CASE(16, true, false) MP_BUILD_CASE(8, true, false) MP_BUILD_CASE(9, true, false) MP_BUILD_CASE(10, true, false)
        MP_BUILD_CASE(32, false, false)
#undef MP_BUILD_CASE
        if (packed) *packed = kn != nullptr;
        if (subbounds_done) *subbounds_done = tpl < 32;        // (R = 1 has none; R > 1: written by the kernel itself)
        return hipGetLastError();
    }
    nw = BUILD_LDS_COUNTERS / NB;
    if (nw > 16) nw = 16;
    lds = ((size_t)nw * NB + 64) * 4;
    hipLaunchKernelGGL(lsh_build_direct_kernel, dim3(rows), dim3(64 * nw), lds, st, codes, (int)n, NB,
                       nbits, M, RS, bounds, table, err);
    return hipGetLastError();
}

// ---- where do blocks land?  The cluster hand-off of lsh_decode_kernel may stay inside one XCD's L2
// only if blocks b and b + 8k run on one XCD.  That is measured, once per process and device: every
// block of two probe launches reports its XCC_ID (round robin over 8 distinct XCDs expected); the decode
// kernel re-checks every launch that the members of a cluster agree on theirs.
__global__ void xcc_probe_kernel(int* __restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
}

// false = not observed; else *map gets the XCC_ID of block residue b % 8 in bits 4b .. 4b+3
bool xcd_round_robin_map(uint32_t* map) {
    static std::mutex mu;
    static int states[64];        // per device: 0 unknown, 1 no, 2 yes
    static uint32_t maps[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    int& state = states[dev];
    if (state != 0) {
        if (map) *map = maps[dev];
        return state == 2;
    }
    state = 1;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(nullptr, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    int* d = nullptr;
    if (hipMalloc(&d, 2 * 96 * sizeof(int)) != hipSuccess) return false;
    int hbuf[2 * 96];
    bool ok = hipMemset(d, 0xff, 2 * 96 * sizeof(int)) == hipSuccess;
    const int grids[2] = {96, 40};
    for (int t = 0; t < 2 && ok; ++t) {
        hipLaunchKernelGGL(xcc_probe_kernel, dim3(grids[t]), dim3(64), 0, 0, d + t * 96);
        ok = ok && hipGetLastError() == hipSuccess;
    }
    ok = ok && hipMemcpy(hbuf, d, sizeof(hbuf), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return false;
    for (int t = 0; t < 2; ++t)
        for (int b = 0; b < grids[t]; ++b) {
            const int x = hbuf[t * 96 + b];
            if (x < 0 || x > 15 || x != hbuf[b % 8]) return false;      // same pattern in both launches
        }
    for (int a = 0; a < 8; ++a)
        for (int b = a + 1; b < 8; ++b)
            if (hbuf[a] == hbuf[b]) return false;                        // 8 distinct XCDs
    uint32_t m = 0;
    for (int a = 0; a < 8; ++a) m |= (uint32_t)hbuf[a] << (4 * a);
    maps[dev] = m;
    state = 2;
    if (map) *map = m;
    return true;
}

bool xcd_round_robin_verified() { return xcd_round_robin_map(nullptr); }

constexpr int DECODE_ID_CAP = 4096;   // ids of the fused kernel's LDS stage per workgroup (128 slices)

static size_t decode_lds_bytes(int64_t tokens, int L, int D, bool lean = false) {
    size_t b = body_lds_bytes(tokens, L) + ((size_t)DECODE_ID_CAP + attn_head_lds_floats(RT_WAVES, D) + 16 + 64) * 4;
    if (lean) {      // per-token counters (8 bits while L <= 255, else 16) in place of the two bitmaps
        const size_t words = (size_t)((tokens + 31) / 32);
        b += ((words << (L <= 255 ? 3 : 4)) - 2 * words) * 4;
    }
    return b;
}

static hipError_t retrieve_attr_set() {
    const void* fns[] = {reinterpret_cast<const void*>(lsh_retrieve_kernel<0, 16>),
                         reinterpret_cast<const void*>(lsh_retrieve_kernel<1, 16>),
                         reinterpret_cast<const void*>(lsh_retrieve_kernel<1, 8>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, false, 2>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<8, 64, false, 2>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, false>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<8, 64, false>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, true>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<8, 64, true>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, false, 3>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<8, 64, false, 3>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, true, 3>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<8, 64, true, 3>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, false, 5>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, true, 5>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, false, 1, true>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<8, 64, false, 1, true>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, true, 1, true>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<8, 64, true, 1, true>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, false, 3, true>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<8, 64, false, 3, true>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<16, 128, true, 3, true>),
                         reinterpret_cast<const void*>(lsh_decode_kernel<8, 64, true, 3, true>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RT_LDS_DYN_MAX);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static hipError_t retrieve_attr_once() {
    static DeviceOnce once;
    return once.run(retrieve_attr_set);
}

hipError_t launch_lsh_retrieve(const int32_t* bounds, const int32_t* table, const int32_t* query,
                               int32_t* results, int32_t* nnz, int BH, int G, int L, int NB,
                               int64_t M, int R, const int* idbits, int32_t* rows2, int32_t* nnz2, uint32_t* rowsum,
                               hipStream_t st) {
    const int words = (int)((M + 31) / 32);
    const int Lpad = (L + 63) & ~63;
    hipError_t e = retrieve_attr_once();
    if (e != hipSuccess) return e;
    HashArgs ha = {};
    hipLaunchKernelGGL((lsh_retrieve_kernel<0, 16>), dim3(BH), dim3(RT_THREADS), retrieve_lds_bytes(M, L),
                       st, bounds, table, query, results, nnz, G, L, NB, M, R, words, Lpad, idbits, ha, rows2, nnz2, rowsum, g_stamp);
    return hipGetLastError();
}

// q-hash fused into the retrieve (the two-launch form of mp_decode_sparse_layer): codes and ||q||
// are by-products
hipError_t launch_lsh_hash_retrieve(const int32_t* bounds, const int32_t* table, const uint16_t* q,
                                    const uint16_t* Wk, const float* wnorm, int D, int K, int KLpad,
                                    int32_t* codes_out, float* qnorm_out, int32_t* results,
                                    int32_t* nnz, int BH, int G, int L, int NB, int64_t M, int R, const int* idbits,
                                    hipStream_t st) {
    const int words = (int)((M + 31) / 32);
    const int Lpad = (L + 63) & ~63;
    hipError_t e = retrieve_attr_once();
    if (e != hipSuccess) return e;
    HashArgs ha = {q, Wk, wnorm, codes_out, qnorm_out, D, K, KLpad, g_exact_norm};
    if (D >= 128)
        hipLaunchKernelGGL((lsh_retrieve_kernel<1, 16>), dim3(BH), dim3(RT_THREADS), retrieve_lds_bytes(M, L),
                           st, bounds, table, (const int32_t*)nullptr, results, nnz, G, L, NB, M, R, words,
                           Lpad, idbits, ha, (int32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr, g_stamp);
    else
        hipLaunchKernelGGL((lsh_retrieve_kernel<1, 8>), dim3(BH), dim3(RT_THREADS), retrieve_lds_bytes(M, L),
                           st, bounds, table, (const int32_t*)nullptr, results, nnz, G, L, NB, M, R, words,
                           Lpad, idbits, ha, (int32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr, g_stamp);
    return hipGetLastError();
}

// the query SimHash alone, one workgroup per row (few rows; simhash.hip's MFMA kernel takes the bulk case)
hipError_t launch_lsh_hash_only(const uint16_t* q, const uint16_t* Wk, const float* wnorm, int D, int K, int KLpad,
                                int32_t* codes_out, float* qnorm_out, int rows, int L, hipStream_t st) {
    const int Lpad = (L + 63) & ~63;
    hipError_t e = retrieve_attr_once();
    if (e != hipSuccess) return e;
    HashArgs ha = {q, Wk, wnorm, codes_out, qnorm_out, D, K, KLpad, g_exact_norm};
    const size_t lds = body_lds_bytes(0, L);
    if (D >= 128)
        hipLaunchKernelGGL((lsh_retrieve_kernel<1, 16>), dim3(rows), dim3(RT_THREADS), lds, st, (const int32_t*)nullptr,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, 1, L,
                           1 << K, (int64_t)0, 1, 0, Lpad, (const int*)nullptr, ha, (int32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr, g_stamp);
    else
        hipLaunchKernelGGL((lsh_retrieve_kernel<1, 8>), dim3(rows), dim3(RT_THREADS), lds, st, (const int32_t*)nullptr,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, 1, L,
                           1 << K, (int64_t)0, 1, 0, Lpad, (const int*)nullptr, ha, (int32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr, g_stamp);
    return hipGetLastError();
}

// the whole sparse layer in ONE launch (mp_decode_sparse_layer): hash + retrieve + attention.
// D = 64 or 128; R = workgroups per head = token ranges of the tables (1, 2, 4, 8, 16 or 32).
bool lsh_decode_supported(int64_t M, int L, int D, int R) {
    return (D == 64 || D == 128) && decode_lds_bytes(lsh_range_len(M, R), L, D) <= RT_LDS_DYN_MAX;
}

hipError_t launch_lsh_decode(const int32_t* bounds, const int32_t* table, const uint16_t* q,
                             const uint16_t* Wk, const float* wnorm, int D, int K, int KLpad,
                             int32_t* codes_out, float* qnorm_out, int32_t* results, int32_t* nnz,
                             const uint16_t* kv, const float* kn, float* part_o, float2* part_ml, int* part_cnt,
                             int* head_cnt, uint16_t* out, float* mve, float2* head_mz, const int32_t* slots, int slot_log2,
                             float* score, int* err, int maxs, int R, bool same_xcd, const uint16_t* win_kv,
                             const int32_t* win_len, int64_t win_M, int BH, int G, int L, int NB, int64_t M,
                             bool codes_given, unsigned long long* xw, unsigned int* xseq, int xwords, int xmode,
                             int idbits, const int* idbits_dev, const int* pay_bad, const unsigned int* att_ver,
                             const unsigned int* kn_ver, bool lean, bool* lean_ran, const uint16_t* Wt, int quad_mode, hipStream_t st) {
    const int range_len = lsh_range_len(M, R);
    const int words = range_len / 32;
    const int Lpad = (L + 63) & ~63;
    hipError_t e = retrieve_attr_once();
    if (e != hipSuccess) return e;
    int clog = 0;
    while ((1 << clog) < R) ++clog;
    if ((1 << clog) != R || R > CLUSTER_MAX) return hipErrorInvalidValue;
    const int BHp = clog > 0 ? (BH + 7) & ~7 : BH;
    const bool sx = same_xcd && clog > 0 && xcd_round_robin_verified();
    HashArgs ha = {q, Wk, wnorm, codes_out, qnorm_out, D, K, KLpad, g_exact_norm};
    // planes split over the cluster + exchange of the sign bits through the XCD's L2: only where the members of a
    // cluster share an XCD (sx) and every unit of 64 planes finds a wave (K*L <= 1024 R)
    const bool split_hash = xmode != 0 && sx && xw != nullptr && xseq != nullptr && !codes_given &&
                            ((K * L + 63) / 64) <= (RT_WAVES << clog) && 2 * ((K * L + 63) / 64) <= xwords;
    AttnArgs aa = {kv, kn, part_o, part_ml, part_cnt, head_cnt, out, mve, head_mz, slots, slot_log2, score, err, BH, BHp, maxs,
                   DECODE_ID_CAP, clog, sx ? 1 : 0, split_hash ? xw : nullptr, split_hash ? xseq : nullptr, xwords, xmode, nullptr, idbits_dev, nullptr, nullptr,
                   win_kv, win_len, win_M};
    const dim3 grid((unsigned)BHp << clog);
    size_t lds = decode_lds_bytes(range_len, L, D);
    // key norms from the table entries' payload: 2 bytes of LDS per token of a member's range, where they fit
    const bool want_pay = pay_bad != nullptr && att_ver != nullptr && kn_ver != nullptr && idbits != 0 &&
                          lds + (size_t)range_len * 2 + 16 <= RT_LDS_DYN_MAX;
    // the LEAN form keeps a counter per token where the other keeps two bits: only where that fits next to everything
    // else (the payload's norms first: they are worth more than the lean form)
    // Where a head has ONE workgroup (clog == 0: B*H >= CUs / 2, cfg 2 / 3) the lean form is not used: there the gather is
    // bound by HBM, and row requests that start while other workgroups still count delay THEIR table loads -- measured
    // +0.7 ... +1.3 us per launch at cfg 2 / 3 (EXPERIMENTS.md R6-1); the flag then only has no effect.
    if (lean) {      // (a selected token's norm travels in its list entry there: no norm array)
        if (clog > 0 && decode_lds_bytes(range_len, L, D, true) <= RT_LDS_DYN_MAX && !codes_given) lds = decode_lds_bytes(range_len, L, D, true);
        else lean = false;
    }
    if (want_pay) {
        aa.pay_bad = pay_bad;
        aa.att_ver = att_ver;
        aa.kn_ver = kn_ver;
        if (!lean) lds += (size_t)range_len * 2 + 16;
    }
    if (lean_ran) *lean_ran = lean;
    if (codes_given) {   // A/B: the codes and ||q|| come from simhash_query_kernel (plain decode only)
        if (win_kv != nullptr) return hipErrorInvalidValue;
        if (D == 128)
            hipLaunchKernelGGL((lsh_decode_kernel<16, 128, false, 2>), grid, dim3(RT_THREADS), lds, st, bounds, table,
                               results, nnz, G, L, NB, M, R, range_len, words, Lpad, idbits, ha, aa, g_stamp);
        else
            hipLaunchKernelGGL((lsh_decode_kernel<8, 64, false, 2>), grid, dim3(RT_THREADS), lds, st, bounds, table,
                               results, nnz, G, L, NB, M, R, range_len, words, Lpad, idbits, ha, aa, g_stamp);
        return hipGetLastError();
    }
    // HASH = 5, the quad MFMA hash: one workgroup per head, the heads in blocks of 32 (four per XCD residue), head_dim 128,
    // the placement block b -> XCD b % 8 observed, an exchange word per 32 planes, at most 64 tiles (16 waves x 4 members)
    const bool sxq = same_xcd && xcd_round_robin_verified();
    if (quad_mode >= 1 && clog == 0 && D == 128 && BH % 32 == 0 && sxq && Wt != nullptr && xw != nullptr && xseq != nullptr &&
        !codes_given && 2 * ((K * L + 63) / 64) <= 64 && 2 * ((K * L + 63) / 64) <= xwords &&
        2 * ((K * L + 63) / 64) * 32 <= KLpad && words % 2 == 0) {     // (the rows' LDS copies are read 16 bytes at a time)
        if (lean_ran) *lean_ran = false;                            // (the quad hash exists in the by-products-on form only)
        lds = decode_lds_bytes(range_len, L, D) + (want_pay ? (size_t)range_len * 2 + 16 : 0);
        aa.xw = xw;
        aa.xseq = xseq;
        aa.xmode = quad_mode == 2 ? 2 : 1;                          // 2: nobody publishes (test: every head falls back to hashing alone)
        aa.rows2 = reinterpret_cast<int32_t*>(const_cast<uint16_t*>(Wt));
        if (win_kv != nullptr)
            hipLaunchKernelGGL((lsh_decode_kernel<16, 128, true, 5>), grid, dim3(RT_THREADS), lds, st, bounds, table, results, nnz,
                               G, L, NB, M, R, range_len, words, Lpad, idbits, ha, aa, g_stamp);
        else
            hipLaunchKernelGGL((lsh_decode_kernel<16, 128, false, 5>), grid, dim3(RT_THREADS), lds, st, bounds, table, results, nnz,
                               G, L, NB, M, R, range_len, words, Lpad, idbits, ha, aa, g_stamp);
        return hipGetLastError();
    }
#define MP_DECODE_CASE(DD, CHH, WW)                                                                            \
    if (D == DD && (win_kv != nullptr) == WW) {                                                                \
        if (split_hash && lean)                                                                                \
            hipLaunchKernelGGL((lsh_decode_kernel<CHH, DD, WW, 3, true>), grid, dim3(RT_THREADS), lds, st, bounds,   \
                               table, results, nnz, G, L, NB, M, R, range_len, words, Lpad, idbits, ha, aa, g_stamp);  \
        else if (split_hash)                                                                                   \
            hipLaunchKernelGGL((lsh_decode_kernel<CHH, DD, WW, 3>), grid, dim3(RT_THREADS), lds, st, bounds,   \
                               table, results, nnz, G, L, NB, M, R, range_len, words, Lpad, idbits, ha, aa, g_stamp);  \
        else if (lean)                                                                                         \
            hipLaunchKernelGGL((lsh_decode_kernel<CHH, DD, WW, 1, true>), grid, dim3(RT_THREADS), lds, st, bounds,   \
                               table, results, nnz, G, L, NB, M, R, range_len, words, Lpad, idbits, ha, aa, g_stamp);  \
        else                                                                                                   \
            hipLaunchKernelGGL((lsh_decode_kernel<CHH, DD, WW>), grid, dim3(RT_THREADS), lds, st, bounds,      \
                               table, results, nnz, G, L, NB, M, R, range_len, words, Lpad, idbits, ha, aa, g_stamp);  \
        return hipGetLastError();                                                                              \
    }
    MP_DECODE_CASE(128, 16, false)
    MP_DECODE_CASE(128, 16, true)
    MP_DECODE_CASE(64, 8, false)
    MP_DECODE_CASE(64, 8, true)
#undef MP_DECODE_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_lsh_compact(uint32_t* rows, const int* part_cnt, int BH, int R, int64_t M, hipStream_t st) {
    if (R <= 1) return hipSuccess;
    hipLaunchKernelGGL(lsh_compact_segments_kernel, dim3(BH), dim3(1024), 0, st, rows, part_cnt, R,
                       lsh_range_len(M, R), M);
    return hipGetLastError();
}

hipError_t launch_lsh_mask(const int32_t* bounds, const int32_t* table, const int32_t* query,
                           int8_t* mask, int BH, int G, int L, int NB, int64_t M, int R, int idbits, hipStream_t st) {
    hipLaunchKernelGGL(lsh_mask_kernel, dim3(BH), dim3(256), 0, st, bounds, table, query, mask, G,
                       L, NB, M, R, idbits ? ((1u << idbits) - 1u) : 0xffffffffu);
    return hipGetLastError();
}

}  // namespace mp
