// attention.hip -- gathered sparse KV attention with importance-sampling correction (gfx950).
//
// Replaces library/sparse_attention/sparse_attention.cc: qk_kernel{,_bf16_impl} (:38-103),
// transform_kernel (:164-184), softmax_kernel (:186-240), wv_kernel* (:321-518), the
// orchestration variants (:748-986, 1039-1211) and full_attention (:988-1037).
//
// HBM layout (per layer): K and V rows INTERLEAVED per token, bf16 [B*Hkv][M][2][D]
// (512 contiguous bytes per token at D = 128: the K row and the V row of a selected token
// share one DRAM page); key norms f32 [B*Hkv][M].
//
// Design (memory-bound gather, ~524 B per selected token, no reuse):
//   * split-KV over the index list: a SLICE is 64 consecutive entries of one head's `ind`, owned
//     by ONE wave; grid = (GX, B*H) workgroups of 4 independent waves stride over the slices of
//     their head (nnz is read on device: no host readback, no cross-workgroup prefix);
//   * a wave owns 64 tokens; one global_load_dwordx4 fetches 4 whole rows (16 lanes x 16 B per
//     256-byte row), so a wave keeps 16 K-row loads + 16 V-row loads (32 KB) in flight;
//   * q.K: 4 v_dot2c_f32_bf16 per row chunk, then a 15-step reduce-scatter over the 16 lanes
//     of a row group leaves exactly ONE token's score per lane, so the transcendental
//     importance transform (acos, two integer powers, log, exp) runs once per token per lane;
//   * softmax is per wave (max / sum by shuffles), P.V accumulates 8 f32 per lane and each slice
//     publishes one (m, l, o[D]) partial (520 B) write-through; the wave that draws the last
//     arrival ticket of a head merges that head's partials in the same launch and writes bf16
//     out + base-2 LSE (no second kernel, no grid-wide wait).
#include "common.h"
#include "attn_head.h"

namespace mp {

extern unsigned long long* g_stamp;   // simhash.hip

constexpr int AT_THREADS = 256;
constexpr int AT_WAVES = AT_THREADS / 64;

// D: head_dim (64 or 128); DENSE: ids are 0..nnz-1 and no importance transform (full_attention);
// QBF16: query is bf16 (the reference's __AVX512BF16__ family) else f32.
//
// grid = (GX, BH), block = 256: every WAVE is an independent worker that owns 64-entry slices
// (x*4 + w) + k*GX*4, k = 0, 1, ... of head h's index list -- no LDS, no barrier.  The first
// slice's index load is issued speculatively, in parallel with the nnz[h] load, so the
// dependent chain of a wave is  {nnz, ind} -> K/V/|k| gathers -> math -> one 520-byte partial.
template <int D, bool DENSE, bool QBF16>
__global__ __launch_bounds__(AT_THREADS) void attn_sparse_kernel(
    const uint16_t* __restrict__ kv,     // [B*Hkv][M][2][D]
    const float* __restrict__ kn,        // [B*Hkv][M]
    const void* __restrict__ query,      // [BH][D] bf16 or f32
    const float* __restrict__ qnorm,     // [BH]
    const int32_t* __restrict__ ind,     // [BH][M]
    const int32_t* __restrict__ nnz,     // [BH]
    float* __restrict__ part_o,          // [BH][MAXS][D]   slice partials (write-through)
    float2* __restrict__ part_ml,        // [BH][MAXS]      (max logit, sum exp)
    int* __restrict__ head_cnt,          // [BH] arrival tickets, zero between launches
    uint16_t* __restrict__ out,          // [BH][D] bf16
    float* __restrict__ mve,             // [2][BH] max*log2e, base-2 LSE
    float2* __restrict__ head_mz,        // [BH] (max logit, Z) for get_score
    float* __restrict__ score,           // [BH][M] transformed logits z_j (nullable)
    int BH, int G, int64_t M, int maxs, int K, int L, unsigned long long* __restrict__ stamp) {
    constexpr int LPR = D / 8;           // lanes per row (16 B each); a wave load fetches 64 / LPR rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane / LPR, c = lane % LPR;
    // 1-D grid, head fastest: block b -> (x = b / BH, h = b % BH).  Blocks land on XCD b % 8, so
    // the blocks that own the first slices of every head (the only ones with work when the list is
    // short) are dispatched first and spread evenly over the 8 XCDs; a (x, h) 2-D grid would pin
    // each x to one XCD residue class.
    const int h = blockIdx.x % BH;
    const int bx = blockIdx.x / BH, gx = gridDim.x / BH;
    const int64_t g = h / G;
    const int stride = gx * AT_WAVES;                   // slices per pass over this head
    int s = bx * AT_WAVES + wave;
    MP_STAMP(stamp, 32);

    // Token slot of (gather step u, row group r) is r*LPR + u, so after the reduce-scatter lane l
    // owns token jb + l, and the LPR ids a row group gathers are LPR CONSECUTIVE ints of `ind`
    // (read as 16-byte loads, no cross-lane traffic before the gathers are issued).
    // The first slice's ids are loaded speculatively, alongside nnz[h].
    const int32_t* ind_h = ind + (int64_t)h * M;
    u32x4 idv[LPR / 4];
    auto load_ids = [&](int64_t jb0) {
#pragma unroll
        for (int v = 0; v < LPR / 4; ++v) {
            const int64_t j0 = jb0 + r * LPR + v * 4;
            if (j0 + 3 < M) {
                idv[v] = *reinterpret_cast<const u32x4*>(ind_h + j0);
            } else {   // last, partial quad of the row: stay inside [0, M)
#pragma unroll
                for (int e = 0; e < 4; ++e) idv[v][e] = (j0 + e < M) ? (uint32_t)ind_h[j0 + e] : 0u;
            }
        }
    };
    if (!DENSE) load_ids((int64_t)s * 64);
    int nz = nnz[h];
    if ((int64_t)nz > M) nz = (int)M;
    if (nz <= 0) {                                      // empty head: out = 0, LSE = -inf (a-10)
        if (bx == 0 && wave == 0) {
            for (int d = lane; d < D; d += 64) out[(int64_t)h * D + d] = 0;
            if (lane == 0) {
                mve[h] = -INFINITY;
                mve[BH + h] = -INFINITY;
                head_mz[h] = make_float2(-INFINITY, 0.f);
            }
        }
        return;
    }
    if ((int64_t)s * 64 >= nz) return;                  // wave-uniform exit
    const int ns = (nz + 63) >> 6;                      // slices (= partials, = tickets) of this head

    // query fragment of this lane: elements c*8 .. c*8+7
    uint32_t qpk[4];
    float qf[8];
    if (QBF16) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(
            reinterpret_cast<const uint16_t*>(query) + (int64_t)h * D + c * 8);
        qpk[0] = t[0]; qpk[1] = t[1]; qpk[2] = t[2]; qpk[3] = t[3];
    } else {
        const float* qp = reinterpret_cast<const float*>(query) + (int64_t)h * D + c * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] = qp[i];
    }
    const float qn_h = DENSE ? 1.f : qnorm[h];
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    const uint16_t* kvg = kv + g * M * 2 * D + c * 8;
    MP_STAMP(stamp, 33);

    bool first = true;
    for (; (int64_t)s * 64 < nz; s += stride) {
        const int jb = s * 64;
        if (!DENSE && !first) load_ids(jb);
        first = false;
        // my token (the one whose score lands on this lane after the reduce-scatter)
        const int j_my = jb + lane;
        const bool valid_my = j_my < nz;

        // ---- issue all row gathers: step u fetches the rows of token slots r*LPR + u
        u32x4 kreg[LPR], vreg[LPR];
        int id_my = 0;
#pragma unroll
        for (int u = 0; u < LPR; ++u) {
            int id_u = DENSE ? (jb + r * LPR + u) : (int)idv[u / 4][u % 4];
            const bool valid_u = (jb + r * LPR + u) < nz;
            if (id_u < 0 || (int64_t)id_u >= M) id_u = 0;      // never fault on a bad index
            if (u == c) id_my = id_u;
            const u32x4 zero = {0u, 0u, 0u, 0u};
            kreg[u] = zero;
            vreg[u] = zero;
            if (valid_u) {
                const uint16_t* row = kvg + (int64_t)id_u * 2 * D;
                // rows are read once and never reused: non-temporal loads (no L2/MALL allocation)
                kreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row));
                vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + D));
            }
        }
        float kn_my = 1.f;
        if (!DENSE && valid_my) kn_my = kn[g * M + id_my];
        MP_STAMP(stamp, 34);

        // ---- q . K partials, then reduce-scatter over the LPR lanes of a row group
        float part[LPR];
#pragma unroll
        for (int u = 0; u < LPR; ++u) {
            float a = 0.f;
            if (QBF16) {
                const u32x4 qv = {qpk[0], qpk[1], qpk[2], qpk[3]};
                dot8_bf16_chain(a, kreg[u], qv);
                dot_settle(a);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a = fmaf(bf16_lo(kreg[u][i]), qf[2 * i], a);
                    a = fmaf(bf16_hi(kreg[u][i]), qf[2 * i + 1], a);
                }
            }
            part[u] = a;
        }
#pragma unroll
        for (int st = LPR / 2; st >= 1; st >>= 1) {
            const bool upper = (c & st) != 0;
#pragma unroll
            for (int u = 0; u < st; ++u) {
                const float send = upper ? part[u] : part[u + st];
                const float keep = upper ? part[u + st] : part[u];
                part[u] = keep + __shfl_xor(send, st);
            }
        }
        const float sc = part[0];   // = q . K[id_my]
        MP_STAMP(stamp, 35);

        // ---- importance-sampling transform (transform_kernel, sparse_attention.cc:164-184)
        float z = -INFINITY;
        if (valid_my) {
            if (DENSE) {
                z = sc * inv_sqrt_d;
            } else {
                z = importance_logit(sc, qn_h * kn_my, inv_sqrt_d, K, L);
            }
            if (score != nullptr) score[(int64_t)h * M + j_my] = z;
        }
        MP_STAMP(stamp, 36);

        // ---- softmax of the slice (max / sum across the wave)
        const float m_w = wave_max(z);
        const float p_my = valid_my ? __expf(z - m_w) : 0.f;    // slice non-empty => m_w finite
        const float l_w = wave_sum(p_my);

        // ---- P . V
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int u = 0; u < LPR; ++u) {
            const float pu = __shfl(p_my, r * LPR + u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] = fmaf(pu, bf16_lo(vreg[u][i]), acc[2 * i]);
                acc[2 * i + 1] = fmaf(pu, bf16_hi(vreg[u][i]), acc[2 * i + 1]);
            }
        }
        // reduce-scatter over the RPL row groups: lane (r, c) ends with VPL = 8/RPL outputs,
        // elements d0 .. d0+VPL-1 of o[D]
        int doff = 0;
        rs_step_h<8, 32>(acc, lane, doff);
        rs_step_h<4, 16>(acc, lane, doff);
        if (LPR == 8) rs_step_h<2, 8>(acc, lane, doff);
        constexpr int VPL = (LPR == 16) ? 2 : 1;
        const int d0 = c * 8 + doff;
        MP_STAMP(stamp, 37);

        // ---- publish the slice partial WRITE-THROUGH (sc1: agent-scope relaxed atomic stores),
        // drain, then take an arrival ticket; the wave that draws the last ticket of the head
        // merges all partials (read back with sc1 loads, which bypass this CU's L1).  Hand-off
        // recipe: cdna_hip_programming.md G16 (R1 with a counter).
        const int64_t pidx = (int64_t)h * maxs + s;
        if (VPL == 2) {
            const unsigned long long pk = (unsigned long long)__float_as_uint(acc[0]) |
                                          ((unsigned long long)__float_as_uint(acc[1]) << 32);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(part_o + pidx * D + d0), pk,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(reinterpret_cast<unsigned int*>(part_o + pidx * D + d0),
                               __float_as_uint(acc[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            const unsigned long long pk = (unsigned long long)__float_as_uint(m_w) |
                                          ((unsigned long long)__float_as_uint(l_w) << 32);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(part_ml + pidx), pk,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing lane of this wave drained
        int ticket = 0;
        if (lane == 0)
            ticket = __hip_atomic_fetch_add(head_cnt + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        MP_STAMP(stamp, 38);
        if (ticket != ns - 1) continue;

        // ---- last arriver: merge the ns partials of head h
        if (lane == 0) __hip_atomic_store(head_cnt + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t pre = (int64_t)h * maxs;
        float m = -INFINITY;
        for (int t = lane; t < ns; t += 64) {
            const unsigned long long pk = __hip_atomic_load(
                reinterpret_cast<unsigned long long*>(part_ml + pre + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m = fmaxf(m, __uint_as_float((uint32_t)pk));
        }
        m = wave_max(m);
        float Z = 0.f, o0 = 0.f, o1 = 0.f;
        for (int t0 = 0; t0 < ns; t0 += 8) {
            unsigned long long ml[8], ov[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ml[u] = 0ull;
                ov[u] = 0ull;
                if (t0 + u < ns) {
                    ml[u] = __hip_atomic_load(reinterpret_cast<unsigned long long*>(part_ml + pre + t0 + u),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (VPL == 2)
                        ov[u] = __hip_atomic_load(
                            reinterpret_cast<unsigned long long*>(part_o + (pre + t0 + u) * D + d0),
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        ov[u] = __hip_atomic_load(
                            reinterpret_cast<unsigned int*>(part_o + (pre + t0 + u) * D + d0),
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u < ns) {
                    const float e = __expf(__uint_as_float((uint32_t)ml[u]) - m);
                    Z = fmaf(e, __uint_as_float((uint32_t)(ml[u] >> 32)), Z);
                    o0 = fmaf(e, __uint_as_float((uint32_t)ov[u]), o0);
                    o1 = fmaf(e, __uint_as_float((uint32_t)(ov[u] >> 32)), o1);
                }
            }
        }
        // out = sum_s e^{m_s-m} o_s / Z as bf16 (RNE); max_value_expsum[0] = m*log2e,
        // [1] = log2 Z + m*log2e (softmax_kernel, sparse_attention.cc:238-239)
        if (VPL == 2) {
            const uint32_t pk = (uint32_t)f32_to_bf16_rne(o0 / Z) | ((uint32_t)f32_to_bf16_rne(o1 / Z) << 16);
            *reinterpret_cast<uint32_t*>(out + (int64_t)h * D + d0) = pk;
        } else {
            out[(int64_t)h * D + d0] = f32_to_bf16_rne(o0 / Z);
        }
        if (lane == 0) {
            const float mv = m * 1.4426950408889634f;
            mve[h] = mv;
            mve[BH + h] = log2f(Z) + mv;
            head_mz[h] = make_float2(m, Z);
        }
        MP_STAMP(stamp, 39);
    }
}

// One workgroup per head: the tail of the fused decode kernel (attn_head.h) with the ids read from
// HBM.  Used by attention_wrapper when every CU has at least one head of its own (B*H >= CUs / 2):
// no partials, no tickets, one barrier -- 25 us instead of 31 us at cfg 2 (B*H = 256).
template <int D>
__global__ __launch_bounds__(1024) void attn_head_kernel(
    const uint16_t* __restrict__ kv, const float* __restrict__ kn, const uint16_t* __restrict__ query,
    const float* __restrict__ qnorm, const int32_t* __restrict__ ind, const int32_t* __restrict__ nnz,
    uint16_t* __restrict__ out, float* __restrict__ mve, float2* __restrict__ head_mz,
    float* __restrict__ score, int BH, int G, int64_t M, int K, int L,
    unsigned long long* __restrict__ stamp) {
    __shared__ float s_merge[attn_head_lds_floats(16, D)];
    const int h = blockIdx.x;
    const int64_t g = h / G;
    MP_STAMP(stamp, 32);
    int nz = nnz[h];
    if ((int64_t)nz > M) nz = (int)M;
    if (nz <= 0) {
        attn_head_empty<D>(out + (int64_t)h * D, mve, BH, h, head_mz);
        return;
    }
    const int32_t* ind_h = ind + (int64_t)h * M;
    auto ids = [&](int j0) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (j0 + 3 < M) v = *reinterpret_cast<const u32x4*>(ind_h + j0);
        else
            for (int e = 0; e < 4; ++e) v[e] = (j0 + e < M) ? (uint32_t)ind_h[j0 + e] : 0u;
        return v;
    };
    MP_STAMP(stamp, 33);
    float m, Z, o0, o1;
    const u32x4 qv = *reinterpret_cast<const u32x4*>(query + (int64_t)h * D + (threadIdx.x % (D / 8)) * 8);
    attn_head_tail<D, 16, AH_SLICE>(kv + g * M * 2 * D, kn + g * M, qv, qnorm[h], nz, M, K, L, 0, 1,
                      ids, s_merge, score ? score + (int64_t)h * M : nullptr, stamp, m, Z, o0, o1);
    if (threadIdx.x < 64) attn_head_finalize<D>(m, Z, o0, o1, out + (int64_t)h * D, mve, BH, h, head_mz);
    MP_STAMP(stamp, 39);
}

// ---------------------------------------------------------------- full_attention, K/V shared by the group
// SparseAttentionServer::full_attention (sparse_attention.cc:988-1037) loops over kv heads and scores the G
// query heads of a group against every loaded K row (qk_kernel_full :106-160, wv_kernel_dim128_full :386-451).
// Same here: a workgroup owns a range of 32-row slices of ONE kv head and folds every row into the softmax
// states of all G query heads, so K and V cross HBM once per group instead of once per query head (the
// per-head DENSE instantiation of attn_sparse_kernel reads them G times: 4x the bytes at Llama's G = 4 --
// it is what the two dense layers of the f-3 decode step were made of).
// grid = groups * split workgroups of 4 waves, group fastest (XCD spread); slice s of the group's rows belongs
// to workgroup x = s / 4 % split, wave s % 4.  Every workgroup publishes one partial per head (m, Z, o[D]; empty:
// m = -inf) and takes that head's arrival ticket; the last of the `split` arrivals merges (agent scope, as
// attn_sparse_kernel).
template <int D, int G, bool QBF16>
__global__ __launch_bounds__(256) void attn_dense_kernel(
    const uint16_t* __restrict__ kv,     // [groups][M][2][D]
    const void* __restrict__ query,      // [BH][D] bf16 or f32
    const int32_t* __restrict__ nnz,     // [BH] rows [0, nnz[h]) take part for head h
    float* __restrict__ part_o,          // [BH][maxs][D]
    float2* __restrict__ part_ml,        // [BH][maxs]
    int* __restrict__ head_cnt,          // [BH] arrival tickets, zero between launches
    uint16_t* __restrict__ out, float* __restrict__ mve, float2* __restrict__ head_mz,
    float* __restrict__ score,           // [BH][M] logits (nullable)
    int BH, int groups, int64_t M, int maxs, int split) {
    constexpr int NW = 4;
    constexpr int LPR = D / 8, UPS = LPR / 2, VPL = D / 64;   // 32-row slices, as attn_head_fold<.., AH_SLICE>
    constexpr int QW = QBF16 ? D / 2 : D;                      // 32-bit words of one query row in LDS
    __shared__ __attribute__((aligned(16))) uint32_t s_q[G * QW];
    __shared__ float s_merge[G * NW * (D + 2)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane / LPR, c = lane % LPR;
    const int gidx = blockIdx.x % groups, x = blockIdx.x / groups;
    const int h0 = gidx * G;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    // the group's queries -> LDS (every lane needs elements c*8 .. c*8+7 of all G rows)
    for (int i = tid; i < G * QW; i += 256)
        s_q[i] = reinterpret_cast<const uint32_t*>(query)[(int64_t)h0 * QW + i];
    int nz[G], nzmax = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        int z = nnz[h0 + g];
        z = z < 0 ? 0 : ((int64_t)z > M ? (int)M : z);
        nz[g] = z;
        nzmax = z > nzmax ? z : nzmax;
    }
    __syncthreads();
    AhState st[G];
#pragma unroll
    for (int g = 0; g < G; ++g) st[g] = ah_state_init(lane, LPR);
    const uint16_t* kvc = kv + (int64_t)gidx * M * 2 * D + c * 8;

    for (int64_t s = (int64_t)x * NW + wave; s * AH_SLICE < nzmax; s += (int64_t)split * NW) {
        const int jb = (int)s * AH_SLICE;
        const int j_my = jb + r * UPS + (c >> 1);
        u32x4 kreg[UPS], vreg[UPS];
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
            int row = jb + r * UPS + u;
            row = row < nzmax ? row : jb;                    // loads stay unconditional
            const uint16_t* p = kvc + (int64_t)row * 2 * D;
            kreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
            vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + D));
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (jb >= nz[g]) continue;                      // wave-uniform: this head's list ended before the slice
            float part[UPS];
            if (QBF16) {
                const u32x4 qv = *reinterpret_cast<const u32x4*>(s_q + g * QW + c * 4);
#pragma unroll
                for (int u = 0; u < UPS; ++u) {
                    float a = 0.f;
                    dot8_bf16_chain(a, kreg[u], qv);
                    dot_settle(a);
                    part[u] = a;
                }
            } else {
                const float* qf = reinterpret_cast<const float*>(s_q) + g * QW + c * 8;
                const float4 qa = *reinterpret_cast<const float4*>(qf), qb = *reinterpret_cast<const float4*>(qf + 4);
#pragma unroll
                for (int u = 0; u < UPS; ++u) {
                    float a = 0.f;
                    a = fmaf(bf16_lo(kreg[u][0]), qa.x, a);
                    a = fmaf(bf16_hi(kreg[u][0]), qa.y, a);
                    a = fmaf(bf16_lo(kreg[u][1]), qa.z, a);
                    a = fmaf(bf16_hi(kreg[u][1]), qa.w, a);
                    a = fmaf(bf16_lo(kreg[u][2]), qb.x, a);
                    a = fmaf(bf16_hi(kreg[u][2]), qb.y, a);
                    a = fmaf(bf16_lo(kreg[u][3]), qb.z, a);
                    a = fmaf(bf16_hi(kreg[u][3]), qb.w, a);
                    part[u] = a;
                }
            }
#pragma unroll
            for (int stp = LPR / 2; stp >= 2; stp >>= 1) {
                const bool upper = (c & stp) != 0;
#pragma unroll
                for (int u = 0; u < UPS / 2; ++u) {
                    if (u < stp / 2) {
                        const float send = upper ? part[u] : part[u + stp / 2];
                        const float keep = upper ? part[u + stp / 2] : part[u];
                        part[u] = keep + __shfl_xor(send, stp);
                    }
                }
            }
            const float sc = part[0] + __shfl_xor(part[0], 1);   // q_g . K[j_my] on lanes c and c^1
            const bool valid = j_my < nz[g];
            const float z = valid ? sc * inv_sqrt_d : -INFINITY;
            if (valid && score != nullptr && (c & 1) == 0) score[(int64_t)(h0 + g) * M + j_my] = z;
            const float m_w = wave_max(z);                       // finite: jb < nz[g]
            const float p_my = valid ? __expf(z - m_w) : 0.f;
            const float l_w = wave_sum((c & 1) ? 0.f : p_my);
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
            for (int u = 0; u < UPS; ++u) {
                const float pu = __shfl(p_my, r * LPR + 2 * u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] = fmaf(pu, bf16_lo(vreg[u][i]), acc[2 * i]);
                    acc[2 * i + 1] = fmaf(pu, bf16_hi(vreg[u][i]), acc[2 * i + 1]);
                }
            }
            int doff = 0;
            rs_step_h<8, 32>(acc, lane, doff);
            rs_step_h<4, 16>(acc, lane, doff);
            if (LPR == 8) rs_step_h<2, 8>(acc, lane, doff);
            st[g].d0 = c * 8 + doff;
            const float m_new = fmaxf(st[g].m, m_w);
            const float a = __expf(st[g].m - m_new), b = __expf(m_w - m_new);
            st[g].l = fmaf(a, st[g].l, b * l_w);
            st[g].o0 = fmaf(a, st[g].o0, b * acc[0]);
            if (LPR == 16) st[g].o1 = fmaf(a, st[g].o1, b * acc[1]);
            st[g].m = m_new;
        }
    }
    // the four waves' states of every head meet in LDS; wave g % 4 takes head g, g + 4 from there
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float* mine = s_merge + (g * NW + wave) * (D + 2);
        mine[st[g].d0] = st[g].o0;
        if (LPR == 16) mine[st[g].d0 + 1] = st[g].o1;
        if (lane == 0) {
            mine[D] = st[g].m;
            mine[D + 1] = st[g].l;
        }
    }
    __syncthreads();
    for (int g = wave; g < G; g += NW) {
        const int h = h0 + g;
        float mw[NW], lw[NW], oa[NW], ob[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float* sw = s_merge + (g * NW + w) * (D + 2);
            mw[w] = sw[D];
            lw[w] = sw[D + 1];
            oa[w] = sw[lane * VPL];
            ob[w] = VPL == 2 ? sw[lane * 2 + 1] : 0.f;
        }
        float m = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3]));
        float Z = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (mw[w] != -INFINITY) {                    // a wave without a slice never wrote its o[]
                const float e = __expf(mw[w] - m);
                Z = fmaf(e, lw[w], Z);
                o0 = fmaf(e, oa[w], o0);
                o1 = fmaf(e, ob[w], o1);
            }
        }
        if (split == 1) {
            attn_head_finalize<D>(m, Z, o0, o1, out + (int64_t)h * D, mve, BH, h, head_mz);
            continue;
        }
        // publish this workgroup's partial of head h write-through, drain, ticket; the last arrival merges
        const int64_t pidx = (int64_t)h * maxs + x;
        __hip_atomic_store(reinterpret_cast<unsigned int*>(part_o + pidx * D + lane * VPL), __float_as_uint(o0),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (VPL == 2)
            __hip_atomic_store(reinterpret_cast<unsigned int*>(part_o + pidx * D + lane * 2 + 1), __float_as_uint(o1),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(part_ml + pidx),
                               (unsigned long long)__float_as_uint(m) | ((unsigned long long)__float_as_uint(Z) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(head_cnt + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != split - 1) continue;
        if (lane == 0) __hip_atomic_store(head_cnt + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t pre = (int64_t)h * maxs;
        float mm = -INFINITY;
        for (int t = lane; t < split; t += 64) {
            const unsigned long long pk = __hip_atomic_load(reinterpret_cast<unsigned long long*>(part_ml + pre + t),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mm = fmaxf(mm, __uint_as_float((uint32_t)pk));
        }
        mm = wave_max(mm);
        float ZZ = 0.f, q0 = 0.f, q1 = 0.f;
        for (int t0 = 0; t0 < split; t0 += 8) {
            unsigned long long ml[8];
            uint32_t va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ml[u] = 0ull;
                va[u] = vb[u] = 0u;
                if (t0 + u < split) {
                    ml[u] = __hip_atomic_load(reinterpret_cast<unsigned long long*>(part_ml + pre + t0 + u),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    va[u] = __hip_atomic_load(reinterpret_cast<unsigned int*>(part_o + (pre + t0 + u) * D + lane * VPL),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (VPL == 2)
                        vb[u] = __hip_atomic_load(
                            reinterpret_cast<unsigned int*>(part_o + (pre + t0 + u) * D + lane * 2 + 1),
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float mu = __uint_as_float((uint32_t)ml[u]);
                if (t0 + u < split && mu != -INFINITY) {
                    const float e = __expf(mu - mm);
                    ZZ = fmaf(e, __uint_as_float((uint32_t)(ml[u] >> 32)), ZZ);
                    q0 = fmaf(e, __uint_as_float(va[u]), q0);
                    q1 = fmaf(e, __uint_as_float(vb[u]), q1);
                }
            }
        }
        attn_head_finalize<D>(mm, ZZ, q0, q1, out + (int64_t)h * D, mve, BH, h, head_mz);
    }
}

// get_score: logits z_j -> probabilities exp(z_j - m)/Z in place (first nnz entries per head).
__global__ void attn_normalize_kernel(float* __restrict__ score, const int32_t* __restrict__ nnz,
                                      const float2* __restrict__ head_mz, int64_t M) {
    const int h = blockIdx.y;
    int nz = nnz[h];
    if ((int64_t)nz > M) nz = (int)M;
    const float2 mz = head_mz[h];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nz; j += gridDim.x * blockDim.x) {
        float* p = score + (int64_t)h * M + j;
        *p = __expf(*p - mz.x) / mz.y;
    }
}

// SparseAttentionServer::fill (.cc:601-627): copy k, v [Hkv][n][D] and kn [Hkv][n] of one request
// into the interleaved layout.  One thread per 16 bytes.
__global__ void attn_fill_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                 const float* __restrict__ knorm, int Hkv, int64_t n, int D, int64_t M,
                                 uint16_t* __restrict__ kv, float* __restrict__ kn) {
    const int cpr = D / 8;  // 16-byte chunks per row
    const int64_t total = (int64_t)Hkv * n * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        const int64_t t = (i / cpr) % n;
        const int64_t hk = i / (cpr * n);
        const u32x4 a = *reinterpret_cast<const u32x4*>(k + (hk * n + t) * D + ch * 8);
        const u32x4 b = *reinterpret_cast<const u32x4*>(v + (hk * n + t) * D + ch * 8);
        uint16_t* dst = kv + (hk * M + t) * 2 * D + ch * 8;
        *reinterpret_cast<u32x4*>(dst) = a;
        *reinterpret_cast<u32x4*>(dst + D) = b;
        if (ch == 0) kn[hk * M + t] = knorm[hk * n + t];
    }
}

// ---------------------------------------------------------------- prefill: centre + store the offloaded keys
// models/attnserver.py:136-148 for one request: avg_k = offload_key.mean(dim=1) (bf16), offload_key - avg_k
// (bf16), kn = that.norm(p=2, dim=-1).float(), then SparseAttentionServer::fill -- as two passes over the model's
// KV cache (token-major bf16 [seq_len][Hkv*D]) instead of torch's mean / sub / norm / transpose().contiguous()
// kernels and their intermediates.  Definitions (the oracle's, oracle.centre_keys): the column sum is taken in
// f64 (exact: bf16 addends, < 2^29 of them), the mean rounded f64 -> f32 -> bf16 RNE; k - avg is the f32
// difference of two bf16 numbers rounded RNE to bf16; the norm is sqrt of the exact f64 sum of squares, rounded
// f64 -> f32 -> bf16.  torch sums in f32 in an order of its own, so it can land one bf16 ulp off where the exact
// value sits within ~1e-6 of a rounding boundary (tests/golden/fill_centre.npz lists those).
//
// pass 1a: per-block column sums of tokens [t0, t0 + n): grid = nblk, thread handles 4 adjacent columns
__global__ __launch_bounds__(256) void key_colsum_kernel(const uint16_t* __restrict__ key_cache, int64_t t0,
                                                         int64_t n, int cols, double* __restrict__ partial) {
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t a = (int64_t)blockIdx.x * per, b = (a + per < n) ? a + per : n;
    for (int cb = threadIdx.x * 4; cb < cols; cb += blockDim.x * 4) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        // eight rows' loads in flight, added in row order (round 6: as a loop of one 8-byte load per iteration the block's ~96
        // rows were 96 dependent round trips: 70 us per layer at cfg 1 for 200 MB; EXPERIMENTS.md R6-8)
        for (int64_t t = a; t < b; t += 8) {
            uint2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t tt = t + u < b ? t + u : b - 1;
                v[u] = *reinterpret_cast<const uint2*>(key_cache + (t0 + tt) * cols + cb);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t + u < b) {
                    acc[0] += (double)bf16_lo(v[u].x);
                    acc[1] += (double)bf16_hi(v[u].x);
                    acc[2] += (double)bf16_lo(v[u].y);
                    acc[3] += (double)bf16_hi(v[u].y);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) partial[(int64_t)blockIdx.x * cols + cb + i] = acc[i];
    }
}
// pass 1b: avg[c] = bf16(float(sum_b partial[b][c] / n)); one wave per column (f64 adds of exact values: order-free)
__global__ __launch_bounds__(256) void key_colmean_kernel(const double* __restrict__ partial, int nblk, int cols,
                                                          int64_t n, uint16_t* __restrict__ avg) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= cols) return;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += partial[(int64_t)b * cols + c];
    s = wave_sum(s);
    if (lane == 0) avg[c] = f32_to_bf16_rne((float)(s / (double)n));
}
// pass 2: row t of kv head g of the store <- (bf16(k - avg) | v), kn <- norm; one thread per 16 bytes of a row
__global__ __launch_bounds__(256) void key_centre_fill_kernel(
    const uint16_t* __restrict__ key_cache, const uint16_t* __restrict__ value_cache, int64_t t0, int64_t n,
    int Hkv, int D, const uint16_t* __restrict__ avg, int64_t M, uint16_t* __restrict__ kv, float* __restrict__ kn) {
    const int cpr = D / 8, cols = Hkv * D;
    const int64_t total = n * Hkv * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ((total + 63) & ~(int64_t)63);
         i += (int64_t)gridDim.x * blockDim.x) {
        const bool live = i < total;
        const int64_t ii = live ? i : total - 1;
        const int ch = (int)(ii % cpr);
        const int g = (int)((ii / cpr) % Hkv);
        const int64_t t = ii / ((int64_t)cpr * Hkv);
        const int64_t src = (t0 + t) * cols + g * D + ch * 8;
        u32x4 a = *reinterpret_cast<const u32x4*>(key_cache + src);
        const u32x4 b = *reinterpret_cast<const u32x4*>(value_cache + src);
        const u32x4 m = *reinterpret_cast<const u32x4*>(avg + g * D + ch * 8);
        double ss = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = f32_to_bf16_rne(bf16_lo(a[j]) - bf16_lo(m[j]));
            const uint32_t hi = f32_to_bf16_rne(bf16_hi(a[j]) - bf16_hi(m[j]));
            a[j] = lo | (hi << 16);
            const double dl = (double)bf16_lo(a[j]), dh = (double)bf16_hi(a[j]);
            ss += dl * dl + dh * dh;
        }
        for (int sft = 1; sft < cpr; sft <<= 1) ss += __shfl_xor(ss, sft);    // the cpr chunks of a row: adjacent lanes
        if (live) {
            uint16_t* dst = kv + ((int64_t)g * M + t) * 2 * D + ch * 8;
            *reinterpret_cast<u32x4*>(dst) = a;
            *reinterpret_cast<u32x4*>(dst + D) = b;
            if (ch == 0) kn[(int64_t)g * M + t] = bf16_bits_to_f32(f32_to_bf16_rne((float)sqrt(ss)));
        }
    }
}

// flashinfer.append_paged_kv_cache as used at models/attnserver.py:281-290: write this step's
// (k, v) of every request at row pos[b] of its kv heads.  k, v: bf16 [B][Hkv][D].
__global__ void attn_append_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                   const int32_t* __restrict__ pos, int pos_delta,
                                   const uint16_t* __restrict__ centre,   // [B*Hkv][D] bf16 or nullptr
                                   int Hkv, int D, int64_t M, uint16_t* __restrict__ kv,
                                   float* __restrict__ kn, unsigned int* __restrict__ kn_ver, int* __restrict__ err) {
    const int b = blockIdx.x / Hkv;
    const int64_t unit = blockIdx.x;                 // b*Hkv + kv head
    const int p = pos[b] + pos_delta;
    if (p < 0 || (int64_t)p >= M) {                  // window full: report, never write out of bounds
        if (threadIdx.x == 0) atomicOr(err, 2);
        return;
    }
    // a norm of this KV group changes: its version becomes "unknown" in stream order (also in a replayed graph), which
    // no LSH table can carry -- a decode kernel whose table words hold this group's norms reads them per token again
    // until the store is refilled (capi.hip: KN_VERSION_UNKNOWN)
    if (threadIdx.x == 0 && kn_ver != nullptr) kn_ver[unit] = 0xffffffffu;
    const int cpr = D / 8;
    double ss = 0.0;
    if ((int)threadIdx.x < cpr) {
        u32x4 a = *reinterpret_cast<const u32x4*>(k + unit * D + threadIdx.x * 8);
        const u32x4 c = *reinterpret_cast<const u32x4*>(v + unit * D + threadIdx.x * 8);
        if (centre != nullptr) {                     // k - avg_k as torch computes it on bf16 tensors:
                                                     // f32 subtraction, RNE back to bf16 (attnserver.py:267)
            const u32x4 m = *reinterpret_cast<const u32x4*>(centre + unit * D + threadIdx.x * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = f32_to_bf16_rne(bf16_lo(a[j]) - bf16_lo(m[j]));
                const uint32_t hi = f32_to_bf16_rne(bf16_hi(a[j]) - bf16_hi(m[j]));
                a[j] = lo | (hi << 16);
            }
        }
        uint16_t* dst = kv + (unit * M + p) * 2 * D + threadIdx.x * 8;
        *reinterpret_cast<u32x4*>(dst) = a;
        *reinterpret_cast<u32x4*>(dst + D) = c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double lo = (double)bf16_lo(a[j]), hi = (double)bf16_hi(a[j]);
            ss += lo * lo + hi * hi;
        }
    }
    ss = wave_sum(ss);                               // cpr <= 16 lanes of wave 0 contribute
    if (threadIdx.x == 0) kn[unit * M + p] = bf16_bits_to_f32(f32_to_bf16_rne((float)sqrt(ss)));
}

// flashinfer.merge_state as used at models/attnserver.py:308 (base-2 LSEs).
__global__ void merge_state_kernel(const uint16_t* __restrict__ va, const float* __restrict__ sa,
                                   const uint16_t* __restrict__ vb, const float* __restrict__ sb,
                                   int R, int D, uint16_t* __restrict__ v, float* __restrict__ s) {
    const int rr = blockIdx.x;
    const float a = sa[rr], b = sb[rr];
    const float mx = fmaxf(a, b);
    float wa = 0.f, wb = 0.f, lse = -INFINITY;
    if (mx > -INFINITY) {
        const float ea = exp2f(a - mx), eb = exp2f(b - mx);
        wa = ea / (ea + eb);
        wb = eb / (ea + eb);
        lse = mx + log2f(ea + eb);
    }
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float x = wa * bf16_bits_to_f32(va[(int64_t)rr * D + d]) +
                        wb * bf16_bits_to_f32(vb[(int64_t)rr * D + d]);
        v[(int64_t)rr * D + d] = f32_to_bf16_rne(x);
    }
    if (s != nullptr && threadIdx.x == 0) s[rr] = lse;
}

// ---------------------------------------------------------------- host-buffer mode: ragged rows <-> one packed buffer
// The reference's callers hand over CPU tensors results / ind [BH][M] of which only the first nnz[h] entries of
// row h mean anything (models/attnserver.py:299-300).  Instead of one hipMemcpy per head, the rows cross PCIe as
// ONE packed buffer: offs = exclusive prefix of min(nnz, M) (one workgroup), rows <-> packed by position.
__global__ __launch_bounds__(1024) void ragged_offsets_kernel(const int32_t* __restrict__ nnz, int BH, int64_t M,
                                                              int32_t* __restrict__ offs) {   // [BH + 1]
    __shared__ int s_tmp[32];
    int carry = 0;
    for (int base = 0; base < BH; base += 1024) {
        const int i = base + threadIdx.x;
        int v = 0;
        if (i < BH) {
            v = nnz[i];
            v = v < 0 ? 0 : ((int64_t)v > M ? (int)M : v);
        }
        int total;
        __syncthreads();
        const int ex = block_excl_scan(v, s_tmp, total) + carry;
        if (i < BH) offs[i] = ex;
        carry += total;
    }
    if (threadIdx.x == 0) offs[BH] = carry;
}
template <bool PACK>
__global__ void ragged_copy_kernel(int32_t* __restrict__ rows, int32_t* __restrict__ packed,
                                   const int32_t* __restrict__ offs, int64_t M) {
    const int h = blockIdx.y;
    const int o = offs[h], n = offs[h + 1] - o;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        if (PACK) packed[o + j] = rows[(int64_t)h * M + j];
        else rows[(int64_t)h * M + j] = packed[o + j];
    }
}

// host-buffer mode: a few KB between the handle's pinned block (as the device sees it) and HBM, 16 bytes per lane
__global__ __launch_bounds__(1024) void relay_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
}
// host-buffer mode of the attention entry, one launch: blocks (x, h < BH) bring the first nnz[h] entries of the caller's
// index row h into HBM (coalesced 256-byte wave reads over PCIe: read in 16-byte pieces by the attention kernel's row
// groups the same ids cost ~20 us at cfg 1), blocks (x, BH) relay the small arguments (q | qn | nnz).
__global__ __launch_bounds__(256) void host_rows_kernel(const int32_t* __restrict__ ind_host, const int32_t* __restrict__ nnz_host,
                                                        int32_t* __restrict__ rows, int64_t M, int BH,
                                                        const uint4* __restrict__ small_src, uint4* __restrict__ small_dst, int n16) {
    const int h = blockIdx.y;
    if (h == BH) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) small_dst[i] = small_src[i];
        return;
    }
    int64_t n = nnz_host[h];
    n = n < 0 ? 0 : (n > M ? M : n);
    const int32_t* src = ind_host + (int64_t)h * M;
    int32_t* dst = rows + (int64_t)h * M;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) dst[j] = src[j];
}
hipError_t launch_host_rows(const int32_t* ind_host, const int32_t* nnz_host, int32_t* rows, int64_t M, int BH,
                            const void* small_src, void* small_dst, size_t small_bytes, int gx, hipStream_t st) {
    hipLaunchKernelGGL(host_rows_kernel, dim3(gx, BH + 1), dim3(256), 0, st, ind_host, nnz_host, rows, M, BH,
                       reinterpret_cast<const uint4*>(small_src), reinterpret_cast<uint4*>(small_dst),
                       (int)((small_bytes + 15) / 16));
    return hipGetLastError();
}
// host-buffer mode: "everything before me on this stream is done" as a word in pinned memory (capi.hip: host_wait)
__global__ void host_flag_kernel(volatile unsigned int* flag, unsigned int value) {
    *flag = value;
    __threadfence_system();
}
hipError_t launch_host_flag(unsigned int* flag_dev, unsigned int value, hipStream_t st) {
    hipLaunchKernelGGL(host_flag_kernel, dim3(1), dim3(1), 0, st, flag_dev, value);
    return hipGetLastError();
}

// ||q|| of every query row (f32 of a bf16 / f32 row: what the reference's caller computes on the host,
// models/attnserver.py:300) for the attention launch the host-mode retrieve issues on its own (capi.hip: speculation): one
// wave per row, written to HBM for the kernel and to pinned memory for the comparison with the norms the caller hands over
// (q_copy: the rows are also copied to HBM -- the attention launch then reads nothing of the caller's any more)
__global__ __launch_bounds__(256) void row_norm_kernel(const void* __restrict__ q, int bf16, int rows, int D,
                                                       float* __restrict__ out_dev, float* __restrict__ out_host,
                                                       void* __restrict__ q_copy) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        float x;
        if (bf16) {
            const uint16_t b = reinterpret_cast<const uint16_t*>(q)[(size_t)row * D + d];
            reinterpret_cast<uint16_t*>(q_copy)[(size_t)row * D + d] = b;
            x = bf16_bits_to_f32(b);
        } else {
            x = reinterpret_cast<const float*>(q)[(size_t)row * D + d];
            reinterpret_cast<float*>(q_copy)[(size_t)row * D + d] = x;
        }
        s = fmaf(x, x, s);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float n = sqrtf(s);
        out_dev[row] = n;
        out_host[row] = n;
    }
}
hipError_t launch_row_norm(const void* q, bool bf16, int rows, int D, float* out_dev, float* out_host, void* q_copy,
                           hipStream_t st) {
    hipLaunchKernelGGL(row_norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, q, bf16 ? 1 : 0, rows, D, out_dev, out_host, q_copy);
    return hipGetLastError();
}

hipError_t launch_relay(const void* src, void* dst, size_t bytes, hipStream_t st) {
    hipLaunchKernelGGL(relay_kernel, dim3(1), dim3(1024), 0, st, reinterpret_cast<const uint4*>(src),
                       reinterpret_cast<uint4*>(dst), (int)((bytes + 15) / 16));
    return hipGetLastError();
}

hipError_t launch_ragged_offsets(const int32_t* nnz, int BH, int64_t M, int32_t* offs, hipStream_t st) {
    hipLaunchKernelGGL(ragged_offsets_kernel, dim3(1), dim3(1024), 0, st, nnz, BH, M, offs);
    return hipGetLastError();
}
hipError_t launch_ragged_copy(bool pack, int32_t* rows, int32_t* packed, const int32_t* offs, int BH, int64_t M,
                              hipStream_t st) {
    int gx = (int)((M + 255) / 256);
    if (gx > 16) gx = 16;
    if (pack) hipLaunchKernelGGL(ragged_copy_kernel<true>, dim3(gx, BH), dim3(256), 0, st, rows, packed, offs, M);
    else hipLaunchKernelGGL(ragged_copy_kernel<false>, dim3(gx, BH), dim3(256), 0, st, rows, packed, offs, M);
    return hipGetLastError();
}

// ---------------------------------------------------------------- host launchers
// partial records per head: one per 64-entry slice of the index list
int attn_slices_per_head(int64_t M) { return (int)((M + 63) / 64); }

int attn_supported_head_dim(int D) { return D == 64 || D == 128; }

template <int D, bool DENSE, bool QBF16>
static hipError_t launch_sparse_t(const uint16_t* kv, const float* kn, const void* q, const float* qn,
                                  const int32_t* ind, const int32_t* nnz, float* part_o,
                                  float2* part_ml, int* head_cnt, uint16_t* out, float* mve,
                                  float2* head_mz, float* score, int BH, int G, int64_t M, int K,
                                  int L, int grid, hipStream_t st) {
    const int maxs = attn_slices_per_head(M);
    hipLaunchKernelGGL((attn_sparse_kernel<D, DENSE, QBF16>), dim3(grid * BH), dim3(AT_THREADS), 0, st,
                       kv, kn, q, qn, ind, nnz, part_o, part_ml, head_cnt, out, mve, head_mz, score, BH,
                       G, M, maxs, K, L, g_stamp);
    return hipGetLastError();
}

hipError_t launch_attn_sparse(int D, bool dense, bool qbf16, const uint16_t* kv, const float* kn,
                              const void* q, const float* qn, const int32_t* ind, const int32_t* nnz,
                              float* part_o, float2* part_ml, int* head_cnt, uint16_t* out, float* mve,
                              float2* head_mz, float* score, int BH, int G, int64_t M, int K, int L,
                              int grid, bool head_kernel, hipStream_t st) {
    if (!dense && qbf16 && head_kernel) {
        if (D == 128)
            hipLaunchKernelGGL(attn_head_kernel<128>, dim3(BH), dim3(1024), 0, st, kv, kn, (const uint16_t*)q, qn,
                               ind, nnz, out, mve, head_mz, score, BH, G, M, K, L, g_stamp);
        else
            hipLaunchKernelGGL(attn_head_kernel<64>, dim3(BH), dim3(1024), 0, st, kv, kn, (const uint16_t*)q, qn,
                               ind, nnz, out, mve, head_mz, score, BH, G, M, K, L, g_stamp);
        return hipGetLastError();
    }
#define MP_AT_CASE(DD, DE, QB)                                                                     \
    if (D == DD && dense == DE && qbf16 == QB)                                                     \
        return launch_sparse_t<DD, DE, QB>(kv, kn, q, qn, ind, nnz, part_o, part_ml, head_cnt, out, \
                                           mve, head_mz, score, BH, G, M, K, L, grid, st);
    MP_AT_CASE(128, false, true)
    MP_AT_CASE(128, false, false)
    MP_AT_CASE(128, true, true)
    MP_AT_CASE(128, true, false)
    MP_AT_CASE(64, false, true)
    MP_AT_CASE(64, false, false)
    MP_AT_CASE(64, true, true)
    MP_AT_CASE(64, true, false)
#undef MP_AT_CASE
    return hipErrorInvalidValue;
}

// full_attention with the group's K/V read once (G = 1, 2, 4 or 8); false = shape not covered, use the per-head kernel
bool launch_attn_dense(int D, int G, bool qbf16, const uint16_t* kv, const void* q, const int32_t* nnz, float* part_o,
                       float2* part_ml, int* head_cnt, uint16_t* out, float* mve, float2* head_mz, float* score,
                       int BH, int64_t M, int cus, hipStream_t st, hipError_t* err) {
    if (!(D == 64 || D == 128) || !(G == 1 || G == 2 || G == 4 || G == 8) || BH % G != 0) return false;
    const int groups = BH / G;
    const int maxs = attn_slices_per_head(M);
    // ~4 workgroups of 4 waves per CU; never more workgroups per group than 4-slice bundles of its rows
    int split = (cus * 4 + groups - 1) / groups;
    const int64_t bundles = (M + 4 * AH_SLICE - 1) / (4 * AH_SLICE);
    if (split > bundles) split = (int)bundles;
    if (split > maxs) split = maxs;
    if (split < 1) split = 1;
#define MP_DENSE_CASE(DD, GG, QB)                                                                            \
    if (D == DD && G == GG && qbf16 == QB) {                                                                 \
        hipLaunchKernelGGL((attn_dense_kernel<DD, GG, QB>), dim3(groups * split), dim3(256), 0, st, kv, q, nnz, \
                           part_o, part_ml, head_cnt, out, mve, head_mz, score, BH, groups, M, maxs, split);  \
        *err = hipGetLastError();                                                                             \
        return true;                                                                                          \
    }
#define MP_DENSE_G(DD, QB) MP_DENSE_CASE(DD, 1, QB) MP_DENSE_CASE(DD, 2, QB) MP_DENSE_CASE(DD, 4, QB) MP_DENSE_CASE(DD, 8, QB)
    MP_DENSE_G(128, true) MP_DENSE_G(128, false) MP_DENSE_G(64, true) MP_DENSE_G(64, false)
#undef MP_DENSE_G
#undef MP_DENSE_CASE
    return false;
}

hipError_t launch_attn_normalize(float* score, const int32_t* nnz, const float2* head_mz, int BH,
                                 int64_t M, hipStream_t st) {
    int gx = (int)((M + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(attn_normalize_kernel, dim3(gx, BH), dim3(256), 0, st, score, nnz, head_mz, M);
    return hipGetLastError();
}

hipError_t launch_attn_fill(const uint16_t* k, const uint16_t* v, const float* knorm, int Hkv,
                            int64_t n, int D, int64_t M, uint16_t* kv, float* kn, hipStream_t st) {
    const int64_t total = (int64_t)Hkv * n * (D / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(attn_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, k, v, knorm, Hkv,
                       n, D, M, kv, kn);
    return hipGetLastError();
}

// avg bf16 [Hkv*D] (device), partial f64 [nblk][Hkv*D] scratch
hipError_t launch_key_centre_fill(const uint16_t* key_cache, const uint16_t* value_cache, int64_t t0, int64_t n,
                                  int Hkv, int D, int64_t M, double* partial, int nblk, uint16_t* avg, uint16_t* kv,
                                  float* kn, hipStream_t st) {
    const int cols = Hkv * D;
    hipLaunchKernelGGL(key_colsum_kernel, dim3(nblk), dim3(256), 0, st, key_cache, t0, n, cols, partial);
    hipLaunchKernelGGL(key_colmean_kernel, dim3((cols + 3) / 4), dim3(256), 0, st, partial, nblk, cols, n, avg);
    int64_t blocks = (n * Hkv * (D / 8) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(key_centre_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, key_cache, value_cache, t0, n,
                       Hkv, D, avg, M, kv, kn);
    return hipGetLastError();
}

hipError_t launch_attn_append(const uint16_t* k, const uint16_t* v, const int32_t* pos, int pos_delta,
                              const uint16_t* centre, int B, int Hkv, int D, int64_t M, uint16_t* kv, float* kn,
                              unsigned int* kn_ver, int* err, hipStream_t st) {
    hipLaunchKernelGGL(attn_append_kernel, dim3(B * Hkv), dim3(64), 0, st, k, v, pos, pos_delta, centre, Hkv, D,
                       M, kv, kn, kn_ver, err);
    return hipGetLastError();
}

// mp_attn_check: the arrival tickets of the in-launch merges (head_cnt) are zero between launches -- the merger of a
// head resets its counter.  A counter that is not zero means tickets were LOST: the cluster hand-off of
// lsh_decode_kernel takes them at workgroup scope in its XCD's L2, so members of a cluster that ran on different
// XCDs each draw ticket 0 from their own L2, nobody merges, nothing is written (and the XCC_ID comparison of the
// merger, err bit 4, never runs).  Sets err bit 8 and zeroes the counters so that later launches start clean.
__global__ void attn_ticket_check_kernel(int* __restrict__ head_cnt, int BH, int* __restrict__ err) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < BH && head_cnt[h] != 0) {
        head_cnt[h] = 0;
        atomicOr(err, 8);
    }
}

hipError_t launch_attn_ticket_check(int* head_cnt, int BH, int* err, hipStream_t st) {
    hipLaunchKernelGGL(attn_ticket_check_kernel, dim3((BH + 255) / 256), dim3(256), 0, st, head_cnt, BH, err);
    return hipGetLastError();
}

hipError_t launch_merge_state(const uint16_t* va, const float* sa, const uint16_t* vb,
                              const float* sb, int R, int D, uint16_t* v, float* s, hipStream_t st) {
    hipLaunchKernelGGL(merge_state_kernel, dim3(R), dim3(128), 0, st, va, sa, vb, sb, R, D, v, s);
    return hipGetLastError();
}

}  // namespace mp
