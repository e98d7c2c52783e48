// attn_head.h -- sparse attention of ONE head by ONE workgroup (gfx950), the tail of the fused
// decode kernel (lsh.hip: lsh_decode_kernel).  Same math as attn_sparse_kernel (attention.hip:
// qk_kernel / transform_kernel / softmax_kernel / wv_kernel of
// library/sparse_attention/sparse_attention.cc:38-518), different decomposition:
//   * every wave of the workgroup owns the 32-entry slices wave, wave + NW, ... of the head's
//     index list and folds them into a running (max, sum, o[D]) state in registers;
//   * the waves' states meet in LDS (one barrier) -- no partials in HBM, no arrival tickets,
//     no write-through traffic between XCDs;
//   * a step gathers K and V rows of 32 tokens at once: 64 VGPRs = 16 KB per wave in flight, which
//     is what a lane of a 1024-thread workgroup (128 VGPRs) can spare.
#pragma once
#include "common.h"

namespace mp {

__device__ __forceinline__ float powi_u(float b, int e) {
    float r = 1.f;
    while (e) {          // e is wave-uniform
        if (e & 1) r *= b;
        b *= b;
        e >>= 1;
    }
    return r;
}

// one reduce-scatter step over lanes l and l^ST: N values -> N/2 values per lane
template <int N, int ST>
__device__ __forceinline__ void rs_step_h(float (&v)[8], int lane, int& doff) {
    constexpr int half = N / 2;
    const bool upper = (lane & ST) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
        const float send = upper ? v[i] : v[i + half];
        const float keep = upper ? v[i + half] : v[i];
        v[i] = keep + __shfl_xor(send, ST);
    }
    doff += upper ? half : 0;
}

// LDS scratch of attn_head_tail: NW * (D + 2) floats
__host__ __device__ constexpr int attn_head_lds_floats(int nw, int D) { return nw * (D + 2); }

constexpr int AH_SLICE = 32;   // entries of the index list per wave step

// The workgroup owns slices slice0 + slice_stride * k, k = 0, 1, ... of the head's list (a slice is
// AH_SLICE = 32 consecutive entries); wave w takes k = w, w + NW, ...
// IDS: callable (int k, int j) -> u32x4 holding entries j .. j+3 of the workgroup's k-th slice
// (entries past nz are never used, but the call must not fault).
//
// A step gathers the K row AND the V row of 32 tokens (64 VGPRs in flight, all that a lane of a
// 1024-thread workgroup can spare): LPR = D/8 lanes cover a row, a load instruction fetches
// RPL = 64/LPR rows, UPS = LPR/2 load steps cover RPL * UPS = 32 tokens; token slot of (step u, row
// group r) is r*UPS + u, so the ids a row group needs are UPS consecutive entries.  The reduce-scatter
// of the UPS partial dot products over the LPR lanes of a row group ends one step early (st = 2) and
// finishes with an all-reduce, so lanes c and c^1 both hold the score of slot r*UPS + (c >> 1): the
// importance transform runs twice per token (VALU is idle anyway), sums count even lanes only.

// a wave's running softmax state over the slices it has folded in
struct AhState {
    float m, l, o0, o1;   // max logit, sum of exp(z - m), this lane's <= 2 elements of sum exp(z - m) V
    int d0;               // first element of o[] this lane holds
};
__device__ __forceinline__ AhState ah_state_init(int lane, int lpr) {
    AhState st;
    st.m = -INFINITY;
    st.l = 0.f;
    st.o0 = 0.f;
    st.o1 = 0.f;
    st.d0 = (lane % lpr) * 8;
    return st;
}

// Fold the workgroup's slices of one index list into the waves' states.  DENSE: the list is
// 0 .. nz-1 itself and the logit is q.k / sqrt(D) with no importance transform (full_attention,
// sparse_attention.cc:988-1037; the static window of models/attnserver.py:293-296) -- `ids`, `kn_g`,
// `qn_h`, K and L are not used.
template <int D, int NW, bool DENSE, typename IDS>   // NW: upper bound of the workgroup's waves (blockDim.x / 64 <= NW)
__device__ __forceinline__ void attn_head_fold(
    AhState& st,
    const uint16_t* __restrict__ kv_g,   // kv rows of this head's kv group: [M][2][D]
    const float* __restrict__ kn_g,      // key norms of the group: [M]
    const u32x4 qv,                      // this lane's 8 query elements: bf16 q[(lane % LPR)*8 ..]
    float qn_h, int nz, int64_t M, int K, int L, int slice0, int slice_stride, IDS&& ids,
    float* __restrict__ score_h,         // [M] transformed logits (nullable)
    unsigned long long* __restrict__ stamp) {
    constexpr int LPR = D / 8;           // lanes per row (16 B each)
    constexpr int UPS = LPR / 2;         // load steps per slice
    constexpr int VPL = (LPR == 16) ? 2 : 1;
    static_assert((64 / LPR) * UPS == AH_SLICE, "a step covers 32 tokens");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int r = lane / LPR, c = lane % LPR;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    const uint16_t* kvc = kv_g + c * 8;

    for (int k = wave;; k += nw) {
        const int64_t s = (int64_t)slice0 + (int64_t)slice_stride * k;
        if (s * AH_SLICE >= nz) break;
        const int jb = (int)s * AH_SLICE;
        u32x4 idv[UPS / 4];
        if (!DENSE) {
#pragma unroll
            for (int v = 0; v < UPS / 4; ++v) idv[v] = ids(k, r * UPS + v * 4);
        }
        const int slot_my = r * UPS + (c >> 1);
        const int j_my = jb + slot_my;
        const bool valid_my = j_my < nz;

        // ---- gathers.  Slots past nz (last slice only) and bad indices are pointed at the slice's
        // first token, which is always a selected one: their weight is forced to 0, and every load
        // stays unconditional -- loads under a branch would make the compiler drain vmcnt at every
        // join.  Rows are read once and never reused: non-temporal loads.
        const int id_first = DENSE ? jb : __builtin_amdgcn_readfirstlane((int)idv[0][0]);
        const uint32_t M32 = (uint32_t)M;                               // max_length <= 2^22
        const int id_safe = ((uint32_t)id_first < M32) ? id_first : 0;
        u32x4 kreg[UPS], vreg[UPS];
        int id_my = 0;
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
            int id_u = DENSE ? (jb + r * UPS + u) : (int)idv[u / 4][u % 4];
            const bool valid_u = (jb + r * UPS + u) < nz;
            if (!valid_u || (uint32_t)id_u >= M32) id_u = id_safe;
            if (u == (c >> 1)) id_my = id_u;
            const uint16_t* row = kvc + (int64_t)id_u * 2 * D;
            kreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row));
            vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + D));
        }
        float kn_my = 1.f;
        if (!DENSE) kn_my = kn_g[id_my];
        if (!DENSE && k == wave) MP_STAMP(stamp, 34);

        // ---- q . K partials, reduce-scatter over the row group down to pairs, then all-reduce
        float part[UPS];
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
            float a = 0.f;
            dot8_bf16_chain(a, kreg[u], qv);
            dot_settle(a);
            part[u] = a;
        }
#pragma unroll
        for (int stp = LPR / 2; stp >= 2; stp >>= 1) {
            const bool upper = (c & stp) != 0;
#pragma unroll
            for (int u = 0; u < stp / 2; ++u) {
                const float send = upper ? part[u] : part[u + stp / 2];
                const float keep = upper ? part[u + stp / 2] : part[u];
                part[u] = keep + __shfl_xor(send, stp);
            }
        }
        const float sc = part[0] + __shfl_xor(part[0], 1);   // = q . K[id_my] on lanes c and c^1
        if (!DENSE && k == wave) MP_STAMP(stamp, 35);

        // ---- importance-sampling transform (transform_kernel, sparse_attention.cc:164-184);
        // cancellation-free weight and cos clamp as in attn_sparse_kernel
        float z = -INFINITY;
        if (valid_my) {
            if (DENSE) {
                z = sc * inv_sqrt_d;
            } else {
                float cs = sc / (qn_h * kn_my);
                cs = fminf(1.f, fmaxf(-1.f, cs));
                const float theta = acosf(cs);
                const float proba = 1.f - theta * 0.31830988618379067f;
                const float p = powi_u(proba, K);
                const float lm1 = (float)(L - 1);
                const float w = -expm1f(lm1 * log1pf(-p) + log1pf(lm1 * p));
                z = sc * inv_sqrt_d - logf(w + 1e-4f);
            }
            if (score_h != nullptr && (c & 1) == 0) score_h[j_my] = z;
        }
        const float m_w = wave_max(z);
        const float p_my = valid_my ? __expf(z - m_w) : 0.f;    // slice non-empty => m_w finite
        const float l_w = wave_sum((c & 1) ? 0.f : p_my);
        if (!DENSE && k == wave) MP_STAMP(stamp, 36);

        // ---- P . V
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
        st.d0 = c * 8 + doff;
        if (!DENSE && k == wave) MP_STAMP(stamp, 37);

        // ---- fold the slice into the wave's running state
        const float m_new = fmaxf(st.m, m_w);
        const float a = __expf(st.m - m_new), b = __expf(m_w - m_new);   // exp(-inf) = 0 on the first slice
        st.l = fmaf(a, st.l, b * l_w);
        st.o0 = fmaf(a, st.o0, b * acc[0]);
        if (VPL == 2) st.o1 = fmaf(a, st.o1, b * acc[1]);
        st.m = m_new;
    }
}

// The waves' states meet in LDS (one barrier).  On return threads tid < D hold the workgroup's
// merged state: m (max logit), Z (sum of exp(z - m)) and o = sum_j exp(z_j - m) V[j][tid];
// m = -inf, Z = 0 when the workgroup had no slice.
template <int D, int NW>
__device__ __forceinline__ void attn_head_merge(const AhState& st, float* s_merge, float& m_out,
                                                float& Z_out, float& o_out) {
    constexpr int VPL = (D / 8 == 16) ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    float* mine = s_merge + wave * (D + 2);
    mine[st.d0] = st.o0;
    if (VPL == 2) mine[st.d0 + 1] = st.o1;
    if (lane == 0) {
        mine[D] = st.m;
        mine[D + 1] = st.l;
    }
    __syncthreads();
    m_out = -INFINITY;
    Z_out = 0.f;
    o_out = 0.f;
    if (tid < D) {
        // all reads of a pass are issued back to back (a rolled loop over the waves serialises 2 x NW
        // dependent LDS round trips: 2.1 us measured)
        float mw[NW], lw[NW], ow[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const bool live = w < nw;
            mw[w] = live ? s_merge[w * (D + 2) + D] : -INFINITY;
            lw[w] = live ? s_merge[w * (D + 2) + D + 1] : 0.f;
            ow[w] = live ? s_merge[w * (D + 2) + tid] : 0.f;
        }
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) m = fmaxf(m, mw[w]);
        float Z = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (mw[w] != -INFINITY) {                    // a wave without a slice never wrote its o[]
                const float e = __expf(mw[w] - m);
                Z = fmaf(e, lw[w], Z);
                o = fmaf(e, ow[w], o);
            }
        }
        m_out = m;
        Z_out = Z;
        o_out = o;
    }
}

// fold the slices of ONE sparse list, then merge (attn_head_kernel)
template <int D, int NW, typename IDS>
__device__ __forceinline__ void attn_head_tail(
    const uint16_t* __restrict__ kv_g, const float* __restrict__ kn_g, const u32x4 qv, float qn_h, int nz,
    int64_t M, int K, int L, int slice0, int slice_stride, IDS&& ids, float* s_merge,
    float* __restrict__ score_h, unsigned long long* __restrict__ stamp, float& m_out, float& Z_out,
    float& o_out) {
    AhState st = ah_state_init(threadIdx.x & 63, D / 8);
    attn_head_fold<D, NW, false>(st, kv_g, kn_g, qv, qn_h, nz, M, K, L, slice0, slice_stride, ids, score_h, stamp);
    attn_head_merge<D, NW>(st, s_merge, m_out, Z_out, o_out);
}

// threads tid < D: out = o / Z as bf16 (RNE); max_value_expsum[0] = m*log2e, [1] = log2 Z + m*log2e
// (softmax_kernel, sparse_attention.cc:238-239)
template <int D>
__device__ __forceinline__ void attn_head_finalize(float m, float Z, float o, uint16_t* __restrict__ out_h,
                                                   float* __restrict__ mve, int BH, int h,
                                                   float2* __restrict__ head_mz) {
    const int tid = threadIdx.x;
    if (tid < D) {
        out_h[tid] = f32_to_bf16_rne(o / Z);
        if (tid == 0) {
            const float mv = m * 1.4426950408889634f;
            mve[h] = mv;
            mve[BH + h] = log2f(Z) + mv;
            head_mz[h] = make_float2(m, Z);
        }
    }
}

// empty head: out = 0, LSE = -inf (SURVEY a-10)
template <int D>
__device__ __forceinline__ void attn_head_empty(uint16_t* __restrict__ out_h, float* __restrict__ mve, int BH,
                                                int h, float2* __restrict__ head_mz) {
    if (threadIdx.x < D) out_h[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        mve[h] = -INFINITY;
        mve[BH + h] = -INFINITY;
        head_mz[h] = make_float2(-INFINITY, 0.f);
    }
}

}  // namespace mp
