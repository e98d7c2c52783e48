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

// ---- the importance weight in ~70 VALU instructions (the libm calls it replaces -- acosf, 2 x log1pf,
// expm1f, logf -- are ~250; at 16 waves per CU a wave64 instruction occupies its SIMD for 4 cycles, so the
// transform was 2.7 us of every 32-token round at B*H >= CUs).  All pieces stay at <= ~1e-6 relative error,
// which the 1e-3 bound on the probabilities and 1 bf16 ulp on the output leave three orders of room for.
// acos on [-1, 1]: sqrt(1 - |x|) * P7(|x|) (Abramowitz & Stegun 4.4.46, |error| <= 2e-8), reflected for x < 0
__device__ __forceinline__ float acos_fast(float x) {
    const float a = fabsf(x);
    float p = -0.0012624911f;
    p = fmaf(p, a, 0.0066700901f);
    p = fmaf(p, a, -0.0170881256f);
    p = fmaf(p, a, 0.0308918810f);
    p = fmaf(p, a, -0.0501743046f);
    p = fmaf(p, a, 0.0889789874f);
    p = fmaf(p, a, -0.2145988016f);
    p = fmaf(p, a, 1.5707963050f);
    const float r = __fsqrt_rn(1.f - a) * p;
    return x < 0.f ? 3.14159265358979f - r : r;
}
// log(1 + x), x > -1: the alternating series to x^9 where |x| < 1/4 (truncation < 4e-7 relative), the native
// log of 1 + x elsewhere (1 + x is then exact to an ulp of a number away from 1); log1p(-1) = -inf
__device__ __forceinline__ float log1p_fast(float x) {
    float s = 0.11111111f;
    s = fmaf(s, x, -0.125f);
    s = fmaf(s, x, 0.14285714f);
    s = fmaf(s, x, -0.16666667f);
    s = fmaf(s, x, 0.2f);
    s = fmaf(s, x, -0.25f);
    s = fmaf(s, x, 0.33333333f);
    s = fmaf(s, x, -0.5f);
    s = fmaf(s * x, x, x);
    return fabsf(x) < 0.25f ? s : __logf(1.f + x);
}
// exp(y) - 1: Taylor to y^7 where |y| < 0.3 (truncation < 2e-9 relative), native exp elsewhere (|result| > 1/4)
__device__ __forceinline__ float expm1_fast(float y) {
    float s = 1.984127e-4f;
    s = fmaf(s, y, 1.3888889e-3f);
    s = fmaf(s, y, 8.3333333e-3f);
    s = fmaf(s, y, 4.1666667e-2f);
    s = fmaf(s, y, 0.16666667f);
    s = fmaf(s, y, 0.5f);
    s = fmaf(s * y, y, y);
    return fabsf(y) < 0.3f ? s : __expf(y) - 1.f;
}
// transform_kernel (sparse_attention.cc:164-184): logit of a sampled token = q.k / sqrt(D) - log(w + 1e-4),
// w = P[>= 2 of L tables collide] = 1 - (1-p)^(L-1) (L p + 1 - p), p = (1 - acos(cos) / pi)^K.  The reference
// evaluates w literally in f32 (two powf and a subtraction from 1), which loses ~3 digits to cancellation
// wherever w ~ 1e-4; here the same quantity without the cancellation: log X = (L-1) log1p(-p) + log1p((L-1) p),
// w = -expm1(log X).  cos is clamped to [-1, 1] (the reference NaNs when a bf16-rounded norm makes it > 1).
__device__ __forceinline__ float importance_logit(float sc, float qn_kn, float inv_sqrt_d, int K, int L) {
    const float cs = fminf(1.f, fmaxf(-1.f, sc * __frcp_rn(qn_kn)));
    const float proba = 1.f - acos_fast(cs) * 0.31830988618379067f;
    float p = 1.f, b = proba;
    for (int e = K; e; e >>= 1) {          // K is wave-uniform
        if (e & 1) p *= b;
        b *= b;
    }
    const float lm1 = (float)(L - 1);
    const float w = -expm1_fast(fmaf(lm1, log1p_fast(-p), log1p_fast(lm1 * p)));
    return fmaf(sc, inv_sqrt_d, -__logf(w + 1e-4f));
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

constexpr int AH_SLICE = 32;   // entries of the index list per wave step (16 for short lists, attn_head_fold)

// The workgroup owns slices slice0 + slice_stride * k, k = 0, 1, ... of the head's list (a slice is SLICE
// consecutive entries); wave w takes k = w, w + NW, ...  Layout of a step: see attn_head_fold.

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
// SLICE = tokens per wave step: 32 (AH_SLICE; 64 VGPRs of rows in flight) or, for lists short enough that one
// round of 16-token steps covers them (a decode member's ~190 ids at cfg 1), 16: twice the waves take part and
// every wave issues, waits for and reduces half as many rows -- the step is latency, not throughput.
// IDS: callable (int j) -> u32x4 holding entries j .. j+3 of the workgroup's list (j a multiple of 4; entries
// past nz are never used, but the call must not fault).
// A load instruction fetches RPL = 64/LPR rows, UPS = SLICE/RPL load steps cover the slice; token slot of
// (step u, row group r) is r*UPS + u, so the ids a row group needs are UPS consecutive entries.  The UPS
// partial dot products are reduce-scattered over the LPR lanes of a row group for log2(UPS) steps and
// all-reduced for the rest, so the DUP = LPR/UPS lanes c with equal c / DUP all hold the score of slot
// r*UPS + c/DUP: the importance transform runs DUP times per token, sums count one lane per token.
// kn_lds != nullptr (wave-uniform): the key norms of the workgroup's selected tokens sit in LDS as bf16, indexed by
// token - kn_t0 (lsh_decode_kernel scatters them there from the table entries' payload when a token is hit the second
// time): a token's norm is then a 2-byte LDS read instead of a random 4-byte HBM access -- one line request in five of
// the gather, which is bound by the requests a CU can keep in flight (EXPERIMENTS.md R3-10).
template <int D, int NW, bool DENSE, int SLICE, typename IDS>   // NW: upper bound of the workgroup's waves
__device__ __forceinline__ void attn_head_fold(
    AhState& st,
    const uint16_t* __restrict__ kv_g,   // kv rows of this head's kv group: [M][2][D]
    const float* __restrict__ kn_g,      // key norms of the group: [M]
    const u32x4 qv,                      // this lane's 8 query elements: bf16 q[(lane % LPR)*8 ..]
    float qn_h, int nz, int64_t M, int K, int L, int slice0, int slice_stride, IDS&& ids,
    float* __restrict__ score_h,         // [M] transformed logits (nullable)
    unsigned long long* __restrict__ stamp,
    int j0 = 0,                          // first entry of the list to fold (a multiple of 32): slices start there
    const uint16_t* kn_lds = nullptr, int kn_t0 = 0) {
    constexpr int LPR = D / 8;           // lanes per row (16 B each)
    constexpr int RPL = 64 / LPR;        // rows per load instruction
    constexpr int UPS = SLICE / RPL;     // load steps per slice
    constexpr int DUP = LPR / UPS;       // lanes that end up with the same token's score
    static_assert(UPS >= 4 && UPS % 4 == 0 && DUP >= 2, "ids are fetched four at a time");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int r = lane / LPR, c = lane % LPR;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    // row addresses = ONE uniform base (a scalar register pair) + a 32-bit byte offset per row (round 4): a token's K | V
    // rows are 4 D bytes, max_length <= 2^22 tokens, so every offset of a KV group is below 2^32 -- half the address
    // registers and none of the 64-bit multiply-adds of a pointer per row (A/B, same box: cfg 3 31.25 -> 30.4 us per layer,
    // cfg 2 33.9 -> 33.7, cfg 1 19.3 -> 19.2; EXPERIMENTS.md R4-2)
    // (mp_attn_alloc enforces max_length x 4 D <= 2^32 -- the one check this arithmetic rests on)
    const char* kvb = reinterpret_cast<const char*>(kv_g);
    const uint32_t coff = (uint32_t)c * 16u;

    for (int k = wave;; k += nw) {
        const int64_t s = (int64_t)slice0 + (int64_t)slice_stride * k;
        if (j0 + s * SLICE >= nz) break;
        const int jb = j0 + (int)s * SLICE;
        u32x4 idv[UPS / 4];
        if (!DENSE) {
#pragma unroll
            for (int v = 0; v < UPS / 4; ++v) idv[v] = ids(jb + r * UPS + v * 4);
        }
        const int slot_my = r * UPS + c / DUP;
        const int j_my = jb + slot_my;
        const bool valid_my = j_my < nz;

        // ---- gathers.  Slots past nz (last slice only) and bad indices are pointed at the slice's
        // first token, which is always a selected one: their weight is forced to 0, and every load
        // stays unconditional -- loads under a branch would make the compiler drain vmcnt at every
        // join.  Rows are read once and never reused: non-temporal loads.
        const int id_first = DENSE ? jb : __builtin_amdgcn_readfirstlane((int)idv[0][0]);
        const uint32_t M32 = (uint32_t)M;                               // max_length <= 2^32 / (4 D)
        const int id_safe = ((uint32_t)id_first < M32) ? id_first : 0;
        u32x4 kreg[UPS], vreg[UPS];
        int idc[UPS];
        int id_my = 0;
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
            int id_u = DENSE ? (jb + r * UPS + u) : (int)idv[u / 4][u % 4];
            const bool valid_u = (jb + r * UPS + u) < nz;
            if (!valid_u || (uint32_t)id_u >= M32) id_u = id_safe;
            if (u == c / DUP) id_my = id_u;
            idc[u] = id_u;
        }
        // Loads return in issue order.  Short lists (latency regime): the key norm first (the transform needs it
        // right after the K rows), then all K rows, then the V rows (needed last): 0.45 us per step at cfg 1.
        // Full-size steps run with HBM saturated (B*H >= CUs): there the K and the V row of a token, which share
        // a DRAM page, are requested back to back (splitting them cost 2 % at cfg 2).
        // the key norm FIRST in both forms: the transform needs it right behind the K rows; as the youngest load of a
        // full-size step it held the transform back until every V row was in (cfg 2 on clustered keys: 45.6 -> 43.3 us)
        float kn_my = 1.f;
        if (!DENSE) {
            if (kn_lds != nullptr) kn_my = bf16_bits_to_f32(kn_lds[id_my - kn_t0]);
            else kn_my = kn_g[id_my];
        }
        uint32_t ro[UPS];
#pragma unroll
        for (int u = 0; u < UPS; ++u) ro[u] = (uint32_t)idc[u] * (uint32_t)(4 * D) + coff;
        if (SLICE < AH_SLICE) {
#pragma unroll
            for (int u = 0; u < UPS; ++u)
                kreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvb + ro[u]));
#pragma unroll
            for (int u = 0; u < UPS; ++u)
                vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvb + ro[u] + 2 * D));
        } else {
#pragma unroll
            for (int u = 0; u < UPS; ++u) {
                kreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvb + ro[u]));
                vreg[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kvb + ro[u] + 2 * D));
            }
        }
        if (!DENSE && k == wave) MP_STAMP(stamp, 34);

        // ---- q . K partials, reduce-scatter over the row group while more than one value is left, then all-reduce
        float part[UPS];
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
            float a = 0.f;
            dot8_bf16_chain(a, kreg[u], qv);
            dot_settle(a);
            part[u] = a;
        }
#pragma unroll
        for (int stp = LPR / 2, vals = UPS; stp >= 1; stp >>= 1) {
            if (vals > 1) {
                const int half = vals / 2;
                const bool upper = (c & stp) != 0;
#pragma unroll
                for (int u = 0; u < UPS / 2; ++u) {
                    if (u < half) {
                        const float send = upper ? part[u] : part[u + half];
                        const float keep = upper ? part[u + half] : part[u];
                        part[u] = keep + __shfl_xor(send, stp);
                    }
                }
                vals = half;
            } else {
                part[0] += __shfl_xor(part[0], stp);
            }
        }
        const float sc = part[0];                            // = q . K[id_my] on the DUP lanes of slot_my
        if (!DENSE && k == wave) MP_STAMP(stamp, 35);

        // ---- importance-sampling transform (transform_kernel, sparse_attention.cc:164-184): importance_logit
        float z = -INFINITY;
        if (valid_my) {
            if (DENSE) {
                z = sc * inv_sqrt_d;
            } else {
                z = importance_logit(sc, qn_h * kn_my, inv_sqrt_d, K, L);
            }
            if (score_h != nullptr && (c % DUP) == 0) score_h[j_my] = z;
        }
        const float m_w = wave_max(z);
        const float p_my = valid_my ? __expf(z - m_w) : 0.f;    // slice non-empty => m_w finite
        const float l_w = wave_sum((c % DUP) ? 0.f : p_my);
        if (!DENSE && k == wave) MP_STAMP(stamp, 36);

        // ---- P . V
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
            const float pu = __shfl(p_my, r * LPR + DUP * u);
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
        if (LPR == 16) st.o1 = fmaf(a, st.o1, b * acc[1]);
        st.m = m_new;
    }
}

// The fold of the LEAN decode (lsh.hip: MP_DECODE_NO_BYPRODUCTS).  Same arithmetic per token as attn_head_fold; what differs:
//   * the list holds RAW table words  id | (bf16 key norm bits 14..0 << idbits)  as the counting step found them -- the
//     norm of a selected token comes out of its own list entry (word_norm; else one HBM read per token as before), no
//     per-range norm array in LDS;
//   * the slices k0, k0 + kstep, ... of the list are folded: (0, 1) for a list that belongs to the calling WAVE alone
//     (the tokens whose second collision this wave counted: gathered the moment its own counting is done, without waiting
//     for the workgroup), (wave, waves) for a list shared by the workgroup;
//   * rows are requested through a buffer descriptor over the KV group's rows: a slot past the end of the list gets an
//     offset beyond num_records, for which the hardware returns zeros WITHOUT a memory request (attn_head_fold points such
//     slots at the slice's first token: a wave-owned list of ~12 tokens in a 16-token step would re-request that row four
//     times -- EXPERIMENTS.md R4-1 (2)).
template <int D, int SLICE, typename IDS>
__device__ __forceinline__ void attn_head_fold_lean(
    AhState& st, const uint16_t* __restrict__ kv_g, const float* __restrict__ kn_g, const u32x4 qv, float qn_h, int nz,
    int64_t M, int K, int L, int k0, int kstep, IDS&& ids, uint32_t idmask, int idbits, bool word_norm,
    unsigned long long* __restrict__ stamp) {
    constexpr int LPR = D / 8, RPL = 64 / LPR, UPS = SLICE / RPL, DUP = LPR / UPS;
    static_assert(UPS >= 4 && UPS % 4 == 0 && DUP >= 2, "ids are fetched four at a time");
    const int lane = threadIdx.x & 63;
    const int r = lane / LPR, c = lane % LPR;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    const uint64_t kv_bytes = (uint64_t)M * (uint64_t)(4 * D);                   // <= 2^32 (mp_attn_alloc)
    const __amdgpu_buffer_rsrc_t rkv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(kv_g), 0, (int)(uint32_t)(kv_bytes > 0xffffffffull ? 0xffffffffull : kv_bytes), 0x00020000);
    constexpr int kNt = 2;                                                      // aux bit 1 = nt on gfx940+
    constexpr uint32_t kOut = 0xffffff00u;                                      // beyond any row: no request, zeros
    const uint32_t coff = (uint32_t)c * 16u;
    const uint32_t M32 = (uint32_t)M;
    for (int k = k0;; k += kstep) {
        const int jb = k * SLICE;
        if (jb >= nz) break;
        u32x4 wv[UPS / 4];
#pragma unroll
        for (int v = 0; v < UPS / 4; ++v) wv[v] = ids(jb + r * UPS + v * 4);
        const int slot_my = r * UPS + c / DUP;
        const bool valid_my = jb + slot_my < nz;
        uint32_t ro[UPS];
        uint32_t w_my = 0u;
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
            const uint32_t w = wv[u / 4][u % 4];
            const uint32_t id = w & idmask;
            if (u == c / DUP) w_my = w;
            ro[u] = ((jb + r * UPS + u) < nz && id < M32) ? id * (uint32_t)(4 * D) + coff : kOut;
        }
        float kn_my = 1.f;
        if (word_norm) kn_my = bf16_bits_to_f32((uint16_t)(w_my >> idbits));
        else if (valid_my && (w_my & idmask) < M32) kn_my = kn_g[w_my & idmask];
        u32x4 kreg[UPS], vreg[UPS];
        if (SLICE < AH_SLICE) {
#pragma unroll
            for (int u = 0; u < UPS; ++u) kreg[u] = __builtin_amdgcn_raw_buffer_load_b128(rkv, ro[u], 0, kNt);
#pragma unroll
            for (int u = 0; u < UPS; ++u) vreg[u] = __builtin_amdgcn_raw_buffer_load_b128(rkv, ro[u] + 2 * D, 0, kNt);
        } else {
#pragma unroll
            for (int u = 0; u < UPS; ++u) {
                kreg[u] = __builtin_amdgcn_raw_buffer_load_b128(rkv, ro[u], 0, kNt);
                vreg[u] = __builtin_amdgcn_raw_buffer_load_b128(rkv, ro[u] + 2 * D, 0, kNt);
            }
        }
        if (k == k0) MP_STAMP(stamp, 34);
        float part[UPS];
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
            float a = 0.f;
            dot8_bf16_chain(a, kreg[u], qv);
            dot_settle(a);
            part[u] = a;
        }
#pragma unroll
        for (int stp = LPR / 2, vals = UPS; stp >= 1; stp >>= 1) {
            if (vals > 1) {
                const int half = vals / 2;
                const bool upper = (c & stp) != 0;
#pragma unroll
                for (int u = 0; u < UPS / 2; ++u) {
                    if (u < half) {
                        const float send = upper ? part[u] : part[u + half];
                        const float keep = upper ? part[u + half] : part[u];
                        part[u] = keep + __shfl_xor(send, stp);
                    }
                }
                vals = half;
            } else {
                part[0] += __shfl_xor(part[0], stp);
            }
        }
        const float sc = part[0];
        if (k == k0) MP_STAMP(stamp, 35);
        float z = -INFINITY;
        if (valid_my) z = importance_logit(sc, qn_h * kn_my, inv_sqrt_d, K, L);
        const float m_w = wave_max(z);
        const float p_my = valid_my ? __expf(z - m_w) : 0.f;
        const float l_w = wave_sum((c % DUP) ? 0.f : p_my);
        if (k == k0) MP_STAMP(stamp, 36);
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
            const float pu = __shfl(p_my, r * LPR + DUP * u);
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
        if (k == k0) MP_STAMP(stamp, 37);
        const float m_new = fmaxf(st.m, m_w);
        const float a = __expf(st.m - m_new), b = __expf(m_w - m_new);
        st.l = fmaf(a, st.l, b * l_w);
        st.o0 = fmaf(a, st.o0, b * acc[0]);
        if (LPR == 16) st.o1 = fmaf(a, st.o1, b * acc[1]);
        st.m = m_new;
    }
}

// The waves' states meet in LDS (one barrier).  On return the lanes of WAVE 0 hold the workgroup's merged
// state -- m (max logit), Z (sum of exp(z - m)) and, per lane, VPL = D / 64 consecutive elements
// o[lane * VPL ..] of sum_j exp(z_j - m) V[j] -- and the other waves are done (they have nothing left to do in
// the kernels that use this: a hand-off by ONE wave needs no further workgroup barrier).  m = -inf, Z = 0 when
// the workgroup had no slice.
// FULL: the workgroup has exactly NW waves -- the reads below are then straight-line code; with a run-time wave
// count every read sits behind a branch, and the compiler puts an s_waitcnt vmcnt(0) in front of each, i.e. wave 0
// first waits for the acknowledgement of the score stores of its last step (~0.5 us on the path to the hand-off).
template <int D, int NW, bool FULL = false>
__device__ __forceinline__ void attn_head_merge(const AhState& st, float* s_merge, float& m_out,
                                                float& Z_out, float& o0_out, float& o1_out) {
    constexpr int VPL = D / 64;          // 2 (D = 128) or 1 (D = 64)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = FULL ? NW : (int)(blockDim.x >> 6);
    float* mine = s_merge + wave * (D + 2);
    mine[st.d0] = st.o0;
    if (D / 8 == 16) mine[st.d0 + 1] = st.o1;
    if (lane == 0) {
        mine[D] = st.m;
        mine[D + 1] = st.l;
    }
    __syncthreads();
    m_out = -INFINITY;
    Z_out = 0.f;
    o0_out = 0.f;
    o1_out = 0.f;
    if (wave == 0) {
        // all reads of a pass are issued back to back (a rolled loop over the waves serialises 2 x NW
        // dependent LDS round trips: 2.1 us measured)
        float mw[NW], lw[NW], oa[NW], ob[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const bool live = FULL || w < nw;
            mw[w] = live ? s_merge[w * (D + 2) + D] : -INFINITY;
            lw[w] = live ? s_merge[w * (D + 2) + D + 1] : 0.f;
            if (VPL == 2) {
                const float2 t = live ? *reinterpret_cast<const float2*>(s_merge + w * (D + 2) + lane * 2)
                                      : make_float2(0.f, 0.f);
                oa[w] = t.x;
                ob[w] = t.y;
            } else {
                oa[w] = live ? s_merge[w * (D + 2) + lane] : 0.f;
                ob[w] = 0.f;
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) m = fmaxf(m, mw[w]);
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
        m_out = m;
        Z_out = Z;
        o0_out = o0;
        o1_out = o1;
    }
}

// The same meeting WITHOUT the workgroup barrier (the decode kernel): every wave
// leaves its state in LDS and draws an LDS ticket; the wave that draws the last one holds the merged state on return
// (true), the others are done (false) -- the workgroup then waits for its slowest wave's rows once, in that wave, instead
// of rows -> barrier -> wave 0's wake-up.  Ordering: a wave's LDS instructions execute in issue order, so its ticket is
// drawn after its state is written, and the last drawer's reads follow every other wave's writes.  No fence: a
// workgroup-scope release would also wait for the wave's outstanding score stores (vmcnt), which is what FULL avoids.
template <int D, int NW>
__device__ __forceinline__ bool attn_head_merge_ticket(const AhState& st, float* s_merge, int* s_ticket, float& m_out,
                                                       float& Z_out, float& o0_out, float& o1_out) {
    constexpr int VPL = D / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = s_merge + wave * (D + 2);
    mine[st.d0] = st.o0;
    if (D / 8 == 16) mine[st.d0 + 1] = st.o1;
    if (lane == 0) {
        mine[D] = st.m;
        mine[D + 1] = st.l;
    }
    asm volatile("" ::: "memory");
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(s_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t = __builtin_amdgcn_readfirstlane(t);
    asm volatile("" ::: "memory");
    m_out = -INFINITY;
    Z_out = 0.f;
    o0_out = 0.f;
    o1_out = 0.f;
    if (t != NW - 1) return false;
    float mw[NW], lw[NW], oa[NW], ob[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        mw[w] = s_merge[w * (D + 2) + D];
        lw[w] = s_merge[w * (D + 2) + D + 1];
        if (VPL == 2) {
            const float2 tt = *reinterpret_cast<const float2*>(s_merge + w * (D + 2) + lane * 2);
            oa[w] = tt.x;
            ob[w] = tt.y;
        } else {
            oa[w] = s_merge[w * (D + 2) + lane];
            ob[w] = 0.f;
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fmaxf(m, mw[w]);
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
    m_out = m;
    Z_out = Z;
    o0_out = o0;
    o1_out = o1;
    return true;
}

// The two halves of attn_head_merge_ticket for a caller that does more between them (the LEAN decode: the wave that draws
// the last ticket may still fold the rare leftovers -- pooled chunks, a spill list -- into its own state and publish again).
template <int D>
__device__ __forceinline__ void attn_head_publish(const AhState& st, float* s_merge) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = s_merge + wave * (D + 2);
    mine[st.d0] = st.o0;
    if (D / 8 == 16) mine[st.d0 + 1] = st.o1;
    if (lane == 0) {
        mine[D] = st.m;
        mine[D + 1] = st.l;
    }
}
template <int D, int NW>
__device__ __forceinline__ void attn_head_merge_read(const float* s_merge, float& m_out, float& Z_out, float& o0_out,
                                                     float& o1_out) {
    constexpr int VPL = D / 64;
    const int lane = threadIdx.x & 63;
    float mw[NW], lw[NW], oa[NW], ob[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        mw[w] = s_merge[w * (D + 2) + D];
        lw[w] = s_merge[w * (D + 2) + D + 1];
        if (VPL == 2) {
            const float2 tt = *reinterpret_cast<const float2*>(s_merge + w * (D + 2) + lane * 2);
            oa[w] = tt.x;
            ob[w] = tt.y;
        } else {
            oa[w] = s_merge[w * (D + 2) + lane];
            ob[w] = 0.f;
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fmaxf(m, mw[w]);
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
    m_out = m;
    Z_out = Z;
    o0_out = o0;
    o1_out = o1;
}

// fold the slices of ONE sparse list, then merge (attn_head_kernel)
template <int D, int NW, int SLICE, typename IDS>
__device__ __forceinline__ void attn_head_tail(
    const uint16_t* __restrict__ kv_g, const float* __restrict__ kn_g, const u32x4 qv, float qn_h, int nz,
    int64_t M, int K, int L, int slice0, int slice_stride, IDS&& ids, float* s_merge,
    float* __restrict__ score_h, unsigned long long* __restrict__ stamp, float& m_out, float& Z_out,
    float& o0_out, float& o1_out) {
    AhState st = ah_state_init(threadIdx.x & 63, D / 8);
    attn_head_fold<D, NW, false, SLICE>(st, kv_g, kn_g, qv, qn_h, nz, M, K, L, slice0, slice_stride, ids, score_h, stamp);
    attn_head_merge<D, NW>(st, s_merge, m_out, Z_out, o0_out, o1_out);
}

// wave 0 (the layout attn_head_merge leaves): out = o / Z as bf16 (RNE); max_value_expsum[0] = m*log2e,
// [1] = log2 Z + m*log2e (softmax_kernel, sparse_attention.cc:238-239); Z = 0 (no token at all): out = 0,
// LSE = -inf (SURVEY a-10)
template <int D>
__device__ __forceinline__ void attn_head_finalize(float m, float Z, float o0, float o1, uint16_t* __restrict__ out_h,
                                                   float* __restrict__ mve, int BH, int h,
                                                   float2* __restrict__ head_mz) {
    const int lane = threadIdx.x & 63;
    const bool none = !(Z > 0.f);
    if (D == 128) {
        const uint32_t pk = none ? 0u : ((uint32_t)f32_to_bf16_rne(o0 / Z) | ((uint32_t)f32_to_bf16_rne(o1 / Z) << 16));
        reinterpret_cast<uint32_t*>(out_h)[lane] = pk;
    } else {
        out_h[lane] = none ? (uint16_t)0 : f32_to_bf16_rne(o0 / Z);
    }
    if (lane == 0) {
        const float mv = none ? -INFINITY : m * 1.4426950408889634f;
        mve[h] = mv;
        mve[BH + h] = none ? -INFINITY : log2f(Z) + mv;
        head_mz[h] = make_float2(none ? -INFINITY : m, none ? 0.f : Z);
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
