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
//   * split-KV over the index list: a SLICE is 256 consecutive entries of one head's `ind`;
//     a persistent grid walks all slices of all heads (slice -> head by a prefix sum over
//     ceil(nnz/256) computed on device: no host readback of nnz);
//   * a wave owns 64 tokens; one global_load_dwordx4 fetches 4 whole rows (16 lanes x 16 B per
//     256-byte row), so a wave keeps 16 K-row loads + 16 V-row loads (32 KB) in flight;
//   * q.K: 4 v_dot2c_f32_bf16 per row chunk, then a 15-step reduce-scatter over the 16 lanes
//     of a row group leaves exactly ONE token's score per lane, so the transcendental
//     importance transform (acos, two integer powers, log, exp) runs once per token per lane;
//   * softmax is per wave (max / sum by DPP shuffles), P.V accumulates 8 f32 per lane, the 4
//     waves of a slice combine through LDS into one (m, l, o[D]) partial; a second small
//     kernel merges the partials of a head and writes bf16 out + base-2 LSE.
#include <hip/hip_ext.h>

#include "common.h"

namespace mp {

constexpr int AT_THREADS = 256;
constexpr int AT_WAVES = AT_THREADS / 64;
constexpr int AT_SLICE = AT_WAVES * 64;   // tokens per slice

__device__ __forceinline__ float powi(float b, int e) {
    float r = 1.f;
    while (e) {          // e is wave-uniform
        if (e & 1) r *= b;
        b *= b;
        e >>= 1;
    }
    return r;
}

// 8-element bf16 dot product: four chained v_dot2c_f32_bf16 (D += a.lo*b.lo + a.hi*b.hi).
// Inline asm on purpose: with ROCm 7.2's hipcc, __builtin_amdgcn_fdot2_f32_bf16 fed from
// ext-vector element extracts selects element 0 for EVERY call (scripts/probe_dot2.hip shows
// `v_dot2c_f32_bf16 v, v2, v6` four times); the asm form is correct.  hipcc pads nothing
// inside asm, so the gfx940-class DOT hazards are handled here: a DOT result may feed the next
// same-opcode DOT as the accumulator with 0 wait states, but any other VALU read / write of it
// needs 3 / 4 wait states (LLVM GCNHazardRecognizer, DotWriteDifferentVALURead/Write) ->
// `s_nop 3` after the chain.
__device__ __forceinline__ float dot8_bf16(const u32x4& k, const uint32_t (&q)[4]) {
    float acc = 0.f;
    asm("v_dot2c_f32_bf16 %0, %1, %5\n\t"
        "v_dot2c_f32_bf16 %0, %2, %6\n\t"
        "v_dot2c_f32_bf16 %0, %3, %7\n\t"
        "v_dot2c_f32_bf16 %0, %4, %8\n\t"
        "s_nop 3"
        : "+v"(acc)
        : "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]));
    return acc;
}

__device__ __forceinline__ int slices_of(int nz, int64_t M) {
    if (nz < 0) nz = 0;
    if ((int64_t)nz > M) nz = (int)M;
    return (nz + AT_SLICE - 1) / AT_SLICE;
}

// D: head_dim (64 or 128); DENSE: ids are 0..nnz-1 and no importance transform (full_attention);
// QBF16: query is bf16 (the reference's __AVX512BF16__ family) else f32.
template <int D, bool DENSE, bool QBF16>
__global__ __launch_bounds__(AT_THREADS) void attn_partial_kernel(
    const uint16_t* __restrict__ kv,     // [B*Hkv][M][2][D]
    const float* __restrict__ kn,        // [B*Hkv][M]
    const void* __restrict__ query,      // [BH][D] bf16 or f32
    const float* __restrict__ qnorm,     // [BH]
    const int32_t* __restrict__ ind,     // [BH][M]
    const int32_t* __restrict__ nnz,     // [BH]
    float* __restrict__ part_o,          // [slices][D]
    float2* __restrict__ part_ml,        // [slices] (max logit, sum exp)
    float* __restrict__ score,           // [BH][M] transformed logits z_j (nullable)
    int BH, int G, int64_t M, int K, int L) {
    constexpr int LPR = D / 8;           // lanes per row (16 B each)
    constexpr int RPL = 64 / LPR;        // rows per wave load
    extern __shared__ int s_pre[];       // [BH + 1] slice prefix
    __shared__ float s_o[AT_WAVES][D];
    __shared__ float s_m[AT_WAVES], s_l[AT_WAVES];
    __shared__ int s_tmp[AT_WAVES + 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane / LPR, c = lane % LPR;

    // ---- slice prefix over heads (every workgroup computes it; BH ints from L2)
    int carry = 0;
    for (int base = 0; base < BH; base += AT_THREADS) {
        const int hh = base + tid;
        const int v = (hh < BH) ? slices_of(nnz[hh], M) : 0;
        int total;
        __syncthreads();
        const int ex = block_excl_scan(v, s_tmp, total);
        if (hh < BH) s_pre[hh] = carry + ex;
        carry += total;
    }
    if (tid == 0) s_pre[BH] = carry;
    __syncthreads();
    const int total_slices = s_pre[BH];
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);

    for (int s = blockIdx.x; s < total_slices; s += gridDim.x) {
        // head of slice s: first h with s_pre[h + 1] > s
        int lo = 0, hi = BH - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_pre[mid + 1] > s) hi = mid; else lo = mid + 1;
        }
        const int h = lo;
        const int64_t g = h / G;
        int nz = nnz[h];
        if ((int64_t)nz > M) nz = (int)M;
        const int jb = (s - s_pre[h]) * AT_SLICE + wave * 64;   // first token of this wave

        // my token (the one whose score lands on this lane after the reduce-scatter)
        const int j_my = jb + c * RPL + r;
        const bool valid_my = j_my < nz;
        int id_my = 0;
        if (valid_my) {
            id_my = DENSE ? j_my : ind[(int64_t)h * M + j_my];
            if (id_my < 0 || (int64_t)id_my >= M) id_my = 0;   // never fault on a bad index
        }
        float kn_my = 1.f;
        if (!DENSE && valid_my) kn_my = kn[g * M + id_my];

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

        // ---- issue all row gathers: step u fetches the rows of token slots u*RPL + r
        const uint16_t* kvg = kv + g * M * 2 * D + c * 8;
        u32x4 kreg[LPR], vreg[LPR];
#pragma unroll
        for (int u = 0; u < LPR; ++u) {
            const int id_u = __shfl(id_my, r * LPR + u);
            const bool valid_u = (jb + u * RPL + r) < nz;
            const u32x4 zero = {0u, 0u, 0u, 0u};
            kreg[u] = zero;
            vreg[u] = zero;
            if (valid_u) {
                const uint16_t* row = kvg + (int64_t)id_u * 2 * D;
                kreg[u] = *reinterpret_cast<const u32x4*>(row);
                vreg[u] = *reinterpret_cast<const u32x4*>(row + D);
            }
        }

        // ---- q . K partials, then reduce-scatter over the LPR lanes of a row group
        float part[LPR];
#pragma unroll
        for (int u = 0; u < LPR; ++u) {
            float a = 0.f;
            if (QBF16) {
                a = dot8_bf16(kreg[u], qpk);
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

        // ---- importance-sampling transform (transform_kernel, sparse_attention.cc:164-184)
        float z = -INFINITY;
        if (valid_my) {
            if (DENSE) {
                z = sc * inv_sqrt_d;
            } else {
                float cs = sc / (qnorm[h] * kn_my);
                cs = fminf(1.f, fmaxf(-1.f, cs));      // the reference does not clamp (NaN when
                                                       // a bf16-rounded norm makes cos > 1)
                const float theta = acosf(cs);
                const float proba = 1.f - theta * 0.31830988618379067f;
                const float p = powi(proba, K);
                const float qq = 1.f - p;
                const float w = 1.f - powi(qq, L - 1) * ((float)L * p + qq);
                z = sc * inv_sqrt_d - logf(w + 1e-4f);
            }
            if (score != nullptr) score[(int64_t)h * M + j_my] = z;
        }

        // ---- per-wave softmax
        const float m_w = wave_max(z);
        const float p_my = (valid_my && m_w > -INFINITY) ? __expf(z - m_w) : 0.f;
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
#pragma unroll
        for (int st = LPR; st < 64; st <<= 1)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], st);

        // ---- combine the 4 waves of the slice in LDS -> one partial
        if (lane < LPR) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_o[wave][c * 8 + i] = acc[i];
        }
        if (lane == 0) { s_m[wave] = m_w; s_l[wave] = l_w; }
        __syncthreads();
        if (tid < D) {
            float m = s_m[0];
#pragma unroll
            for (int w = 1; w < AT_WAVES; ++w) m = fmaxf(m, s_m[w]);
            float o = 0.f, l = 0.f;
#pragma unroll
            for (int w = 0; w < AT_WAVES; ++w) {
                const float e = (s_m[w] > -INFINITY) ? __expf(s_m[w] - m) : 0.f;
                o = fmaf(e, s_o[w][tid], o);
                l = fmaf(e, s_l[w], l);
            }
            part_o[(int64_t)s * D + tid] = o;
            if (tid == 0) part_ml[s] = make_float2(m, l);
        }
        __syncthreads();
    }
}

// Merge the slice partials of each head: out = sum_s e^{m_s-m} o_s / Z, bf16 (RNE);
// max_value_expsum[0][h] = m*log2e, [1][h] = log2 Z + m*log2e (softmax_kernel, .cc:238-239).
template <int D>
__global__ __launch_bounds__(D) void attn_merge_kernel(
    const float* __restrict__ part_o, const float2* __restrict__ part_ml,
    const int32_t* __restrict__ nnz, int BH, int64_t M, uint16_t* __restrict__ out,
    float* __restrict__ mve, float2* __restrict__ head_mz) {
    __shared__ int s_red[D / 64 + 1];
    const int h = blockIdx.x, tid = threadIdx.x;
    // slices before head h
    int before = 0;
    for (int hh = tid; hh < h; hh += D) before += slices_of(nnz[hh], M);
    before = (int)wave_sum((float)before);  // exact: counts < 2^24
    if ((tid & 63) == 0) s_red[tid >> 6] = before;
    __syncthreads();
    int pre = 0;
#pragma unroll
    for (int w = 0; w < D / 64; ++w) pre += s_red[w];
    const int ns = slices_of(nnz[h], M);
    float m = -INFINITY;
    for (int s = 0; s < ns; ++s) m = fmaxf(m, part_ml[pre + s].x);
    float o = 0.f, Z = 0.f;
    for (int s = 0; s < ns; ++s) {
        const float2 ml = part_ml[pre + s];
        const float e = (ml.x > -INFINITY) ? __expf(ml.x - m) : 0.f;
        o = fmaf(e, part_o[(int64_t)(pre + s) * D + tid], o);
        Z = fmaf(e, ml.y, Z);
    }
    const bool empty = !(Z > 0.f);
    out[(int64_t)h * D + tid] = empty ? (uint16_t)0 : f32_to_bf16_rne(o / Z);
    if (tid == 0) {
        const float mv = empty ? -INFINITY : m * 1.4426950408889634f;
        mve[h] = mv;
        mve[BH + h] = empty ? -INFINITY : (log2f(Z) + mv);
        head_mz[h] = make_float2(m, Z);
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

// ---------------------------------------------------------------- host launchers
int64_t attn_max_slices(int BH, int64_t M) {
    return (int64_t)BH * ((M + AT_SLICE - 1) / AT_SLICE + 1);
}

int attn_supported_head_dim(int D) { return D == 64 || D == 128; }

template <int D, bool DENSE, bool QBF16>
static hipError_t launch_partial_t(const uint16_t* kv, const float* kn, const void* q,
                                   const float* qn, const int32_t* ind, const int32_t* nnz,
                                   float* part_o, float2* part_ml, float* score, int BH, int G,
                                   int64_t M, int K, int L, int grid, hipStream_t st,
                                   hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds = (size_t)(BH + 1) * sizeof(int);
    if (ev0 != nullptr)   // per-dispatch begin/end timestamps (bench.py roofline leg)
        hipExtLaunchKernelGGL((attn_partial_kernel<D, DENSE, QBF16>), dim3(grid), dim3(AT_THREADS), lds,
                              st, ev0, ev1, 0, kv, kn, q, qn, ind, nnz, part_o, part_ml, score, BH, G,
                              M, K, L);
    else
        hipLaunchKernelGGL((attn_partial_kernel<D, DENSE, QBF16>), dim3(grid), dim3(AT_THREADS), lds,
                           st, kv, kn, q, qn, ind, nnz, part_o, part_ml, score, BH, G, M, K, L);
    return hipGetLastError();
}

hipError_t launch_attn_partial(int D, bool dense, bool qbf16, const uint16_t* kv, const float* kn,
                               const void* q, const float* qn, const int32_t* ind,
                               const int32_t* nnz, float* part_o, float2* part_ml, float* score,
                               int BH, int G, int64_t M, int K, int L, int grid, hipStream_t st,
                               hipEvent_t ev0, hipEvent_t ev1) {
#define MP_AT_CASE(DD, DE, QB)                                                                    \
    if (D == DD && dense == DE && qbf16 == QB)                                                    \
        return launch_partial_t<DD, DE, QB>(kv, kn, q, qn, ind, nnz, part_o, part_ml, score, BH, \
                                            G, M, K, L, grid, st, ev0, ev1);
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

hipError_t launch_attn_merge(int D, const float* part_o, const float2* part_ml, const int32_t* nnz,
                             int BH, int64_t M, uint16_t* out, float* mve, float2* head_mz,
                             hipStream_t st) {
    if (D == 128)
        hipLaunchKernelGGL((attn_merge_kernel<128>), dim3(BH), dim3(128), 0, st, part_o, part_ml,
                           nnz, BH, M, out, mve, head_mz);
    else if (D == 64)
        hipLaunchKernelGGL((attn_merge_kernel<64>), dim3(BH), dim3(64), 0, st, part_o, part_ml, nnz,
                           BH, M, out, mve, head_mz);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
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

hipError_t launch_merge_state(const uint16_t* va, const float* sa, const uint16_t* vb,
                              const float* sb, int R, int D, uint16_t* v, float* s, hipStream_t st) {
    hipLaunchKernelGGL(merge_state_kernel, dim3(R), dim3(128), 0, st, va, sa, vb, sb, R, D, v, s);
    return hipGetLastError();
}

}  // namespace mp
