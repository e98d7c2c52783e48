/*
 * mp_oracle.c -- CPU ORACLE for the MagicPIG LSH-sampled sparse decode attention path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may call it, and only as the checker / the timed
 * CPU baseline.  The product path (magicpig_amd/) never links or imports it.
 *
 * It is an independent plain-C restatement of the reference's algorithm; every function
 * cites the reference lines (relative to /root/reference) it follows.  It is pinned
 * (tests/test_oracle_golden.py) against golden vectors produced by the reference's own
 * compiled C++ (oracle/_ref, built by oracle/build_ref.py) and by the literal torch-CPU
 * restatement of models/attnserver.py:264-270 (tests/golden/make_golden.py).
 *
 * Conventions: bf16 values travel as uint16_t bit patterns.  h = b*H + head is the
 * request-major query-head index, g = h / G its kv-head unit (library/lsh/lsh.cc:251,
 * library/sparse_attention/sparse_attention.cc:773).  All offsets are 64-bit (the
 * reference's `int` offsets overflow at BASELINE cfg 3).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_LOG2E
#define M_LOG2E 1.4426950408889634074
#endif

/* ------------------------------------------------------------------ bf16 helpers */

static inline float bf16_to_f32(uint16_t h) {
    /* 3rdparty/FBGEMM/src/FbgemmBfloat16ConvertAvx512.cc:38-60: shift-left-16 */
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* torch semantics (round-to-nearest-even, NaN preserved) -- used wherever the
 * reference's value is produced by a torch op (models/attnserver.py:264-266). */
static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

/* FBGEMM FloatToBfloat16_avx512 (src/FbgemmBfloat16ConvertAvx512.cc:20-36):
 * add 0x8000 then truncate = round-half-up on the magnitude. Used for `output`. */
static inline uint16_t f32_to_bf16_rhu(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x8000u;
    return (uint16_t)(u >> 16);
}

int mpo_version(void) { return 1; }

int mpo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void mpo_f32_to_bf16_rne(const float* src, uint16_t* dst, int64_t n) {
    for (int64_t i = 0; i < n; ++i) dst[i] = f32_to_bf16_rne(src[i]);
}

void mpo_bf16_to_f32(const uint16_t* src, float* dst, int64_t n) {
    for (int64_t i = 0; i < n; ++i) dst[i] = bf16_to_f32(src[i]);
}

/* ------------------------------------------------------------------ a-1 SimHash */

/* Exact sign of sum_d a[d]*w[d*ldw] for bf16 a, w: every product of two bf16 values is
 * exact in binary64 and the 128-term sum is accumulated in binary64, so the sign is the
 * exact sign unless |sum| < ~1e-16 * sum|terms| (SURVEY.md 7 "Bit-exact hash codes"). */
static inline int exact_sign_bit(const uint16_t* a, const uint16_t* w, int D, int64_t ldw) {
    double acc = 0.0;
    for (int d = 0; d < D; ++d)
        acc += (double)bf16_to_f32(a[d]) * (double)bf16_to_f32(w[(int64_t)d * ldw]);
    return acc > 0.0; /* `.gt(0)`: 0 and NaN map to bit 0 (attnserver.py:267) */
}

/*
 * Query SimHash, models/attnserver.py:264-270:
 *   norm_q = q / q.norm(p=2, dim=-1, keepdim=True)         (bf16 tensors)
 *   bits   = matmul(norm_q, hash_func).gt(0)               (bf16 x bf16, f32 accumulate)
 *   code[h,l] = sum_{i<K} bits[h, l*K+i] * 2^i             (fp16 mv with binary_pack, :56-57,268-269)
 * torch computes the bf16 norm with an f32 accumulator and rounds to bf16; the division is
 * done in f32 and rounded to bf16.  The accumulation ORDER inside torch's reduction / GEMM is
 * unspecified, so the oracle fixes the order-independent definition: exact sum of squares ->
 * f32 -> sqrtf -> bf16; exact sign of the bf16 x bf16 dot product.
 * q: bf16 [R, D]; W = hash_func: bf16 [D, K*L] row-major; codes: int32 [R, L];
 * qnorm (optional): f32 [R] = ||q||_2 of the bf16 query in f32 (attnserver.py:300,
 * `pinned_query.float().norm(p=2, dim=-1)`).
 */
void mpo_simhash_query(const uint16_t* q, const uint16_t* W, int R, int D, int K, int L,
                       int32_t* codes, float* qnorm) {
    const int64_t KL = (int64_t)K * L;
    uint16_t* nq = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)D);
    for (int r = 0; r < R; ++r) {
        const uint16_t* qr = q + (int64_t)r * D;
        double ss = 0.0;
        for (int d = 0; d < D; ++d) {
            double x = (double)bf16_to_f32(qr[d]);
            ss += x * x;
        }
        float nrm = sqrtf((float)ss);
        if (qnorm) qnorm[r] = nrm;
        float nb = bf16_to_f32(f32_to_bf16_rne(nrm));
        for (int d = 0; d < D; ++d) nq[d] = f32_to_bf16_rne(bf16_to_f32(qr[d]) / nb);
        for (int l = 0; l < L; ++l) {
            int32_t code = 0;
            for (int i = 0; i < K; ++i)
                code |= exact_sign_bit(nq, W + (int64_t)l * K + i, D, KL) << i;
            codes[(int64_t)r * L + l] = code;
        }
    }
    free(nq);
}

/*
 * Key SimHash at prefill, models/attnserver.py:159-168: same projection on the centred
 * keys WITHOUT normalisation; codes are stored int16 and transposed to [L, n] per kv head.
 * keys: bf16 [n, D] (one kv head); codes: int16 [L, n].
 */
void mpo_simhash_keys(const uint16_t* keys, const uint16_t* W, int64_t n, int D, int K, int L,
                      int16_t* codes) {
    const int64_t KL = (int64_t)K * L;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n; ++t) {
        const uint16_t* kt = keys + t * D;
        for (int l = 0; l < L; ++l) {
            int code = 0;
            for (int i = 0; i < K; ++i)
                code |= exact_sign_bit(kt, W + (int64_t)l * K + i, D, KL) << i;
            codes[(int64_t)l * n + t] = (int16_t)code;
        }
    }
}

/* ------------------------------------------------------------------ a-4/a-5 tables */

/*
 * LSH::fill, library/lsh/lsh.cc:143-201, for one (layer, request):
 *   sorted_codes int16 [Hkv, L, n], sorted_ids int32 [Hkv, L, n]
 *   table_start/table_end int32 [Hkv, L, NB] (this request's slice; must be zero on entry,
 *   the reference uses end==0 as "unseen", lsh.cc:177-185)
 *   table int32 [Hkv, L, M] (row stride M, first n entries written, lsh.cc:196-200)
 */
void mpo_lsh_fill(const int16_t* sorted_codes, const int32_t* sorted_ids, int Hkv, int L,
                  int64_t n, int NB, int64_t M, int32_t* table_start, int32_t* table_end,
                  int32_t* table) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int j = 0; j < L; ++j) {
        for (int i = 0; i < Hkv; ++i) {
            const int16_t* v = sorted_codes + ((int64_t)i * L + j) * n;
            int32_t* ms = table_start + ((int64_t)i * L + j) * NB;
            int32_t* me = table_end + ((int64_t)i * L + j) * NB;
            for (int64_t k = 0; k < n; ++k) {
                const int c = (int)v[k];
                if (me[c] == 0) {
                    ms[c] = (int32_t)k;
                    me[c] = (int32_t)(k + 1);
                } else {
                    me[c] = me[c] + 1;
                }
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)Hkv * L; ++r)
        memcpy(table + r * M, sorted_ids + r * n, sizeof(int32_t) * (size_t)n);
}

/* ------------------------------------------------------------------ a-2/a-3 retrieve */

/*
 * LSH::retrieve, library/lsh/lsh.cc:243-288 (one query head).  Returns nnz; results get the
 * selected token ids in the reference's second-hit order; mask gets min(count, 2).
 * start/end/table point at the head's kv-group slice ([L,NB], [L,NB], [L,M]).
 */
static int64_t retrieve_head(const int32_t* start, const int32_t* end, const int32_t* table,
                             const int32_t* q, int L, int NB, int64_t M, uint8_t* mask,
                             int32_t* result) {
    memset(mask, 0, (size_t)M);
    int32_t* out = result;
    for (int i = 0; i < L; ++i) {
        const int qi = q[i];
        const int32_t s = start[(int64_t)i * NB + qi];
        const int32_t e = end[(int64_t)i * NB + qi];
        const int32_t* content = table + (int64_t)i * M;
        for (int32_t j = s; j < e; ++j) {
            const int32_t idx = content[j];
            const uint8_t mv = mask[idx];
            if (mv == 0) {
                mask[idx] = 1;
            } else if (mv == 1) {
                mask[idx] = 2;
                *out++ = idx;
            }
        }
    }
    return (int64_t)(out - result);
}

/*
 * LSH::batch_retrieve, library/lsh/lsh.cc:210-241: all BH = B*H heads of one layer.
 *   table_start/table_end: int32 [B*Hkv, L, NB]; table: int32 [B*Hkv, L, M]
 *   query: int32 [BH, L]; results: int32 [BH, M] (first nnz valid, rest untouched);
 *   nnz: int32 [BH]; mask: uint8 [BH, M] scratch (== get_mask(), lsh.cc:308-314)
 *   nthreads <= 0: OpenMP default.  Schedule static,1 as lsh.cc:229.
 */
void mpo_lsh_batch_retrieve(const int32_t* table_start, const int32_t* table_end,
                            const int32_t* table, const int32_t* query, int BH, int G, int L,
                            int NB, int64_t M, int32_t* results, int32_t* nnz, uint8_t* mask,
                            int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
    for (int h = 0; h < BH; ++h) {
        const int64_t g = h / G;
        nnz[h] = (int32_t)retrieve_head(table_start + g * L * NB, table_end + g * L * NB,
                                         table + g * L * M, query + (int64_t)h * L, L, NB, M,
                                         mask + (int64_t)h * M, results + (int64_t)h * M);
    }
}

/* ------------------------------------------------------------------ a-8..a-11 attention */

/* avx512_exp_ps, library/sparse_attention/sparse_attention.cc:17-36, one lane, same
 * operation order (truncating split, cubic polynomial of 2^f, exponent bit-stuffing). */
static inline float ref_poly_exp(float x) {
    float scaled = x * 1.44269504089f;
    int32_t ip = (int32_t)scaled; /* cvttps: truncate toward zero */
    float frac = scaled - (float)ip;
    uint32_t eb = (uint32_t)(ip + 127) << 23;
    float int_exp;
    memcpy(&int_exp, &eb, 4);
    float poly = 0.05550410866f;
    poly = fmaf(poly, frac, 0.2402265069f);
    poly = fmaf(poly, frac, 0.6931471806f);
    poly = fmaf(poly, frac, 1.0000000000f);
    return int_exp * poly;
}

/*
 * One head of the sparse attention:
 *   qk_kernel / qk_kernel_bf16_impl   sparse_attention.cc:38-67 / 69-103
 *   transform_kernel                  sparse_attention.cc:164-184
 *   softmax_kernel                    sparse_attention.cc:186-240
 *   wv_kernel                         sparse_attention.cc:321-347
 * key/value: bf16 [M, D] of the head's kv group; kn: f32 [M]; q: f32 [D] (already widened);
 * ind: int32 [>= nnz]; score: f32 scratch [>= nnz] -> probabilities (== get_score());
 * out: bf16 [D]; mv/es: scalars of max_value_expsum rows 0/1.
 * exp_mode bit 0 clear: exact expf everywhere (the oracle proper).
 * exp_mode bit 0 set  : the reference's polynomial exp on the first 16*floor(nnz/16) elements and
 *             expf on the tail (sparse_attention.cc:200-222) -- used only to pin this restatement
 *             tightly against oracle/_ref.
 * exp_mode bit 1 set  : the importance weight w is evaluated in binary64 without cancellation
 *             (w = -expm1((L-1) log1p(-p) + log1p((L-1) p))) instead of the reference's literal
 *             f32 expression (.cc:176-181), whose subtraction from 1 carries ~1e-7 absolute noise,
 *             i.e. up to ~1e-3 relative noise in w + 1e-4.  Used to separate that inherent noise of
 *             the reference formula from kernel error in the parity tests.
 * clamp_cos 1: clamp cos to [-1,1] before acosf (the reference does not, .cc:177: NaN when
 *             a bf16-rounded norm makes cos > 1).  The HIP path clamps.
 */
static void sparse_attention_head(const uint16_t* key, const uint16_t* value, const float* kn,
                                  const float* q, float qn, const int32_t* ind, int64_t nnz,
                                  int D, int K, int L, int exp_mode, int clamp_cos,
                                  float* score, uint16_t* out, float* mv, float* es) {
    const float sqrt_dim = sqrtf((float)D);
    /* qk: f32 accumulate of q[d] * K[ind[j]][d] */
    for (int64_t j = 0; j < nnz; ++j) {
        const uint16_t* kr = key + (int64_t)ind[j] * D;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += q[d] * bf16_to_f32(kr[d]);
        score[j] = acc;
    }
    /* transform: importance-sampling correction, P[>= 2 of L tables collide] */
    for (int64_t j = 0; j < nnz; ++j) {
        const float norm = qn * kn[ind[j]];
        float c = score[j] / norm;
        if (clamp_cos) c = fminf(1.f, fmaxf(-1.f, c));
        const float theta = acosf(c);
        const float proba = (float)(1 - theta / M_PI);
        if (exp_mode & 2) {
            const double pd = pow((double)proba, (double)K);
            const double wd = -expm1((L - 1) * log1p(-pd) + log1p((L - 1) * pd));
            score[j] = (float)((double)score[j] / sqrt((double)D) - log(wd + 1e-4));
            continue;
        }
        const float p = powf(proba, (float)K);
        const float qq = 1 - p;
        const float w = 1 - powf(qq, (float)(L - 1)) * (L * p + qq);
        score[j] = score[j] / sqrt_dim - logf(w + 1e-4);
    }
    if (nnz <= 0) {
        /* measured on the compiled reference: out = 0, LSE = -inf */
        for (int d = 0; d < D; ++d) out[d] = 0;
        *mv = -INFINITY;
        *es = -INFINITY;
        return;
    }
    /* softmax */
    float m = score[0];
    for (int64_t j = 1; j < nnz; ++j) m = fmaxf(m, score[j]);
    float sum = 0.f;
    const int64_t vec_end = (exp_mode & 1) ? (nnz / 16) * 16 : 0;
    if (exp_mode & 1) {
        /* 16 partial sums, lane i accumulates elements i, i+16, ... (.cc:200-214) */
        float lanes[16];
        for (int i = 0; i < 16; ++i) lanes[i] = 0.f;
        for (int64_t j = 0; j < vec_end; ++j) {
            score[j] = ref_poly_exp(score[j] - m);
            lanes[j & 15] += score[j];
        }
        for (int i = 0; i < 16; ++i) sum += lanes[i];
    }
    for (int64_t j = vec_end; j < nnz; ++j) {
        score[j] = expf(score[j] - m);
        sum += score[j];
    }
    for (int64_t j = 0; j < nnz; ++j) score[j] /= sum;
    *mv = (float)(m * M_LOG2E);
    *es = log2f(sum) + *mv;
    /* wv: f32 accumulate, bf16 store with FBGEMM rounding */
    for (int d0 = 0; d0 < D; d0 += 16) {
        float acc[16];
        const int w = (D - d0 < 16) ? (D - d0) : 16;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int64_t j = 0; j < nnz; ++j) {
            const uint16_t* vr = value + (int64_t)ind[j] * D + d0;
            const float pj = score[j];
            for (int i = 0; i < w; ++i) acc[i] = fmaf(bf16_to_f32(vr[i]), pj, acc[i]);
        }
        for (int i = 0; i < w; ++i) out[d0 + i] = f32_to_bf16_rhu(acc[i]);
    }
}

/*
 * SparseAttentionServer::attention_wrapper -> dynamic_attention{,_bf16},
 * sparse_attention.cc:629-745, 748-865: all BH heads of one layer, one head per OpenMP
 * iteration, schedule(dynamic,1).
 *   key/value: bf16 [B*Hkv, M, D]; key_norm: f32 [B*Hkv, M]
 *   query: bf16 [BH, D] (query_is_bf16=1, the __AVX512BF16__ build reads bf16, .cc:820)
 *          or f32 [BH, D] (query_is_bf16=0, `query_pt.to(kFloat32)`, .cc:761)
 *   query_norm f32 [BH]; ind int32 [BH, M]; nnz int32 [BH]
 *   output bf16 [BH, D]; max_value_expsum f32 [2, BH]; score f32 [BH, M] scratch/probabilities
 */
void mpo_sparse_attention(const uint16_t* key, const uint16_t* value, const float* key_norm,
                          const void* query, int query_is_bf16, const float* query_norm,
                          const int32_t* ind, const int32_t* nnz, int BH, int G, int D,
                          int64_t M, int K, int L, int exp_mode, int clamp_cos,
                          uint16_t* output, float* max_value_expsum, float* score,
                          int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int h = 0; h < BH; ++h) {
        const int64_t g = h / G;
        float qf[1024];
        for (int d = 0; d < D; ++d)
            qf[d] = query_is_bf16 ? bf16_to_f32(((const uint16_t*)query)[(int64_t)h * D + d])
                                  : ((const float*)query)[(int64_t)h * D + d];
        sparse_attention_head(key + g * M * D, value + g * M * D, key_norm + g * M, qf,
                              query_norm[h], ind + (int64_t)h * M, nnz[h], D, K, L, exp_mode,
                              clamp_cos, score + (int64_t)h * M, output + (int64_t)h * D,
                              max_value_expsum + h, max_value_expsum + BH + h);
    }
}

/* ------------------------------------------------------------------ a-15 full attention */

/*
 * SparseAttentionServer::full_attention, sparse_attention.cc:988-1037 (+ qk_kernel_full
 * :106-160, softmax_kernel_optimized :242-286, wv_kernel_dim128_full :386-451): dense
 * decode attention over rows [0, nnz) of each kv head for its G query heads.  The reference
 * indexes nnz by kv-head loop index for QK/PV and by head for softmax (.cc:1010,1018,1032);
 * callers fill all entries equal (models/attnserver.py:470) and so does every test, so the
 * oracle uses nnz[h].
 *
 * PINNED against the compiled reference by tests/golden/full_dense.npz (G in {1, 4, 8},
 * nnz in {0, 1, 15, 16, 63, 64, 65, ...}).  `quirks` reproduces what the reference does beyond
 * the definition, so that the pin also holds where the two differ:
 *   bit 0: exp by the reference's polynomial (avx512_exp_ps, as exp_mode bit 0 of the sparse path);
 *   bit 1: softmax_kernel_optimized walks round_up(nnz, 16) score slots with a full 16-lane mask
 *          (:249-283: `i < nnz ? 0xFFFF : ...` is always true at a block start), i.e. it also counts
 *          the up-to-15 slots behind the list with whatever the score buffer holds there (zeros on
 *          a fresh server, .cc:579-580) in max and sum, and overwrites them; P.V uses nnz rows.
 * quirks = 0 is the definition (exact exp, exactly nnz rows) the HIP path is checked against.
 */
void mpo_full_attention(const uint16_t* key, const uint16_t* value, const float* query,
                        const int32_t* nnz, int BH, int G, int D, int64_t M, uint16_t* output,
                        float* max_value_expsum, float* score, int quirks, int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#endif
    const float scale = 1.0f / sqrtf((float)D);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int h = 0; h < BH; ++h) {
        const int64_t g = h / G;
        const uint16_t* kk = key + g * M * D;
        const uint16_t* vv = value + g * M * D;
        const float* q = query + (int64_t)h * D;
        float* s = score + (int64_t)h * M;
        uint16_t* out = output + (int64_t)h * D;
        const int64_t n = nnz[h];
        if (n <= 0) {
            for (int d = 0; d < D; ++d) out[d] = 0;
            max_value_expsum[h] = -INFINITY;
            max_value_expsum[BH + h] = -INFINITY;
            continue;
        }
        int64_t ns = n;                                  /* slots the softmax covers */
        if (quirks & 2) {
            ns = (n + 15) & ~(int64_t)15;
            if (ns > M) ns = M;
        }
        float m = -INFINITY;
        for (int64_t j = 0; j < ns; ++j) {
            if (j < n) {
                float acc = 0.f;
                for (int d = 0; d < D; ++d) acc += q[d] * bf16_to_f32(kk[j * D + d]);
                s[j] = acc;
            }                                            /* else: stale slot, raw score as found */
            s[j] *= scale;
            m = fmaxf(m, s[j]);
        }
        float sum = 0.f;
        for (int64_t j = 0; j < ns; ++j) {
            s[j] = (quirks & 1) ? ref_poly_exp(s[j] - m) : expf(s[j] - m);
            sum += s[j];
        }
        for (int64_t j = 0; j < ns; ++j) s[j] /= sum;
        max_value_expsum[h] = (float)(m * M_LOG2E);
        max_value_expsum[BH + h] = log2f(sum) + max_value_expsum[h];
        for (int d = 0; d < D; ++d) {
            float acc = 0.f;
            for (int64_t j = 0; j < n; ++j) acc = fmaf(bf16_to_f32(vv[j * D + d]), s[j], acc);
            out[d] = f32_to_bf16_rhu(acc);
        }
    }
}

/* ------------------------------------------------------------------ a-13 LSE merge */

/*
 * flashinfer.merge_state as used at models/attnserver.py:308; FlashInfer is a third-party
 * dependency absent from /root/reference (un-vendored, unpinned wheel; install.sh:4).  Its
 * published definition on base-2 LSEs (the ones run_return_lse returns and the reference converts
 * its own to, evaluations/RULER/pred/attnserver_dist.py:848-849) is
 *   s = log2(2^a + 2^b);  v = (2^a v_a + 2^b v_b) / 2^s
 * PINNED by tests/golden/window_merge.npz: a committed torch-CPU statement of the reference's call
 * site (attnserver_dist.py:813-851, 882; models/attnserver.py:293-308) including the property that
 * defines the operator -- merge(window part, sampled part) equals ONE softmax attention over the
 * union of the two token sets (tests/golden/make_golden.py: run_window_merge).
 * va, vb: bf16 [R, D]; sa, sb: f32 [R]; v: bf16 [R, D] (RNE); s: f32 [R].
 */
void mpo_merge_state(const uint16_t* va, const float* sa, const uint16_t* vb, const float* sb,
                     int R, int D, uint16_t* v, float* s) {
    for (int r = 0; r < R; ++r) {
        const float a = sa[r], b = sb[r];
        const float mx = fmaxf(a, b);
        float wa, wb, lse;
        if (mx == -INFINITY) {
            wa = 0.f; wb = 0.f; lse = -INFINITY;
        } else {
            const float ea = exp2f(a - mx), eb = exp2f(b - mx);
            wa = ea / (ea + eb);
            wb = eb / (ea + eb);
            lse = mx + log2f(ea + eb);
        }
        for (int d = 0; d < D; ++d) {
            const float x = wa * bf16_to_f32(va[(int64_t)r * D + d]) +
                            wb * bf16_to_f32(vb[(int64_t)r * D + d]);
            v[(int64_t)r * D + d] = f32_to_bf16_rne(x);
        }
        if (s) s[r] = lse;
    }
}
