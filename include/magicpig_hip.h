/*
 * magicpig_hip.h -- C ABI of the MI355X-native (gfx950) implementation of MagicPIG's
 * LSH-sampled sparse decode attention.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Each entry point
 * names the reference interface it replaces (paths relative to the MagicPIG repository).
 * The reference exposes this path as two pybind11 classes (library/lsh/lsh.cc:316-326,
 * library/sparse_attention/sparse_attention.cc:1243-1263) plus four lines of torch
 * (models/attnserver.py:264-270); INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *   - every function returns MP_OK (0) or an MP_ERR_* code; mp_last_error() gives the text
 *     (an ADDITION over the reference, which has no error reporting at all: lsh.cc asserts
 *     are compiled out by -DNDEBUG and wrong shapes corrupt memory);
 *   - `mem` says where caller buffers live: MP_MEM_HOST buffers (the reference's CPU-tensor callers,
 *     models/attnserver.py:59-66) are used IN PLACE by the kernels where they are pinned; for a
 *     pageable buffer the kernels work on a pinned mirror owned by the handle and the host copies the
 *     live entries across (the library never registers a caller's memory);
 *     MP_MEM_DEVICE buffers are used in place (fast path: codes, results and nnz never leave HBM);
 *   - `stream` is a hipStream_t (NULL = default stream); work is enqueued asynchronously for
 *     MP_MEM_DEVICE arguments, synchronously completed for MP_MEM_HOST arguments;
 *   - all state (tables, KV, norms, scratch) lives in HBM of the device that was current at
 *     alloc time (mp_simhash_set_planes for the hasher) and is owned by the handle, as the
 *     reference objects own theirs (lsh.cc:29-42, sparse_attention.cc:529-544); every call on a
 *     handle switches to that device for its duration and restores the caller's current device,
 *     `stream` must be a stream of the handle's device, and the handles of one
 *     mp_decode_* call must live on the same device (MP_ERR_INVALID otherwise);
 *   - bf16 travels as uint16_t; h = b*H + head is the request-major query-head index,
 *     g = h / (H/Hkv) its kv-head unit (lsh.cc:251, sparse_attention.cc:773);
 *   - one call at a time per handle (the reference's scratch is per object too,
 *     sparse_attention.cc:577).
 */
#ifndef MAGICPIG_HIP_H
#define MAGICPIG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_OK 0
#define MP_ERR_INVALID 1     /* bad argument (shape, range, null pointer) */
#define MP_ERR_STATE 2       /* handle not allocated / already allocated */
#define MP_ERR_HIP 3         /* HIP runtime error, text in mp_last_error() */
#define MP_ERR_NOMEM 4       /* hipMalloc failed */
#define MP_ERR_UNSUPPORTED 5 /* configuration outside the built kernels (e.g. head_dim) */
#define MP_ERR_DATA 6        /* device-side validation failed (unsorted codes, id out of range) */

#define MP_MEM_HOST 0
#define MP_MEM_DEVICE 1

#define MP_DTYPE_BF16 0
#define MP_DTYPE_F32 1

typedef struct mp_lsh mp_lsh_t;         /* replaces class LSH                    (library/lsh/lsh.h:14-43) */
typedef struct mp_attn mp_attn_t;       /* replaces class SparseAttentionServer  (library/sparse_attention/sparse_attention.h:14-52) */
typedef struct mp_simhash mp_simhash_t; /* replaces hash_func + binary_pack      (models/attnserver.py:55-57) */
typedef void* mp_stream_t;              /* hipStream_t */

/* ---------------------------------------------------------------- library */
int mp_version(void);
const char* mp_last_error(void);
/* name of the gfx target the kernels were compiled for ("gfx950") */
const char* mp_arch(void);

/* ---------------------------------------------------------------- query / key SimHash
 * Replaces models/attnserver.py:55-57 (hash_func [D, K*L] bf16, binary_pack) and :264-270
 * (q / ||q|| -> matmul -> gt(0) -> K-bit pack).  MFMA kernel, codes bit-exact. */
int mp_simhash_create(mp_simhash_t** out);
int mp_simhash_destroy(mp_simhash_t* s);
/* hash_func: bf16 [D, K*L] row-major, exactly the tensor of attnserver.py:55. */
int mp_simhash_set_planes(mp_simhash_t* s, int D, int K, int L, const uint16_t* hash_func,
                          int mem, mp_stream_t stream);
/* q: bf16 [R, D]; codes: int32 [R, L] (== q_hashcode, attnserver.py:270);
 * qnorm: f32 [R] or NULL (== pinned_query.float().norm(p=2, dim=-1), attnserver.py:300).
 * ||q|| is exact (f32 sqrt of the f32-rounded exact sum of squares) from the MFMA kernel (R > 64); the fused prologue of
 * the one-launch decode and the R <= 64 form compute it from an f32 sum and a Newton-refined v_sqrt: within 2 ulps of the
 * exact value, identical between those two, NOT bit-identical with the MFMA kernel's.  The CODES are identical everywhere
 * (the fast form falls back to the exact sequence wherever a bf16 rounding could differ). */
int mp_simhash_query(mp_simhash_t* s, const uint16_t* q, int R, int32_t* codes, float* qnorm,
                     int mem, mp_stream_t stream);
/* keys: bf16 [Hkv, n, D] (centred keys); codes: int16 [Hkv, L, n] (== hash_code_buffer[:, :, :n],
 * attnserver.py:159-168: no normalisation, transposed, int16).
 * Key and query codes are the sign of the EXACT dot product (f32 accumulation + exact recomputation inside a guard
 * band).  The reference's bf16 GEMM decides a bit by its summation order where the exact value is zero or within
 * rounding of it: ~1e-8 of the bits can differ from torch's (the fixtures list them, tests/golden: kcodes_ties), so
 * hash the keys AND the queries with this library. */
int mp_simhash_keys(mp_simhash_t* s, const uint16_t* keys, int Hkv, int64_t n, int16_t* codes,
                    int mem, mp_stream_t stream);

/* ---------------------------------------------------------------- LSH tables + retrieve */
int mp_lsh_create(mp_lsh_t** out);                    /* LSH::LSH()            lsh.cc:25-27 */
int mp_lsh_destroy(mp_lsh_t* h);                      /* LSH::~LSH()           lsh.cc:29-42 */
/* LSH::alloc, lsh.cc:44-91 (same argument order). */
int mp_lsh_alloc(mp_lsh_t* h, int K, int L, int num_layers, int num_attention_heads,
                 int num_key_value_heads, int batch_size, int max_length);
/* LSH::alloc with the two decisions mp_lsh_alloc takes by itself stated by the caller (the reference's alloc takes exactly
 * what its arguments say, lsh.cc:44-91; mp_lsh_alloc = mp_lsh_alloc_ex(..., -1, 0)):
 *   accel_budget_bytes  HBM this handle may spend, over ALL its layers, on structures that only make the decode faster: the
 *                       direct piece slots (1.26 GB per layer at cfg 1 for -1.4 us per launch) and the HBM copy of
 *                       the rows a MP_MEM_HOST batch_retrieve hands out (4 B x B*H x max_length).  0 = none of them; < 0 =
 *                       mp_lsh_alloc's rule: each is taken where it pays and needs less than a third of the HBM that is FREE
 *                       when it is allocated -- which makes layout and speed depend on what else the process has allocated
 *                       by then; a serving process that loads its weights later should state a figure.  The tables and
 *                       bounds are not accelerators and are always allocated.
 *   ranges              token ranges per table row = workgroups per query head of the one-launch decode: 0 = by B*H, the
 *                       CU count and the tokens per range (mp_lsh_alloc), else 1, 2, 4, 8, 16 or 32.
 * The process-wide debug options "decode_cluster" / "decode_direct" / "decode_slot_log2" still override both (A/B runs). */
int mp_lsh_alloc_ex(mp_lsh_t* h, int K, int L, int num_layers, int num_attention_heads,
                    int num_key_value_heads, int batch_size, int max_length, int64_t accel_budget_bytes, int ranges);
/* LSH::fill, lsh.cc:143-201.  sorted_codes int16 [Hkv, L, n], sorted_ids int32 [Hkv, L, n].
 * Unlike the reference the slot need not be cleared first (rows are re-zeroed on device). */
int mp_lsh_fill(mp_lsh_t* h, int layer_id, int request_id, const int16_t* sorted_codes,
                const int32_t* sorted_ids, int64_t n, int mem, mp_stream_t stream);
/* Working replacement of the reference's half-written LSH::fastfill (lsh.cc:93-142): builds
 * the tables on device from UNSORTED codes int16 [Hkv, L, n] (counting sort; ascending token
 * ids inside every bucket; the sub-bounds of the R token ranges are written by the same kernel).
 * Inside a 64-token group the tokens of a bucket are ranked by the order in which the LDS serves the lanes of one
 * returning atomic -- lane order on gfx950 -- and every bucket run the kernel writes is checked to ascend; a request whose
 * check fails is rebuilt with the exact (match-any) ranking before the call returns ("build_rank_exact" forces that
 * form, "build_rank_fallbacks" counts rebuilds: mp_debug_set_option / _get_option).  The result is the stable sort either way. */
int mp_lsh_build(mp_lsh_t* h, int layer_id, int request_id, const int16_t* codes, int64_t n,
                 int mem, mp_stream_t stream);
/* mp_lsh_build for a caller that has ALREADY filled the attention store of (layer_id, request_id) -- the reference's
 * order: key norms and K/V at models/attnserver.py:146, 174, the tables at :178-193.  The sort then writes every table
 * word as  id | bf16 key norm << 17  itself (see mp_lsh_get_id_bits) and the direct slots are built once from the
 * packed words, so the first mp_decode_* call of the layer has nothing left to pack.  Same tables, same results as
 * mp_lsh_build; falls back to plain ids where the payload does not apply (an id >= 2^17 in the layer, norms changed by
 * an append, K >= 14).  `attn` must agree with `h` on device, batch size, kv heads and max_length. */
int mp_lsh_build_with_norms(mp_lsh_t* h, mp_attn_t* attn, int layer_id, int request_id, const int16_t* codes,
                            int64_t n, int mem, mp_stream_t stream);
/* LSH::batch_retrieve, lsh.cc:210-288.  query int32 [B*H, L]; results int32 [B*H, M] (first
 * nnz[h] entries valid, ASCENDING token ids; the rest untouched); nnz int32 [B*H]. */
int mp_lsh_batch_retrieve(mp_lsh_t* h, int layer_id, const int32_t* query, int32_t* results,
                          int32_t* nnz, int mem, mp_stream_t stream);
int mp_lsh_clear(mp_lsh_t* h, mp_stream_t stream);    /* LSH::clear            lsh.cc:293-306 */
/* LSH::get_mask, lsh.cc:308-314: int8 [B, H, M] collision counters min(count, 2) of the LAST
 * batch_retrieve call (recomputed on demand; the hot path keeps only bitmaps in LDS). */
int mp_lsh_get_mask(mp_lsh_t* h, int8_t* mask, int mem, mp_stream_t stream);
/* debug views (the reference declares get_table*, lsh.h:24-26, but never defines them):
 * bounds int32 [B*Hkv, L, NB, R+1]: entry 0 = start and entry R = end of a bucket inside its table row (the
 * reference's table_start / table_end, lsh.h:38-39), entry r = first position of the bucket whose token id is
 * >= r * range_len (ids ascend inside a bucket when R > 1); table int32 [B*Hkv, L, M].
 * R (1, 2, 4, 8, 16 or 32) is the number of token ranges a table row is cut into = the number of workgroups that serve
 * one query head in mp_decode_sparse_layer, chosen at alloc from B*H and the device's CU count. */
int mp_lsh_get_tables(mp_lsh_t* h, int layer_id, void** bounds_dev, void** table_dev);
int mp_lsh_get_ranges(mp_lsh_t* h, int* ranges, int* range_len);
/* HBM the handle holds PER LAYER, bytes: [0] bounds, [1] table, [2] direct piece slots (0 where the handle keeps none),
 * [3] bytes of one slot (128, 64, 32 or 0).  The reference's tables are [1] alone (lsh.cc:44-91: table + table_start /
 * table_end); bounds with R + 1 entries and the slots are this implementation's accelerators. */
int mp_lsh_get_footprint(mp_lsh_t* h, int64_t* bytes4);
/* The same plus what the host-buffer mode and the budget of mp_lsh_alloc_ex add, bytes8: [0..3] as above (PER LAYER);
 * [4] HBM copy of the rows a MP_MEM_HOST batch_retrieve hands out (whole handle; 0 until the first such call, or when the
 * budget refused it); [5] pinned host memory the handle holds for that mode; [6] the accelerator budget (< 0: "a third of
 * what is free"); [7] accelerator bytes in use: all layers' slots + [4]. */
int mp_lsh_get_footprint_ex(mp_lsh_t* h, int64_t* bytes8);
/* Width of the id field of the layer's table words: 17 while every token id the layer's tables hold is below 2^17
 * (any max_length), 0 (plain ids) from the first mp_lsh_fill / mp_lsh_build that brings a wider one until mp_lsh_clear.
 * Where it is 17, a table word is  token id | (payload << 17): the one-launch decode entries let the entries carry
 * their tokens' key norms (bits 14..0 of the bf16 norm the attention store holds: what models/attnserver.py:143
 * stores) the first time a layer is decoded after its tables or the store's norms changed -- two extra kernels per
 * request, once, never under stream capture -- so that a selected token's norm is found on chip instead of costing one
 * random HBM access (one line request in five of the gather).  A norm that is not a non-negative bf16 number keeps its
 * KV group on the per-token reads.  Every retrieve masks the ids; results are plain token ids; mp_lsh_get_tables'
 * `table` holds the words as they are (mask with (1 << id_bits) - 1).  mp_lsh_fill / mp_lsh_build write plain ids. */
int mp_lsh_get_id_bits(mp_lsh_t* h, int layer_id, int* id_bits);

/* ---------------------------------------------------------------- sparse attention */
int mp_attn_create(mp_attn_t** out);                  /* sparse_attention.cc:519-527 */
int mp_attn_destroy(mp_attn_t* h);                    /* sparse_attention.cc:529-544 */
/* SparseAttentionServer::alloc, sparse_attention.cc:546-583 (same argument order).  max_length x 4 x head_dim bytes must
 * fit a 32-bit row offset (2^23 tokens at head_dim 128, 2^24 at 64); a store that pairs with an LSH handle in
 * mp_decode_* shares that handle's max_length (<= 2^22). */
int mp_attn_alloc(mp_attn_t* h, int num_layers, int num_attention_heads,
                  int num_key_value_heads, int head_dim, int batch_size, int max_length);
/* SparseAttentionServer::fill, sparse_attention.cc:601-627.  k, v bf16 [Hkv, n, D];
 * kn f32 [Hkv, n]. */
int mp_attn_fill(mp_attn_t* h, int layer_id, int request_id, const uint16_t* k,
                 const uint16_t* v, const float* kn, int64_t n, int mem, mp_stream_t stream);
/* The sparse-layer branch of LSHSparseAttnServer.fill (models/attnserver.py:126-175) for one request, on device
 * buffers: key_cache / value_cache bf16 [seq_len, Hkv, D] (token-major, as the model's KV cache holds them).  The
 * offloaded tokens [num_sink, seq_len - num_local): avg_k = their per-(kv head, dim) mean (bf16 [Hkv, D], written
 * to `avg_k`, :142), keys centred with it (:145), key norms (:146), K|V rows stored (SparseAttentionServer::fill,
 * :174) and, when `codes` is not NULL, the key SimHash of the centred keys (:159-168; int16 [Hkv, L, n], n =
 * seq_len - num_sink - num_local; `s` = the hasher) -- three passes over the KV cache instead of torch's mean / sub /
 * norm / transpose().contiguous() kernels.  Sums are exact (f64); see csrc/attention.hip for the rounding points. */
int mp_attn_fill_offload(mp_attn_t* h, mp_simhash_t* s, int layer_id, int request_id, const uint16_t* key_cache,
                         const uint16_t* value_cache, int64_t seq_len, int num_sink, int num_local,
                         uint16_t* avg_k, int16_t* codes, mp_stream_t stream);
/* SparseAttentionServer::attention_wrapper (and attention / scheduled_attention / *_bf16:
 * one function on the GPU), sparse_attention.cc:629-986, 1039-1211.
 *   output bf16 [B*H, D]; max_value_expsum f32 [2, B*H] (row 0 = max*log2e, row 1 = base-2
 *   LSE); query [B*H, D] of `query_dtype` (MP_DTYPE_BF16 as the __AVX512BF16__ build reads
 *   it, or MP_DTYPE_F32); query_norm f32 [B*H]; ind int32 [B*H, M]; nnz int32 [B*H]. */
int mp_attn_sparse(mp_attn_t* h, int layer_id, int K, int L, uint16_t* output,
                   float* max_value_expsum, const void* query, int query_dtype,
                   const float* query_norm, const int32_t* ind, const int32_t* nnz, int mem,
                   mp_stream_t stream);
/* SparseAttentionServer::full_attention, sparse_attention.cc:988-1037: dense attention over
 * rows [0, nnz[h]) (K == 0 baseline).  query f32 or bf16 [B*H, D]. */
int mp_attn_full(mp_attn_t* h, int layer_id, uint16_t* output, float* max_value_expsum,
                 const void* query, int query_dtype, const int32_t* nnz, int mem,
                 mp_stream_t stream);
int mp_attn_clear(mp_attn_t* h, mp_stream_t stream);  /* sparse_attention.cc:586-598 */
/* One decode step's (k, v) appended at row pos[b] of every kv head of request b -- the role of
 * flashinfer.append_paged_kv_cache at models/attnserver.py:281-290 for the static-window store
 * (a second mp_attn_t whose max_length is sink + local + generation buffer; the window's exact
 * attention is mp_attn_full on it, replacing BatchDecodeWithPagedKVCacheWrapper.run_return_lse,
 * :293-296).  k, v bf16 [B, Hkv, D]; pos int32 [B]; device pointers; the key norm of the new row
 * is computed on the fly.  A position >= max_length is reported by mp_attn_check (MP_ERR_DATA).
 * An append changes norms at caller-chosen rows of EVERY request of the layer: if this store is the one an LSH handle's
 * table words carry norms of (mp_lsh_build_with_norms / the decode entries' lazy packing), those payloads are no longer
 * used for the layer -- the decode kernels read a selected token's norm from HBM again (one more line request per
 * token: ~+1.5 us per layer at cfg 1) -- until the next mp_attn_fill* + table build.  Append into the window store (as
 * the reference does), not into the LSH-indexed one, to keep the payload path. */
int mp_attn_append(mp_attn_t* h, int layer_id, const uint16_t* k, const uint16_t* v,
                   const int32_t* pos, mp_stream_t stream);
/* The same with the two torch lines in front of it folded in (models/attnserver.py:267, 281-290):
 * the stored key is bf16(k - centre) (centre = the request's avg_k, bf16 [B, Hkv, D]) and the row is
 * pos[b] + pos_delta (the reference appends at kv_last_page_len - 1 after plan() incremented it). */
int mp_attn_append_centred(mp_attn_t* h, int layer_id, const uint16_t* k, const uint16_t* v,
                           const uint16_t* centre, const int32_t* pos, int pos_delta, mp_stream_t stream);
int mp_attn_check(mp_attn_t* h, mp_stream_t stream);
/* get_key_cache / get_value_cache / get_key_norm, sparse_attention.cc:1213-1233: device
 * pointers into the handle's storage.  K and V rows are INTERLEAVED per token in HBM
 * ([B*Hkv, M, 2, D]); *row_stride_elems = 2*D, the V pointer is the K pointer + D elements. */
int mp_attn_get_kv(mp_attn_t* h, int layer_id, void** key_dev, void** value_dev,
                   int64_t* row_stride_elems);
int mp_attn_get_key_norm(mp_attn_t* h, int layer_id, void** kn_dev);
/* HBM the store holds PER LAYER, bytes: [0] the interleaved K | V rows, [1] the f32 key norms
 * (sparse_attention.cc:546-583 allocates the same three arrays in host memory). */
int mp_attn_get_footprint(mp_attn_t* h, int64_t* bytes2);
/* The key-norm view above is writable (the reference's get_key_norm is a from_blob alias too,
 * sparse_attention.cc:1228-1233).  The one-launch decode entries may carry a request's norms inside its LSH table words
 * (see mp_lsh_get_id_bits): a caller that WRITES norms through the view says so here -- the request's norms get a new
 * version and the next mp_decode_* call of the layer packs them again.  mp_attn_fill* do this themselves;
 * mp_attn_append* mark the layer's norms "changed outside a fill" (in stream order, also inside a replayed graph): the
 * decode kernels then read the norms per token until the next fill.  No reference counterpart. */
int mp_attn_invalidate_norms(mp_attn_t* h, int layer_id, int request_id, mp_stream_t stream);
/* get_score, sparse_attention.cc:1235-1241: probabilities of the last sparse/full call,
 * f32 [B, H, M], first nnz entries per head in `ind` order.  Normalised (and, after the one-launch
 * decode, compacted) on demand, ONCE per call that produced logits: the library tracks that on the host,
 * so the view is defined after a call the host issued, not after the replay of a captured graph. */
int mp_attn_get_score(mp_attn_t* h, void** score_dev, mp_stream_t stream);

/* Debug (builds with -DMP_STAMPS=1 only -- scripts/build_variant.py stamps -DMP_STAMPS=1; a no-op in the product build):
 * device buffer of >= 64 uint64 receiving 100 MHz wall-clock stamps at the phase
 * boundaries of workgroup 0 of the hot kernels (scripts/phase_times.py); NULL switches it off. */
int mp_debug_set_stamp_buffer(void* dev_u64x64);

/* Debug: 1 if workgroup b of a launch was observed to run on XCD b % 8 on the current device (measured
 * once per process); this is what lets the cluster hand-off of the fused decode kernel stay inside one
 * XCD's L2.  0 = not observed, the hand-off writes through to memory instead. */
int mp_debug_xcd_round_robin(void);

/* Debug: A/B switches for measurements and tests, process-wide, read at every call (never needed in
 * production; unknown names return MP_ERR_INVALID):
 *   "decode_two_launch"  0/1   mp_decode_sparse_layer as (hash + retrieve) then attention: two launches
 *   "decode_cluster"     0 = auto, n = workgroups per head of the one-launch decode = token ranges of the
 *                        tables (rounded down to a power of two, at most 32); read by mp_lsh_alloc
 *   "decode_agent_scope" 0/1   cluster hand-off through memory even where the XCD placement was observed
 *   "stamp_stride"       n > 0: every workgroup b of the decode kernel writes its phase stamps at
 *                        [b * n + slot] of the stamp buffer (which must hold grid * n entries); 0 = workgroup 0 only
 *   "decode_mfma_hash"   0/1   mp_decode_sparse_layer with the query SimHash as the MFMA kernel's own launch in
 *                        front of the decode kernel instead of the hash fused into it (measured slower: profiles/)
 *   "decode_split_hash"  -1 = auto, 0 = never, 1 = always (clusters whose workgroups share an XCD): the hyperplanes
 *                        are split over the workgroups of a head's cluster and the sign bits exchanged through the
 *                        XCD's L2, with a bounded wait and hashing alone as the fallback; 2 = split but nobody
 *                        publishes (test: every workgroup takes the fallback)
 *   "decode_quad_hash"   one workgroup per head (B*H >= CUs / 2), B*H a multiple of 32, head_dim 128, XCD placement observed:
 *                        1 = the four heads of an XCD residue inside a block of 32 hash together -- each evaluates a quarter of
 *                        the hyperplanes against the four query rows with v_mfma_f32_32x32x16_bf16 and hands the sign bits to
 *                        their heads through the XCD's L2 (words tagged with the launch's number, bounded wait, hashing alone as
 *                        the fallback); 2 = the same but nobody publishes (test: every head falls back); 0 = never; -1 = the
 *                        library's choice
 *   "decode_kn_payload"  1 = the decode entries pack the key norms into the table entries and use them (default, while
 *                        the layer's ids fit 17 bits), 0 = one HBM access per selected token
 *   "decode_direct"      -1 = auto, 0 = never, 1 = always (where R > 1): keep direct slots (length, position + the first
 *                        ids: 128, 64 or 32 bytes by the mean piece length) for every (table, bucket, token range)
 *                        piece; read by mp_lsh_alloc
 *                        (2 = also at R = 1, one workgroup per head: measured -0.5 us of 28.9 per layer at cfg 3 for
 *                        +1.26 GB per layer, not taken by default)
 *   "decode_slot_log2"   0 = slot width by the mean piece length (default), 3 / 4 / 5 = 32- / 64- / 128-byte slots forced;
 *                        process-wide, read at ALLOC only (a handle keeps the width it was allocated with)
 *   "attn_head_kernel"   -1 = auto, 0 = split-KV kernel with the in-launch ticket merge, 1 = one workgroup per head
 *   "attn_gx"            0 = auto, n = split-KV workgroups per head
 *   "attn_dense_grouped" 1 = mp_attn_full reads K/V once per kv group (default), 0 = once per query head
 *   "host_zero_copy"     1 = MP_MEM_HOST calls let the kernels work on pinned memory in place -- the caller's pinned
 *                        buffers, or the handle's pinned mirror of a pageable one (default); 0 = staged copies through
 *                        the copy engine (the fallback, kept under test)
 *   "host_flag_wait"     1 = MP_MEM_HOST calls wait for their launches by spinning on a word a one-thread kernel writes
 *                        to pinned memory instead of hipStreamSynchronize (A/B; measured no gain over the whole layer)
 *   "build_rank_exact"   1 = mp_lsh_build* rank the tokens of a bucket by match-any ballots always; 0 (default) = by the order in
 *                        which the LDS serves the lanes of one atomic instruction (lane order on gfx950), every written bucket run
 *                        verified, the request rebuilt with the exact ranking if one does not ascend
 *   "build_rank_fallbacks"   COUNTER of such rebuilds (expected 0)
 *   "build_rank_inject"  n > 0: the next n table builds behave as if that verification had failed (test hook for the rebuild)
 *   "host_fast_hits" / "host_fast_edited" / "host_fast_unpaired"   COUNTERS (get to read, set 0 to reset): MP_MEM_HOST
 *                        mp_attn_sparse calls that recognised the rows mp_lsh_batch_retrieve had just handed out (no index
 *                        upload) / found the pairing but a row edited (launch dropped, rows uploaded) / found no pairing
 *   "host_speculate"     1 (default) = a MP_MEM_HOST mp_lsh_batch_retrieve enqueues the paired store's attention launch behind
 *                        its own kernel once the store's last MP_MEM_HOST mp_attn_sparse call came with a pinned query tensor
 *                        (the reference's caller reuses one, models/attnserver.py:61-66, and fills it before batch_retrieve,
 *                        :273): the attention call then checks that it is the call that launch assumed -- same store, layer,
 *                        K, L, dtype, the same query tensor holding the same bytes, ||q|| within 2e-6 of the launch's own,
 *                        rows untouched -- and copies the outputs out: two library calls, one wait.  Anything else is
 *                        served as without the option.  0 = never
 *   "host_spec_hits" / "host_spec_misses"   COUNTERS: attention calls served by such a launch / launches whose assumptions the
 *                        call did not meet
 *   "host_copy_prefetch" MP_MEM_HOST mp_lsh_batch_retrieve with pageable `results`: while a row is copied out of the handle's
 *                        pinned mirror the NEXT row is prefetched -- whole where it is at most this many 64-byte lines (default
 *                        48), its first 8 lines otherwise; 0 = no prefetch (A/B: EXPERIMENTS.md R6-2)
 *   "host_ret_calls", "host_ret_ns_enqueue" / "_wait" / "_copy"   COUNTERS (reset by setting 0): zero-copy MP_MEM_HOST
 *                        batch_retrieve calls, and the nanoseconds they spent up to their last launch, waiting for their
 *                        completion word, copying counts and rows out (int: good for ~40 000 calls between resets)
 *   "simhash_exact_norm" 1 = the fused query hash normalises the row by the exact f64 sequence always (A/B, tests);
 *                        0 (default) = a fast f32 form with the exact sequence as its fallback: identical codes
 *   "decode_cluster"     0 = auto, else workgroups per query head of the one-launch decode (1 .. 32); read by mp_lsh_alloc
 * (the `host_register` option of rounds 2-3 -- hipHostRegister of a caller's pageable buffer -- was removed in round 4) */
int mp_debug_set_option(const char* name, int value);
int mp_debug_get_option(const char* name, int* value);

/* Debug (test hook): route the raw MFMA accumulators of the following mp_simhash_query calls on `s` to a
 * device buffer f32 [R, K*L] (NULL switches it off) so that the exact-sign guard band can be validated. */
int mp_simhash_debug_acc(mp_simhash_t* s, float* dev_buf);

/* ---------------------------------------------------------------- one decode step of one layer
 * The device-resident equivalent of LSHSparseAttnServer.decode lines 264-300
 * (models/attnserver.py): q-hash -> batch_retrieve -> attention_wrapper as ONE kernel launch
 * (head_dim 64 or 128; other shapes and very long max_length run it as two launches): the selected
 * ids stay on chip, while codes, results and nnz are still written to the handles' HBM buffers as
 * by-products (get_mask / get_score keep working).  q bf16 [B*H, D] device; output bf16 [B*H, D]
 * device; max_value_expsum f32 [2, B*H] device.  nnz_out (optional, device int32 [B*H]) receives
 * the per-head selected-token counts for statistics.
 * Side effect on `lsh` (see mp_lsh_get_id_bits): the first call for a layer after its tables or `attn`'s key norms
 * changed packs the norms into the layer's table words -- two extra kernels per request on `stream`, once (at cfg 1
 * ~1.4 ms per layer: one pass over the 472-MB tables, the 1.26-GB direct slots rebuilt); a call under stream capture
 * never packs.  Whether a KV group's payloads are USED is decided by the kernel from device words the fills and the
 * packing write in stream order (the version of the norms the rows carry against the version the store holds now): a
 * graph captured before the packing uses the payloads once they exist, one replayed after mp_attn_fill* / mp_lsh_fill /
 * mp_lsh_build reads the norms per token until an eager call has packed again. */
int mp_decode_sparse_layer(mp_simhash_t* s, mp_lsh_t* lsh, mp_attn_t* attn, int layer_id,
                           const uint16_t* q, uint16_t* output, float* max_value_expsum,
                           int32_t* nnz_out, mp_stream_t stream);

/* mp_decode_sparse_layer / mp_decode_layer_window with a `flags` argument (0 = exactly the calls above).  No reference
 * counterpart: the reference's two calls (models/attnserver.py:299-300) hand `results` / `nnz` from one to the other through
 * the caller, and library/lsh/test.py, library/sparse_attention/test.py read get_mask / get_score back -- the by-products the
 * one-launch entries keep writing for them.  A serving loop reads neither:
 *   MP_DECODE_NO_BYPRODUCTS  the launch writes `output`, `max_value_expsum` and the per-head counts (nnz_out) and NOTHING
 *       else -- no query codes, no ||q||, no result rows, no logits -- and hands its selected ids to the gather through an
 *       unordered on-chip list (a token joins it the moment its second collision is counted), which takes the popcount
 *       sweep, the block scan and the ordered emission off the launch's dependent chain.  Same selected SET and counts
 *       (lsh.cc:266-283), same arithmetic per token (sparse_attention.cc:164-240); the tokens are folded into the softmax in
 *       the order they were found, so `output` / the LSE agree with the flags = 0 call to f32 summation order (<= 1 bf16
 *       ulp, 1e-3 on the LSE -- the parity tolerance), not bit for bit, and two runs need not agree bit for bit either.
 *       After such a call mp_lsh_get_mask and mp_attn_get_score return MP_ERR_STATE until the next call that produces
 *       their inputs.  Shapes without the one-launch form (and the decode_two_launch / decode_mfma_hash debug options) run
 *       as flags = 0. */
#define MP_DECODE_NO_BYPRODUCTS 1u
int mp_decode_sparse_layer_ex(mp_simhash_t* s, mp_lsh_t* lsh, mp_attn_t* attn, int layer_id,
                              const uint16_t* q, uint16_t* output, float* max_value_expsum,
                              int32_t* nnz_out, unsigned int flags, mp_stream_t stream);

/* The same launch with the static window of the layer folded in (models/attnserver.py:281-308, after
 * the step's k, v have been appended with mp_attn_append): `window` is a second KV store holding the
 * sink + local + generated tokens, window_len int32 [B*H] (device) the number of its rows that are
 * live for each head.  Exact attention over those rows joins the softmax of the sampled tokens, which
 * is what BatchDecodeWithPagedKVCacheWrapper.run_return_lse (:293-296) followed by
 * flashinfer.merge_state (:305-308) computes; output is the merged hidden state, max_value_expsum[1]
 * the base-2 LSE over both parts (mp_attn_get_score after this call gives each sampled token's share of
 * that merged softmax, so a head's scores sum to the sampled part's weight, not to 1).
 * MP_ERR_UNSUPPORTED when the one-launch form does not exist for the
 * shape (then: mp_attn_full on the window + mp_decode_sparse_layer + mp_merge_state). */
int mp_decode_layer_window(mp_simhash_t* s, mp_lsh_t* lsh, mp_attn_t* attn, mp_attn_t* window,
                           int layer_id, const uint16_t* q, const int32_t* window_len, uint16_t* output,
                           float* max_value_expsum, int32_t* nnz_out, mp_stream_t stream);

int mp_decode_layer_window_ex(mp_simhash_t* s, mp_lsh_t* lsh, mp_attn_t* attn, mp_attn_t* window,
                              int layer_id, const uint16_t* q, const int32_t* window_len, uint16_t* output,
                              float* max_value_expsum, int32_t* nnz_out, unsigned int flags, mp_stream_t stream);

/* ---------------------------------------------------------------- LSE merge
 * Replaces flashinfer.merge_state as called at models/attnserver.py:308 (base-2 LSEs):
 * v = (2^sa va + 2^sb vb) / 2^s, s = log2(2^sa + 2^sb).  va, vb, v bf16 [R, D]; sa, sb, s f32 [R]
 * (s may be NULL).  Device pointers only. */
int mp_merge_state(const uint16_t* va, const float* sa, const uint16_t* vb, const float* sb,
                   int R, int D, uint16_t* v, float* s, mp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGICPIG_HIP_H */
