// capi.hip -- the C ABI declared in include/magicpig_hip.h: handle state in HBM + launches.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "common.h"

#include "capi_launchers.h"
#include "capi_support.h"

using namespace mp;

#include "capi_handles.h"

extern "C" {

int mp_version(void) { return 1; }
const char* mp_last_error(void) { return g_err.c_str(); }
const char* mp_arch(void) { return "gfx950"; }

// =================================================================== SimHash

int mp_simhash_create(mp_simhash_t** out) {
    MP_REQUIRE(out != nullptr, MP_ERR_INVALID, "mp_simhash_create: null out");
    *out = new (std::nothrow) mp_simhash();
    MP_REQUIRE(*out != nullptr, MP_ERR_NOMEM, "mp_simhash_create: out of host memory");
    return MP_OK;
}

int mp_simhash_destroy(mp_simhash_t* s) {
    MP_ON_DEVICE(s);
    if (!s) return MP_OK;
    if (s->Wt) (void)hipFree(s->Wt);
    if (s->Wk) (void)hipFree(s->Wk);
    if (s->wnorm) (void)hipFree(s->wnorm);
    delete s;
    return MP_OK;
}

int mp_simhash_set_planes(mp_simhash_t* s, int D, int K, int L, const uint16_t* hash_func, int mem,
                          mp_stream_t stream) {
    MP_REQUIRE(s && hash_func, MP_ERR_INVALID, "mp_simhash_set_planes: null argument");
    MP_REQUIRE(L >= 1 && L < 65536, MP_ERR_INVALID, "mp_simhash_set_planes: L out of range");
    MP_REQUIRE(simhash_supported(D, K), MP_ERR_UNSUPPORTED,
               "mp_simhash_set_planes: need head_dim % 16 == 0, head_dim <= 256, 1 <= K <= 15");
    hipStream_t st = (hipStream_t)stream;
    MP_ON_DEVICE(s);                                   // re-setting planes keeps the handle's device
    if (s->device < 0) s->device = current_device();
    if (s->Wt) { (void)hipFree(s->Wt); s->Wt = nullptr; }
    if (s->Wk) { (void)hipFree(s->Wk); s->Wk = nullptr; }
    if (s->wnorm) { (void)hipFree(s->wnorm); s->wnorm = nullptr; }
    s->D = D; s->K = K; s->L = L;
    s->KLpad = simhash_padded_cols(K, L);
    MP_HIP_CHECK(hipMalloc((void**)&s->Wt, (size_t)s->KLpad * D * 2));
    MP_HIP_CHECK(hipMalloc((void**)&s->Wk, (size_t)s->KLpad * D * 2));
    MP_HIP_CHECK(hipMalloc((void**)&s->wnorm, (size_t)s->KLpad * 4));
    DevBuf tmp;
    const void* W = nullptr;
    int rc = stage_in(hash_func, (size_t)D * K * L * 2, mem, tmp, &W);
    if (rc) return rc;
    MP_HIP_CHECK(launch_simhash_prepare((const uint16_t*)W, D, K, L, s->Wt, s->Wk, s->wnorm, st));
    MP_HIP_CHECK(hipStreamSynchronize(st));
    return MP_OK;
}

// test hook: route raw MFMA accumulators of the next query calls to a device buffer f32 [R][K*L] so
// the guard band can be validated against an exact reference.
int mp_simhash_debug_acc(mp_simhash_t* s, float* dev_buf) {
    MP_REQUIRE(s, MP_ERR_INVALID, "mp_simhash_debug_acc: null handle");
    s->dbg_acc = dev_buf;
    return MP_OK;
}

int mp_simhash_query(mp_simhash_t* s, const uint16_t* q, int R, int32_t* codes, float* qnorm,
                     int mem, mp_stream_t stream) {
    MP_ON_DEVICE(s);
    MP_REQUIRE(s && s->Wt, MP_ERR_STATE, "mp_simhash_query: planes not set");
    MP_REQUIRE(q && codes && R >= 1, MP_ERR_INVALID, "mp_simhash_query: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (mem == MP_MEM_DEVICE) {
        // a handful of rows (a decode step's B*H <= 64 query heads): one workgroup per row on the vector pipes; the MFMA
        // kernel's 32-row tiles take the bulk case (same exact-sign definition: identical codes, tests/test_gpu_parity.py)
        if (R <= 64 && s->dbg_acc == nullptr && qnorm != nullptr && lsh_hash_only_supported(s->L))
            MP_HIP_CHECK(launch_lsh_hash_only(q, s->Wk, s->wnorm, s->D, s->K, s->KLpad, codes, qnorm, R, s->L, st));
        else
            MP_HIP_CHECK(launch_simhash_query(q, s->Wt, s->wnorm, R, s->D, s->K, s->L, codes, qnorm,
                                              s->dbg_acc, st));
        return MP_OK;
    }
    DevBuf dq, dc, dn;
    MP_HIP_CHECK(dq.alloc((size_t)R * s->D * 2));
    MP_HIP_CHECK(dc.alloc((size_t)R * s->L * 4));
    MP_HIP_CHECK(dn.alloc((size_t)R * 4));
    MP_HIP_CHECK(hipMemcpy(dq.p, q, (size_t)R * s->D * 2, hipMemcpyHostToDevice));
    MP_HIP_CHECK(launch_simhash_query(dq.as<uint16_t>(), s->Wt, s->wnorm, R, s->D, s->K, s->L,
                                      dc.as<int32_t>(), dn.as<float>(), s->dbg_acc, st));
    MP_HIP_CHECK(hipStreamSynchronize(st));
    MP_HIP_CHECK(hipMemcpy(codes, dc.p, (size_t)R * s->L * 4, hipMemcpyDeviceToHost));
    if (qnorm) MP_HIP_CHECK(hipMemcpy(qnorm, dn.p, (size_t)R * 4, hipMemcpyDeviceToHost));
    return MP_OK;
}

int mp_simhash_keys(mp_simhash_t* s, const uint16_t* keys, int Hkv, int64_t n, int16_t* codes,
                    int mem, mp_stream_t stream) {
    MP_ON_DEVICE(s);
    MP_REQUIRE(s && s->Wt, MP_ERR_STATE, "mp_simhash_keys: planes not set");
    MP_REQUIRE(keys && codes && Hkv >= 1 && n >= 1, MP_ERR_INVALID, "mp_simhash_keys: bad argument");
    hipStream_t st = (hipStream_t)stream;
    DevBuf dk, dc;
    const uint16_t* kd = keys;
    int16_t* cd = codes;
    const size_t kb = (size_t)Hkv * n * s->D * 2, cb = (size_t)Hkv * s->L * n * 2;
    if (mem == MP_MEM_HOST) {
        MP_HIP_CHECK(dk.alloc(kb));
        MP_HIP_CHECK(dc.alloc(cb));
        MP_HIP_CHECK(hipMemcpy(dk.p, keys, kb, hipMemcpyHostToDevice));
        kd = dk.as<uint16_t>();
        cd = dc.as<int16_t>();
    }
    MP_HIP_CHECK(launch_simhash_keys(kd, s->Wt, s->wnorm, Hkv, n, s->D, s->K, s->L, cd, st));
    if (mem == MP_MEM_HOST) {
        MP_HIP_CHECK(hipStreamSynchronize(st));
        MP_HIP_CHECK(hipMemcpy(codes, dc.p, cb, hipMemcpyDeviceToHost));
    }
    return MP_OK;
}

// =================================================================== LSH

int mp_lsh_create(mp_lsh_t** out) {
    MP_REQUIRE(out != nullptr, MP_ERR_INVALID, "mp_lsh_create: null out");
    *out = new (std::nothrow) mp_lsh();
    MP_REQUIRE(*out != nullptr, MP_ERR_NOMEM, "mp_lsh_create: out of host memory");
    return MP_OK;
}

static void lsh_free(mp_lsh_t* h) {
    host_ret_forget(h);
    {   // the store this handle speculates for no longer lists it
        std::lock_guard<std::mutex> lock(g_host_ret_mu);
        if (h->spec.attn != nullptr) {
            auto& ow = h->spec.attn->spec_owners;
            for (size_t i = 0; i < ow.size(); ++i)
                if (ow[i] == h) { ow.erase(ow.begin() + i); break; }
            h->spec.attn = nullptr;
        }
        h->spec.launched = false;
    }
    for (auto p : h->bounds) if (p) (void)hipFree(p);
    for (auto p : h->table) if (p) (void)hipFree(p);
    for (auto p : h->slots) if (p) (void)hipFree(p);
    h->bounds.clear();
    h->table.clear();
    h->slots.clear();
    void* ptrs[] = {h->last_query, h->err, h->codes, h->results, h->nnz, h->qnorm, h->xw, h->xseq, h->pay_bad, h->att_ver_dev, h->idbits_dev,
                    h->hr_rows, h->hr_nnz};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    h->hr_rows = nullptr; h->hr_nnz = nullptr;
    h->last_query = nullptr; h->err = nullptr; h->codes = nullptr; h->results = nullptr;
    h->nnz = nullptr; h->qnorm = nullptr; h->xw = nullptr; h->xseq = nullptr; h->pay_bad = nullptr; h->att_ver_dev = nullptr; h->idbits_dev = nullptr;
    h->small.release();
    h->big.release();
    h->hostmap.release();
    h->hostflag.release();
    h->allocated = false;
}

int mp_lsh_destroy(mp_lsh_t* h) {
    MP_ON_DEVICE(h);
    if (!h) return MP_OK;
    lsh_free(h);
    delete h;
    return MP_OK;
}

static int decode_cluster_size(int BH, int64_t M, int asked);

int mp_lsh_alloc(mp_lsh_t* h, int K, int L, int num_layers, int num_attention_heads,
                 int num_key_value_heads, int batch_size, int max_length) {
    return mp_lsh_alloc_ex(h, K, L, num_layers, num_attention_heads, num_key_value_heads, batch_size, max_length, -1, 0);
}

int mp_lsh_alloc_ex(mp_lsh_t* h, int K, int L, int num_layers, int num_attention_heads,
                    int num_key_value_heads, int batch_size, int max_length, int64_t accel_budget_bytes, int ranges) {
    MP_REQUIRE(h, MP_ERR_INVALID, "mp_lsh_alloc: null handle");
    MP_REQUIRE(ranges == 0 || (ranges >= 1 && ranges <= MAX_CLUSTER && (ranges & (ranges - 1)) == 0), MP_ERR_INVALID,
               "mp_lsh_alloc_ex: ranges must be 0 (auto) or a power of two in [1, 32]");
    MP_REQUIRE(!h->allocated, MP_ERR_STATE, "mp_lsh_alloc: already allocated");
    MP_REQUIRE(K >= 1 && K <= 15, MP_ERR_INVALID, "mp_lsh_alloc: K must be in [1, 15] (int16 codes)");
    MP_REQUIRE(L >= 1 && L < 65536, MP_ERR_INVALID, "mp_lsh_alloc: L out of range");
    MP_REQUIRE(num_layers >= 1 && batch_size >= 1 && num_key_value_heads >= 1 &&
                   num_attention_heads >= num_key_value_heads &&
                   num_attention_heads % num_key_value_heads == 0,
               MP_ERR_INVALID, "mp_lsh_alloc: bad head/layer/batch counts");
    MP_REQUIRE(max_length >= 1 && max_length <= (1 << 22), MP_ERR_INVALID,
               "mp_lsh_alloc: max_length must be in [1, 2^22]");
    MP_REQUIRE(retrieve_lds_bytes(max_length, L) <= lsh_lds_limit(), MP_ERR_UNSUPPORTED,
               "mp_lsh_alloc: collision bitmaps for max_length do not fit the 160 KiB LDS");
    h->device = current_device();
    h->K = K; h->L = L; h->NB = 1 << K; h->layers = num_layers;
    h->H = num_attention_heads; h->Hkv = num_key_value_heads; h->B = batch_size;
    h->G = h->H / h->Hkv; h->M = max_length;
    const size_t groups = (size_t)h->B * h->Hkv, BH = (size_t)h->B * h->H;
    h->idbits_of.assign((size_t)num_layers, 17);
    h->att_ver.assign((size_t)num_layers, std::vector<uint32_t>((size_t)batch_size, 0));
    h->R = decode_cluster_size((int)BH, h->M, ranges);
    h->range_len = lsh_range_len(h->M, h->R);
    h->accel_budget = accel_budget_bytes;
    h->accel_used = 0;
    h->hr_refused = false;
    // direct piece slots (lsh.hip: lsh_slots_kernel): one 128-byte record per (table, bucket, token range)
    // holding the piece's length, position and first 30 ids.  Worth their memory (groups x L x 2^K x R x 128 B per layer:
    // 1.26 GB at cfg 1) where a head is split over several workgroups AND a piece rarely overflows a slot:
    // mean piece length max_length / (2^K R) <= 12.5 ids (P[Poisson(12.5) > 31] = 2e-6).
    h->slot_log2 = lsh_slot_log2(h->M, h->NB, h->R);                     // 32, 16 or 8 words per slot (lsh.hip)
    h->slot_words = 1 << h->slot_log2;
    bool direct = h->R > 1 && (double)h->M <= 12.5 * (double)h->NB * h->R &&
                  (double)L * h->NB * h->R * (double)h->slot_words < 2147483648.0;     // 32-bit slot offsets inside a group
    const size_t slot_bytes = groups * L * h->NB * (size_t)h->R * h->slot_words * 4;
    if (direct) {   // an accelerator, not a requirement
        if (accel_budget_bytes >= 0) {                // the caller's figure: the slots exist iff all layers' fit it
            if ((double)slot_bytes * num_layers > (double)accel_budget_bytes) direct = false;
        } else {                                      // mp_lsh_alloc: never more than a third of what is free right now
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)slot_bytes * num_layers > (double)free_b / 3.0)
                direct = false;
        }
    }
    if (const int o = g_opt.decode_direct.load(); o >= 0)                                  // A/B switch, read at alloc
        direct = o != 0 && (h->R > 1 || o == 2) &&     // 2: also at R = 1 (round 5 experiment: one workgroup per head)
                 (double)L * h->NB * h->R * (double)h->slot_words < 2147483648.0;
    int rc = MP_OK;
    for (int i = 0; i < num_layers && rc == MP_OK; ++i) {
        void* b = nullptr; void* t = nullptr; void* sl = nullptr;
        rc = alloc_zero(&b, groups * L * h->NB * (size_t)(h->R + 1) * 4);
        if (rc == MP_OK) rc = alloc_zero(&t, groups * L * (size_t)h->M * 4);
        h->bounds.push_back((int32_t*)b);
        h->table.push_back((int32_t*)t);
        if (rc == MP_OK && direct) {
            if (alloc_zero(&sl, slot_bytes) == MP_OK) {
                h->slots.push_back((int32_t*)sl);
            } else {                              // out of memory for the slots: run without them (sub-bounds path)
                (void)hipGetLastError();
                for (auto p : h->slots) if (p) (void)hipFree(p);
                h->slots.clear();
                direct = false;
            }
        }
    }
    h->xwords = 2 * ((K * L + 63) / 64);
    const bool quads = h->R == 1 && BH % 32 == 0;              // one workgroup per head, heads in blocks of 32: the quad hash's exchange
    if (rc == MP_OK && (h->R > 1 || quads)) rc = alloc_zero((void**)&h->xw, BH * (size_t)h->xwords * 8);
    if (rc == MP_OK && (h->R > 1 || quads)) {
        rc = alloc_zero((void**)&h->xseq, BH * 4);
        // words start at sequence 0, launches at 1: nothing stale can pass for a word of the first launch
        // (quads: the word of a quad's first head is its arrival counter, launch number = counter / 4: starts at 4)
        if (rc == MP_OK && hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(h->xseq), quads ? 4 : 1, BH) != hipSuccess) rc = MP_ERR_HIP;
    }
    if (rc == MP_OK) rc = alloc_zero((void**)&h->pay_bad, (size_t)num_layers * batch_size * num_key_value_heads * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->att_ver_dev, (size_t)num_layers * batch_size * num_key_value_heads * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->idbits_dev, (size_t)num_layers * 4);
    if (rc == MP_OK && hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(h->idbits_dev), 17, (size_t)num_layers) != hipSuccess)
        rc = MP_ERR_HIP;
    if (rc == MP_OK) rc = alloc_zero((void**)&h->last_query, BH * L * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->err, 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->codes, BH * L * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->results, BH * (size_t)h->M * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->nnz, BH * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->qnorm, BH * 4);
    if (rc != MP_OK) { lsh_free(h); return rc; }
    h->accel_used = h->slots.empty() ? 0 : (int64_t)slot_bytes * num_layers;
    h->allocated = true;
    return MP_OK;
}

// Workgroups per query head of the one-launch decode (= token ranges of the tables): spread a head over several CUs
// while there are idle ones.  Up to 8 members: whenever B*H <= CUs / 8.  16 and 32 (round 4: cfg 4's 8 query heads per
// GPU used 64 of the 256 CUs): only while a member still owns >= 8 192 tokens -- with fewer the chain of a member is
// latency, not bytes, and more members only lengthen the hand-off.  Measured at cfg 4 (131 264 tokens, EXPERIMENTS.md
// R4-3): 8 members 20.7 us per layer, 16 members 17.0, 32 members 17.7 (the ticket of 32 arrivals and the merge of 32
// records cost what the shorter gather saves) -> 16; cfg 0's 4 288 tokens stay at 8.
static int decode_cluster_size(int BH, int64_t M, int asked) {
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        cus = prop.multiProcessorCount;
    int cluster = cus / (BH > 0 ? BH : 1);
    int forced = g_opt.decode_cluster.load();                                // debug override (process-wide), else the caller's
    if (forced < 1) forced = asked;                                          // mp_lsh_alloc_ex figure, else (0) the rule below
    if (forced >= 1) cluster = forced;
    if (cluster > MAX_CLUSTER) cluster = MAX_CLUSTER;
    const int64_t slices = (M + 63) / 64;
    if (cluster > slices) cluster = (int)slices;
    int r = 1;
    while (2 * r <= cluster) r *= 2;
    while (forced < 1 && r > 8 && M / r < 8192) r /= 2;
    return r;
}

static int lsh_check_slot(mp_lsh_t* h, int layer_id, int request_id, int64_t n, const char* who) {
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, std::string(who) + ": not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, std::string(who) + ": layer_id out of range");
    MP_REQUIRE(request_id >= 0 && request_id < h->B, MP_ERR_INVALID, std::string(who) + ": request_id out of range");
    MP_REQUIRE(n >= 0 && n <= h->M, MP_ERR_INVALID, std::string(who) + ": sequence longer than max_length");
    return MP_OK;
}

// reads and clears the device flag: bit 0 -> MP_ERR_DATA; *unsorted (optional) <- bit 2 (a bucket whose ids
// do not ascend: not an error, the caller re-sorts)
static int lsh_read_err(mp_lsh_t* h, hipStream_t st, const char* who, bool* unsorted = nullptr, bool* wide = nullptr,
                        bool* misranked = nullptr) {
    int flag = 0;
    MP_HIP_CHECK(hipMemcpyAsync(&flag, h->err, 4, hipMemcpyDeviceToHost, st));
    MP_HIP_CHECK(hipStreamSynchronize(st));
    if (flag) MP_HIP_CHECK(hipMemsetAsync(h->err, 0, 4, st));
    if (unsorted) *unsorted = (flag & 4) != 0;
    if (wide) *wide = (flag & 32) != 0;
    if (misranked) *misranked = (flag & 64) != 0;        // the build's fast ranking saw a bucket run that does not ascend
    if (flag & 1)
        return fail(MP_ERR_DATA, std::string(who) + ": device-side validation failed (codes not sorted / "
                                                    "out of [0, 2^K) or token id out of [0, max_length))");
    return MP_OK;
}

// the version of the store's norms the rows of (layer, request) carry, on the host and -- in stream order -- on the device
static int lsh_set_version(mp_lsh_t* h, int layer_id, int request_id, uint32_t v, hipStream_t st) {
    h->att_ver[layer_id][request_id] = v;
    unsigned int* d = h->att_ver_dev + ((size_t)layer_id * h->B + request_id) * h->Hkv;
    MP_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d), (int)v, (size_t)h->Hkv, st));
    return MP_OK;
}

// A table word is  id | payload << 17  while every id of the LAYER is below 2^17.  The first fill that brings a wider
// id (a context past 131 072 offloaded tokens) turns the layer's words into plain ids for good (until mp_lsh_clear):
// the payloads the other requests' rows may carry are stripped, their direct slots rebuilt.
static int lsh_widen(mp_lsh_t* h, int layer_id, int except_request, hipStream_t st) {
    if (h->idbits_of[layer_id] == 0) return MP_OK;
    const int rows = h->Hkv * h->L;
    for (int r = 0; r < h->B; ++r) {
        if (r == except_request || h->att_ver[layer_id][r] == 0) continue;
        int32_t* t = h->table[layer_id] + (size_t)r * rows * h->M;
        int* flag = h->pay_bad + ((size_t)layer_id * h->B + r) * h->Hkv;
        MP_HIP_CHECK(launch_lsh_attach_norms(t, nullptr, h->Hkv, h->L, h->M, 17, flag, st));
        if (!h->slots.empty()) {
            int32_t* b = h->bounds[layer_id] + (size_t)r * rows * h->NB * (h->R + 1);
            MP_HIP_CHECK(launch_lsh_slots(t, b, h->slots[layer_id] + (size_t)r * rows * h->NB * h->R * h->slot_words, rows, h->NB,
                                          h->R, h->M, h->slot_log2, st));
        }
        int rc = lsh_set_version(h, layer_id, r, 0, st);
        if (rc) return rc;
    }
    h->idbits_of[layer_id] = 0;
    MP_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->idbits_dev + layer_id), 0, 1, st));
    return MP_OK;
}

int mp_lsh_fill(mp_lsh_t* h, int layer_id, int request_id, const int16_t* sorted_codes,
                const int32_t* sorted_ids, int64_t n, int mem, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    int rc = lsh_check_slot(h, layer_id, request_id, n, "mp_lsh_fill");
    if (rc) return rc;
    MP_REQUIRE(sorted_codes && sorted_ids, MP_ERR_INVALID, "mp_lsh_fill: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int rows = h->Hkv * h->L;
    DevBuf dc, di;
    const void *c = nullptr, *i = nullptr;
    rc = stage_in(sorted_codes, (size_t)rows * n * 2, mem, dc, &c);
    if (rc) return rc;
    rc = stage_in(sorted_ids, (size_t)rows * n * 4, mem, di, &i);
    if (rc) return rc;
    int32_t* b = h->bounds[layer_id] + (size_t)request_id * rows * h->NB * (h->R + 1);
    int32_t* t = h->table[layer_id] + (size_t)request_id * rows * h->M;
    if ((rc = lsh_set_version(h, layer_id, request_id, 0, st)) != MP_OK) return rc;   // the rows are rewritten with plain ids
    MP_HIP_CHECK(launch_lsh_fill((const int16_t*)c, (const int32_t*)i, rows, n, h->NB, h->M, h->R, b, t,
                                 h->err, st));
    bool unsorted = false, wide = false;
    rc = lsh_read_err(h, st, "mp_lsh_fill", &unsorted, &wide);
    if (rc) return rc;
    if (wide && (rc = lsh_widen(h, layer_id, request_id, st)) != MP_OK) return rc;
    if (unsorted) {
        // R > 1 and some bucket's ids do not ascend (an unstable sort, models/attnserver.py:187): put the codes
        // back in token order and let the device counting sort rebuild the rows -- same buckets, ascending ids
        DevBuf tok;
        MP_HIP_CHECK(tok.alloc((size_t)rows * n * 2));
        MP_HIP_CHECK(hipMemsetAsync(tok.p, 0xff, (size_t)rows * n * 2, st));      // code -1: a token no id named
        MP_HIP_CHECK(launch_lsh_unsort((const int16_t*)c, (const int32_t*)i, rows, n, tok.as<int16_t>(), h->err, st));
        MP_HIP_CHECK(launch_lsh_build(tok.as<int16_t>(), rows, n, h->NB, h->M, h->R, b, t, h->err, nullptr, h->L, 0,
                                      nullptr, nullptr, nullptr, /*exact_rank=*/true, st));
        // an id outside [0, n) is flagged by the unsort, a token missing from the id list shows up as code -1
        if (lsh_read_err(h, st, "mp_lsh_fill") != MP_OK)
            return fail(MP_ERR_DATA, "mp_lsh_fill: a bucket's ids do not ascend (unstable sort) and the ids of a row are "
                                     "not a permutation of [0, n): the rows cannot be re-sorted on device");
    }
    MP_HIP_CHECK(launch_lsh_subbounds(t, b, rows, h->NB, h->R, h->M, 0, st));
    if (!h->slots.empty())
        MP_HIP_CHECK(launch_lsh_slots(t, b, h->slots[layer_id] + (size_t)request_id * rows * h->NB * h->R * h->slot_words, rows,
                                      h->NB, h->R, h->M, h->slot_log2, st));
    if (mem == MP_MEM_HOST) MP_HIP_CHECK(hipStreamSynchronize(st));
    return MP_OK;
}

// shared by mp_lsh_build (attn == nullptr: plain ids) and mp_lsh_build_with_norms
static int lsh_build_entry(mp_lsh_t* h, mp_attn_t* attn, int layer_id, int request_id, const int16_t* codes, int64_t n,
                           int mem, hipStream_t st, const char* who);

int mp_lsh_build(mp_lsh_t* h, int layer_id, int request_id, const int16_t* codes, int64_t n,
                 int mem, mp_stream_t stream) {
    return lsh_build_entry(h, nullptr, layer_id, request_id, codes, n, mem, (hipStream_t)stream, "mp_lsh_build");
}

// the store's pinned block in host-buffer mode: (q | qn | nnz | offsets) up, (out | mve) down, and the ||q|| a
// speculative launch computed
struct AttnHostLayout {
    size_t qbytes, o_q, o_qn, o_nnz, o_offs, o_out, o_mve, o_sqn, o_qsnap, o_end;
};
static AttnHostLayout attn_host_layout(int BH, int D, int query_dtype) {
    AttnHostLayout a;
    a.qbytes = (size_t)BH * D * (query_dtype == MP_DTYPE_BF16 ? 2 : 4);
    a.o_q = 0;
    a.o_qn = a.o_q + ((a.qbytes + 15) & ~(size_t)15);
    a.o_nnz = a.o_qn + (size_t)BH * 4;
    a.o_offs = a.o_nnz + (size_t)BH * 4;
    a.o_out = (a.o_offs + (size_t)(BH + 1) * 4 + 15) & ~(size_t)15;
    a.o_mve = a.o_out + (size_t)BH * D * 2;
    a.o_sqn = (a.o_mve + (size_t)2 * BH * 4 + 15) & ~(size_t)15;
    a.o_qsnap = (a.o_sqn + (size_t)BH * 4 + 15) & ~(size_t)15;     // mp_lsh::Spec: the query bytes the launch ahead works on
    a.o_end = a.o_qsnap + ((a.qbytes + 15) & ~(size_t)15);
    return a;
}
static int attn_run(mp_attn_t* h, int layer_id, bool dense, int K, int L, uint16_t* output, float* mve, const void* query,
                    int query_dtype, const float* qn, const int32_t* ind, const int32_t* nnz, hipStream_t st);

// mp_lsh::Spec, in two steps around the retrieve kernel's launch.  prepare (IN FRONT of it, host work only): is there a store
// to launch for, is its caller's query tensor still pinned and mapped?  then snapshot the query bytes into the STORE's pinned
// block -- nothing enqueued ever reads the caller's memory.  launch (BEHIND the retrieve kernel and the completion word the call
// waits for, so none of it is on the retrieve's way): the kernel that brings the snapshot into HBM and computes the rows'
// norms, the attention kernel on that copy, the store's own completion word.  (Through the first version the copy + norm kernel
// read the caller's tensor and therefore ran in FRONT of the retrieve kernel: 4 us of kernel and launch gap on the call's path,
// and the host enqueued two launches before the one it was going to wait for -- EXPERIMENTS.md R6-2.)
static bool lsh_spec_prepare(mp_lsh_t* h, int layer_id, hipStream_t st) {
    h->spec.launched = h->spec.prepared = false;
    if (g_opt.host_speculate.load() == 0 || h->hr_rows == nullptr) return false;
    mp_attn_t* a = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_host_ret_mu);       // (the store's destroy clears the pointer under this mutex)
        a = h->spec.attn;
    }
    const int BH = h->B * h->H;
    if (a == nullptr || !a->allocated || a->device != h->device || a->B * a->H != BH || a->M != h->M ||
        layer_id >= a->layers || h->spec.q_host == nullptr)
        return false;
    const AttnHostLayout lo = attn_host_layout(BH, a->D, h->spec.q_dtype);
    if (a->small.reserve(lo.o_end) != MP_OK || a->small.hd == nullptr) return false;
    void* qdev = a->hostmap.resolve(h->spec.q_host, lo.qbytes);   // still pinned and mapped?  (the tensor may be gone)
    if (qdev == nullptr) return false;
    if (a->spec_qn == nullptr && hipMalloc((void**)&a->spec_qn, (size_t)BH * 4) != hipSuccess) {
        (void)hipGetLastError();
        a->spec_qn = nullptr;
        return false;
    }
    if (a->score_alt == nullptr) {
        if (hipMalloc((void**)&a->score_alt, (size_t)BH * a->M * 4) != hipSuccess ||
            hipMalloc((void**)&a->head_mz_alt, (size_t)BH * sizeof(float2)) != hipSuccess) {
            (void)hipGetLastError();
            if (a->score_alt) (void)hipFree(a->score_alt);
            a->score_alt = nullptr;
            a->head_mz_alt = nullptr;
            return false;
        }
    }
    (void)st;
    memcpy(reinterpret_cast<char*>(a->small.hp) + lo.o_qsnap, h->spec.q_host, lo.qbytes);    // what the launch will work on
    h->spec.q_bytes = lo.qbytes;
    h->spec.prepared = true;
    return true;
}

static bool lsh_spec_launch(mp_lsh_t* h, int layer_id, hipStream_t st) {
    if (!h->spec.prepared) return false;
    h->spec.prepared = false;
    mp_attn_t* a = h->spec.attn;
    const int BH = h->B * h->H;
    const AttnHostLayout lo = attn_host_layout(BH, a->D, h->spec.q_dtype);
    char* hd = reinterpret_cast<char*>(a->small.hd);
    char* dp = reinterpret_cast<char*>(a->small.dp);
    // the launch works on the store's SECOND set of score buffers and leaves the bookkeeping of the last caller-visible call
    // alone: until the attention call accepts it, it has not happened as far as get_score is concerned
    const int32_t* keep_lastz = a->lastz;
    const int keep_state = a->score_state, keep_R = a->seg_R;
    const int* keep_seg = a->seg_cnt;
    // Few heads (B*H <= 64, as in the attention entry): the rows' norms are computed HERE, on the host, while the retrieve kernel
    // runs -- f32 partial sums in 16 lanes, the device kernel's precision -- and the attention kernel's workgroups read
    // (q | ||q||) straight from the pinned snapshot: no launch in front of it.  Many heads: one kernel brings the snapshot into
    // HBM and computes the norms there.
    const bool direct = BH <= 64;
    const void* q_src = dp + lo.o_q;
    const float* qn_src = a->spec_qn;
    if (direct) {
        const char* hp = reinterpret_cast<const char*>(a->small.hp);
        float* sq = reinterpret_cast<float*>(const_cast<char*>(hp) + lo.o_sqn);
        const int D = a->D;
        for (int r = 0; r < BH; ++r) {
            float acc[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (h->spec.q_dtype == MP_DTYPE_BF16) {
                const uint16_t* x = reinterpret_cast<const uint16_t*>(hp + lo.o_qsnap) + (size_t)r * D;
                for (int d = 0; d + 16 <= D; d += 16)
                    for (int j = 0; j < 16; ++j) {
                        const uint32_t b = (uint32_t)x[d + j] << 16;
                        float f;
                        memcpy(&f, &b, 4);
                        acc[j] += f * f;
                    }
            } else {
                const float* x = reinterpret_cast<const float*>(hp + lo.o_qsnap) + (size_t)r * D;
                for (int d = 0; d + 16 <= D; d += 16)
                    for (int j = 0; j < 16; ++j) acc[j] += x[d + j] * x[d + j];
            }
            double sum = 0.0;
            for (int j = 0; j < 16; ++j) sum += (double)acc[j];
            sq[r] = (float)sqrt(sum);
        }
        q_src = hd + lo.o_qsnap;
        qn_src = reinterpret_cast<const float*>(hd + lo.o_sqn);
    } else if (launch_row_norm(hd + lo.o_qsnap, h->spec.q_dtype == MP_DTYPE_BF16, BH, a->D, a->spec_qn,
                               reinterpret_cast<float*>(hd + lo.o_sqn), dp + lo.o_q, st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    std::swap(a->score, a->score_alt);
    std::swap(a->head_mz, a->head_mz_alt);
    const int arc = attn_run(a, layer_id, false, h->spec.K, h->spec.L, reinterpret_cast<uint16_t*>(hd + lo.o_out),
                             reinterpret_cast<float*>(hd + lo.o_mve), q_src, h->spec.q_dtype, qn_src, h->hr_rows,
                             h->hr_nnz, st);
    std::swap(a->score, a->score_alt);
    std::swap(a->head_mz, a->head_mz_alt);
    a->lastz = keep_lastz;
    a->score_state = keep_state;
    a->seg_cnt = keep_seg;
    a->seg_R = keep_R;
    if (arc != MP_OK) return false;
    h->spec.done_flag = a->hostflag.arm(st);
    h->spec.stream = st;
    h->spec.attn_seq = ++a->host_seq;
    h->spec.layer = layer_id;
    h->spec.launched = true;
    return true;
}

int mp_lsh_batch_retrieve(mp_lsh_t* h, int layer_id, const int32_t* query, int32_t* results,
                          int32_t* nnz, int mem, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_lsh_batch_retrieve: not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, "mp_lsh_batch_retrieve: layer_id out of range");
    MP_REQUIRE(query && results && nnz, MP_ERR_INVALID, "mp_lsh_batch_retrieve: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int BH = h->B * h->H;
    const size_t qb = (size_t)BH * h->L * 4;
    if (mem == MP_MEM_DEVICE) {
        h->lastq = query;
        h->last_layer = layer_id;
        h->last_lean = false;
        MP_HIP_CHECK(launch_lsh_retrieve(h->bounds[layer_id], h->table[layer_id], query, results,
                                         nnz, BH, h->G, h->L, h->NB, h->M, h->R, h->idbits_dev + layer_id, nullptr, nullptr, nullptr, st));
        return MP_OK;
    }
    // host callers (models/attnserver.py:299 passes CPU tensors).  Zero copy: the kernel reads the codes from the handle's
    // pinned block and writes the ids straight into the caller's `results` rows (mapped once, HostMap) and the counts
    // into the pinned block: ONE launch, ONE synchronisation, no copy engine.
    host_ret_forget(h);                                   // hr_rows / hr_nnz are about to be rewritten
    h->spec.launched = false;
    const size_t o_codes = (size_t)(2 * BH + 1) * 4, o_sums = (o_codes + qb + 7) & ~(size_t)7;
    int rc = h->small.reserve(o_sums + (size_t)BH * 8);
    if (rc) return rc;
    if (g_opt.host_zero_copy.load() != 0 && h->small.hd != nullptr) {
        const size_t rbytes = (size_t)BH * h->M * 4;
        // Where the attention launch rides behind this call (mp_lsh::Spec), the caller's rows are verified by the HOST at the
        // attention call with nothing left to hide that under: an exact compare against the handle's mirror (both in cache
        // by then: the rows were just copied from one to the other) costs a fifth of the two checksums over 190 KB the GPU
        // has just written -- so PINNED caller rows go through the mirror too (measured at cfg 1: attention_wrapper 36 -> 15 us)
        const bool spec_likely = g_opt.host_speculate.load() != 0 && h->spec.attn != nullptr && h->spec.q_host != nullptr;
        void* res_dev = spec_likely ? nullptr : h->hostmap.resolve(results, rbytes);
        bool mirror = false;
        if (res_dev == nullptr) {                 // pageable `results`: the rows go to the handle's pinned mirror
            rc = h->big.reserve(rbytes, true);
            if (rc) return rc;
            res_dev = h->big.hd;
            mirror = res_dev != nullptr;
        }
        if (res_dev != nullptr && h->hr_rows == nullptr && !h->hr_refused) {
            // the HBM copy of the handed-out rows is an accelerator too (4 B x B*H x max_length: 4 GiB at 256 heads x 2^22):
            // inside the handle's budget, or -- mp_lsh_alloc's rule -- a third of what is free now; else the staged path
            bool fits = true;
            if (h->accel_budget >= 0) fits = h->accel_used + (int64_t)rbytes <= h->accel_budget;
            else {
                size_t free_b = 0, total_b = 0;
                fits = hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)rbytes <= (double)free_b / 3.0;
            }
            h->hr_refused = !fits;
        }
        if (h->hr_refused) res_dev = nullptr;
        if (res_dev != nullptr && h->hr_rows == nullptr) {            // the HBM copy's own buffers, once
            if (hipMalloc((void**)&h->hr_rows, rbytes) != hipSuccess) { (void)hipGetLastError(); h->hr_rows = nullptr; res_dev = nullptr; }
            else if (hipMalloc((void**)&h->hr_nnz, (size_t)BH * 4) != hipSuccess) {
                (void)hipGetLastError(); (void)hipFree(h->hr_rows); h->hr_rows = nullptr; h->hr_nnz = nullptr; res_dev = nullptr;
            } else {
                h->accel_used += (int64_t)rbytes + (int64_t)BH * 4;
            }
        }
        if (res_dev != nullptr) {
            const auto t_in = std::chrono::steady_clock::now();
            char* hp = reinterpret_cast<char*>(h->small.hp);
            char* hd = reinterpret_cast<char*>(h->small.hd);
            memcpy(hp + o_codes, query, qb);
            h->lastq = reinterpret_cast<const int32_t*>(hd + o_codes);      // (get_mask reads them again)
            h->last_layer = layer_id;
            h->last_lean = false;
            // ONE launch, ONE synchronisation, no copy engine: the kernel reads the codes from the pinned block, writes
            // the ids straight into the caller's rows (where they are pinned) or the pinned mirror and the counts into the
            // pinned block -- and leaves a second copy of the rows in HBM (hr_rows / hr_nnz: buffers no other launch writes)
            // with a checksum per row, so that the attention entry of the paired store need not upload what it is handed next
            MP_HIP_CHECK(launch_lsh_retrieve(h->bounds[layer_id], h->table[layer_id],
                                             reinterpret_cast<const int32_t*>(hd + o_codes),
                                             reinterpret_cast<int32_t*>(res_dev), reinterpret_cast<int32_t*>(hd), BH,
                                             h->G, h->L, h->NB, h->M, h->R, h->idbits_dev + layer_id, h->hr_rows, h->hr_nnz,
                                             reinterpret_cast<uint32_t*>(hd + o_sums), st));
            auto t_enq = std::chrono::steady_clock::now();
            h->spec.prepared = false;
            if (g_opt.host_speculate.load() != 0 && h->hr_rows != nullptr && h->spec.attn != nullptr && h->spec.q_host != nullptr) {
                // the call waits for ITS kernel's completion word; the paired store's attention launch rides behind that word
                // and is waited for by the attention call (or by whatever synchronises the stream next).  Everything of it --
                // the snapshot of the caller's query too -- happens behind the retrieve kernel's launch, while that runs
                const unsigned int mine = h->hostflag.arm(st);
                if (lsh_spec_prepare(h, layer_id, st)) (void)lsh_spec_launch(h, layer_id, st);
                t_enq = std::chrono::steady_clock::now();
                if (!h->hostflag.reached(mine)) {
                    if (mine != 0u) g_opt.host_flag_timeouts.fetch_add(1, std::memory_order_relaxed);
                    MP_HIP_CHECK(hipStreamSynchronize(st));
                }
            } else if ((rc = h->hostflag.wait(st, g_opt.host_flag_wait.load() != 0)) != MP_OK) {
                return rc;
            }
            const auto t_got = std::chrono::steady_clock::now();
            memcpy(nnz, hp, (size_t)BH * 4);
            if (mirror) {                         // only the first nnz[h] entries of a row mean anything
                // (the rows sit max_length apart and the GPU has just written them: every row starts cold, and the hardware
                // prefetcher trains anew on each -- the next row's lines are asked for while this one is copied)
                const int32_t* rows = reinterpret_cast<const int32_t*>(h->big.hp);
                auto live = [&](int i) -> int64_t {
                    const int64_t z = nnz[i];
                    return z < 0 ? 0 : (z > h->M ? h->M : z);
                };
                const int pf_lines = g_opt.host_copy_prefetch.load(std::memory_order_relaxed);
                auto prefetch_row = [&](int i) {
#if defined(__x86_64__)
                    const char* p = reinterpret_cast<const char*>(rows + (size_t)i * h->M);
                    int64_t bytes = live(i) * 4;
                    if (bytes > (int64_t)pf_lines * 64) bytes = 8 * 64;      // a long row: its start only, the prefetcher does the rest
                    for (int64_t o = 0; o < bytes; o += 64) __builtin_prefetch(p + o, 0, 2);
#else
                    (void)i;
#endif
                };
                const bool pf = pf_lines > 0;
                if (pf && BH > 0) prefetch_row(0);
                for (int i = 0; i < BH; ++i) {
                    if (pf && i + 1 < BH) prefetch_row(i + 1);
                    const int64_t z = live(i);
                    if (z > 0) memcpy(results + (size_t)i * h->M, rows + (size_t)i * h->M, (size_t)z * 4);
                }
            }
            {   // remember what was handed out: the attention entry of the paired store may be given these very rows
                std::lock_guard<std::mutex> lock(g_host_ret_mu);
                h->host_ret.results = results;
                h->host_ret.nnz = nnz;
                h->host_ret.nnzv.assign(nnz, nnz + BH);
                const uint32_t* sums = reinterpret_cast<const uint32_t*>(hp + o_sums);
                h->host_ret.sums.assign(sums, sums + 2 * (size_t)BH);
                h->host_ret.kept = mirror ? reinterpret_cast<const int32_t*>(h->big.hp) : nullptr;
                h->host_ret.valid = true;
                g_host_ret_lsh = h;
            }
            {
                const auto t_out = std::chrono::steady_clock::now();
                auto ns = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
                    return (int)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
                };
                g_opt.host_ret_calls.fetch_add(1, std::memory_order_relaxed);
                g_opt.host_ret_ns_enqueue.fetch_add(ns(t_in, t_enq), std::memory_order_relaxed);
                g_opt.host_ret_ns_wait.fetch_add(ns(t_enq, t_got), std::memory_order_relaxed);
                g_opt.host_ret_ns_copy.fetch_add(ns(t_got, t_out), std::memory_order_relaxed);
            }
            return MP_OK;
        }
    }
    // staged: through the handle's step buffers; only the first nnz[h] entries of each row mean anything, and they
    // come back as ONE packed copy
    MP_HIP_CHECK(hipMemcpyAsync(h->last_query, query, qb, hipMemcpyHostToDevice, st));
    h->lastq = h->last_query;
    h->last_layer = layer_id;
    h->last_lean = false;
    MP_HIP_CHECK(launch_lsh_retrieve(h->bounds[layer_id], h->table[layer_id], h->last_query,
                                     h->results, h->nnz, BH, h->G, h->L, h->NB, h->M, h->R, h->idbits_dev + layer_id, nullptr, nullptr, nullptr, st));
    int32_t* d_offs = reinterpret_cast<int32_t*>(h->small.dp) + BH;           // dp: [nnz BH | offs BH + 1]
    MP_HIP_CHECK(launch_ragged_offsets(h->nnz, BH, h->M, d_offs, st));
    MP_HIP_CHECK(hipMemcpyAsync(h->small.dp, h->nnz, (size_t)BH * 4, hipMemcpyDeviceToDevice, st));
    MP_HIP_CHECK(hipMemcpyAsync(h->small.hp, h->small.dp, (size_t)(2 * BH + 1) * 4, hipMemcpyDeviceToHost, st));
    MP_HIP_CHECK(hipStreamSynchronize(st));
    const int32_t* hs = reinterpret_cast<const int32_t*>(h->small.hp);
    memcpy(nnz, hs, (size_t)BH * 4);
    const int32_t* offs = hs + BH;
    const size_t total = (size_t)offs[BH];
    if (total > 0) {
        rc = h->big.reserve(total * 4);
        if (rc) return rc;
        MP_HIP_CHECK(launch_ragged_copy(true, h->results, reinterpret_cast<int32_t*>(h->big.dp), d_offs, BH, h->M, st));
        MP_HIP_CHECK(hipMemcpyAsync(h->big.hp, h->big.dp, total * 4, hipMemcpyDeviceToHost, st));
        MP_HIP_CHECK(hipStreamSynchronize(st));
        const int32_t* packed = reinterpret_cast<const int32_t*>(h->big.hp);
        for (int i = 0; i < BH; ++i)
            if (offs[i + 1] > offs[i])
                memcpy(results + (size_t)i * h->M, packed + offs[i], (size_t)(offs[i + 1] - offs[i]) * 4);
    }
    return MP_OK;
}

int mp_lsh_clear(mp_lsh_t* h, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_lsh_clear: not allocated");
    hipStream_t st = (hipStream_t)stream;
    const size_t groups = (size_t)h->B * h->Hkv;
    for (int i = 0; i < h->layers; ++i) {
        MP_HIP_CHECK(hipMemsetAsync(h->bounds[i], 0, groups * h->L * h->NB * (size_t)(h->R + 1) * 4, st));
        MP_HIP_CHECK(hipMemsetAsync(h->table[i], 0, groups * h->L * (size_t)h->M * 4, st));
        if (!h->slots.empty())
            MP_HIP_CHECK(hipMemsetAsync(h->slots[i], 0, groups * h->L * h->NB * (size_t)h->R * h->slot_words * 4, st));
    }
    for (auto& v : h->att_ver) std::fill(v.begin(), v.end(), 0);
    MP_HIP_CHECK(hipMemsetAsync(h->att_ver_dev, 0, (size_t)h->layers * h->B * h->Hkv * 4, st));
    std::fill(h->idbits_of.begin(), h->idbits_of.end(), 17);
    MP_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->idbits_dev), 17, (size_t)h->layers, st));
    h->last_layer = -1;
    h->last_lean = false;
    return MP_OK;
}

// Key norms as a payload of the table entries (lsh.hip: lsh_attach_norms_kernel): done by the decode entry itself, the
// first time a layer is decoded after its tables or its store's norms changed (never under stream capture).  No
// synchronisation: a norm that cannot ride along (not a non-negative bf16 number) sets the slot's device flag, which the
// decode kernel reads -- the workgroups of that request then read the norms per token as before.
static int lsh_attach_norms(mp_lsh_t* h, int layer_id, int request_id, const float* kn, hipStream_t st) {
    const int rows = h->Hkv * h->L;
    int32_t* t = h->table[layer_id] + (size_t)request_id * rows * h->M;
    int* flag = h->pay_bad + ((size_t)layer_id * h->B + request_id) * h->Hkv;
    MP_HIP_CHECK(hipMemsetAsync(flag, 0, (size_t)h->Hkv * 4, st));
    MP_HIP_CHECK(launch_lsh_attach_norms(t, kn, h->Hkv, h->L, h->M, 17, flag, st));
    if (!h->slots.empty()) {                            // the direct slots copy table words: rebuild them
        int32_t* b = h->bounds[layer_id] + (size_t)request_id * rows * h->NB * (h->R + 1);
        MP_HIP_CHECK(launch_lsh_slots(t, b, h->slots[layer_id] + (size_t)request_id * rows * h->NB * h->R * h->slot_words, rows,
                                      h->NB, h->R, h->M, h->slot_log2, st));
    }
    return MP_OK;
}

int mp_lsh_get_id_bits(mp_lsh_t* h, int layer_id, int* id_bits) {
    MP_REQUIRE(h && h->allocated && id_bits, MP_ERR_STATE, "mp_lsh_get_id_bits: not allocated / null argument");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, "mp_lsh_get_id_bits: layer_id out of range");
    *id_bits = h->idbits_of[layer_id];
    return MP_OK;
}

int mp_lsh_get_mask(mp_lsh_t* h, int8_t* mask, int mem, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_lsh_get_mask: not allocated");
    MP_REQUIRE(mask, MP_ERR_INVALID, "mp_lsh_get_mask: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int BH = h->B * h->H;
    const size_t bytes = (size_t)BH * h->M;
    MP_REQUIRE(!h->last_lean, MP_ERR_STATE, "mp_lsh_get_mask: the last call was mp_decode_*_ex with MP_DECODE_NO_BYPRODUCTS: "
                                            "it left no query codes to recompute the mask from");
    if (h->last_layer < 0) {  // no retrieve since alloc/clear: the reference's mask is all zero
        if (mem == MP_MEM_DEVICE) MP_HIP_CHECK(hipMemsetAsync(mask, 0, bytes, st));
        else memset(mask, 0, bytes);
        return MP_OK;
    }
    DevBuf tmp;
    int8_t* d = mask;
    if (mem == MP_MEM_HOST) {
        MP_HIP_CHECK(tmp.alloc(bytes));
        d = tmp.as<int8_t>();
    }
    MP_HIP_CHECK(launch_lsh_mask(h->bounds[h->last_layer], h->table[h->last_layer], h->lastq,
                                 d, BH, h->G, h->L, h->NB, h->M, h->R, h->idbits_of[h->last_layer], st));
    if (mem == MP_MEM_HOST) {
        MP_HIP_CHECK(hipStreamSynchronize(st));
        MP_HIP_CHECK(hipMemcpy(mask, d, bytes, hipMemcpyDeviceToHost));
    }
    return MP_OK;
}

int mp_lsh_get_tables(mp_lsh_t* h, int layer_id, void** bounds_dev, void** table_dev) {
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_lsh_get_tables: not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, "mp_lsh_get_tables: layer_id out of range");
    if (bounds_dev) *bounds_dev = h->bounds[layer_id];
    if (table_dev) *table_dev = h->table[layer_id];
    return MP_OK;
}

int mp_lsh_get_footprint(mp_lsh_t* h, int64_t* bytes4) {
    MP_REQUIRE(h && h->allocated && bytes4, MP_ERR_STATE, "mp_lsh_get_footprint: not allocated / null argument");
    const int64_t groups = (int64_t)h->B * h->Hkv;
    bytes4[0] = groups * h->L * h->NB * (int64_t)(h->R + 1) * 4;
    bytes4[1] = groups * h->L * h->M * 4;
    bytes4[2] = h->slots.empty() ? 0 : groups * h->L * h->NB * (int64_t)h->R * h->slot_words * 4;
    bytes4[3] = h->slots.empty() ? 0 : (int64_t)h->slot_words * 4;
    return MP_OK;
}

int mp_lsh_get_footprint_ex(mp_lsh_t* h, int64_t* bytes8) {
    int rc = mp_lsh_get_footprint(h, bytes8);
    if (rc) return rc;
    const int64_t BH = (int64_t)h->B * h->H;
    bytes8[4] = h->hr_rows ? BH * h->M * 4 + BH * 4 : 0;       // HBM copy of the rows a host-mode batch_retrieve hands out (whole handle)
    bytes8[5] = (int64_t)h->big.cap + (int64_t)h->small.cap;   // pinned host memory of the host-buffer mode
    bytes8[6] = h->accel_budget;                               // what the accelerators may take (< 0: a third of free HBM, at the time)
    bytes8[7] = h->accel_used;                                 // ... and what they hold: all layers' slots + [4]
    return MP_OK;
}

int mp_lsh_get_ranges(mp_lsh_t* h, int* ranges, int* range_len) {
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_lsh_get_ranges: not allocated");
    if (ranges) *ranges = h->R;
    if (range_len) *range_len = h->range_len;
    return MP_OK;
}

// =================================================================== sparse attention

int mp_attn_create(mp_attn_t** out) {
    MP_REQUIRE(out != nullptr, MP_ERR_INVALID, "mp_attn_create: null out");
    *out = new (std::nothrow) mp_attn();
    MP_REQUIRE(*out != nullptr, MP_ERR_NOMEM, "mp_attn_create: out of host memory");
    return MP_OK;
}

// Versions of the key norms a store holds: unique across stores and fills, never 0.  A table's words may carry the
// norms of version v (mp_lsh_t::att_ver*); the decode kernel uses them only while the store still says v.
// KN_VERSION_UNKNOWN: what an append (or mp_attn_get_key_norm's writable view, through mp_attn_invalidate_norms' absence)
// leaves behind -- "these norms changed outside a fill": never equal to a version a table carries, and never packed.
constexpr uint32_t KN_VERSION_UNKNOWN = 0xffffffffu;
static std::atomic<uint32_t> g_kn_version{0};
static uint32_t next_kn_version() {
    uint32_t v = ++g_kn_version;
    while (v == 0 || v == KN_VERSION_UNKNOWN) v = ++g_kn_version;
    return v;
}
static int attn_new_version(mp_attn_t* h, int layer_id, int request_id, hipStream_t st) {
    const uint32_t v = next_kn_version();
    h->kn_ver[layer_id][request_id] = v;
    unsigned int* d = h->kn_ver_dev + ((size_t)layer_id * h->B + request_id) * h->Hkv;
    MP_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d), (int)v, (size_t)h->Hkv, st));
    return MP_OK;
}

static void attn_free(mp_attn_t* h) {
    {   // LSH handles that would enqueue this store's attention launch forget it
        std::lock_guard<std::mutex> lock(g_host_ret_mu);
        for (mp_lsh_t* l : h->spec_owners) {
            l->spec.attn = nullptr;
            l->spec.launched = false;
        }
        h->spec_owners.clear();
    }
    if (h->spec_qn) (void)hipFree(h->spec_qn);
    if (h->score_alt) (void)hipFree(h->score_alt);
    if (h->head_mz_alt) (void)hipFree(h->head_mz_alt);
    h->spec_qn = nullptr; h->score_alt = nullptr; h->head_mz_alt = nullptr;
    for (auto p : h->kv) if (p) (void)hipFree(p);
    for (auto p : h->kn) if (p) (void)hipFree(p);
    h->kv.clear();
    h->kn.clear();
    void* ptrs[] = {h->score, h->part_o, h->part_ml, h->head_mz, h->last_nnz, h->head_cnt, h->err, h->colsum, h->part_cnt, h->kn_ver_dev};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    h->score = nullptr; h->part_o = nullptr; h->part_ml = nullptr; h->head_mz = nullptr;
    h->last_nnz = nullptr; h->head_cnt = nullptr; h->err = nullptr; h->colsum = nullptr; h->part_cnt = nullptr; h->kn_ver_dev = nullptr;
    if (h->ind_rows) (void)hipFree(h->ind_rows);
    h->ind_rows = nullptr;
    h->small.release();
    h->big.release();
    h->hostmap.release();
    h->hostflag.release();
    h->allocated = false;
}

int mp_attn_destroy(mp_attn_t* h) {
    MP_ON_DEVICE(h);
    if (!h) return MP_OK;
    attn_free(h);
    delete h;
    return MP_OK;
}

int mp_attn_alloc(mp_attn_t* h, int num_layers, int num_attention_heads, int num_key_value_heads,
                  int head_dim, int batch_size, int max_length) {
    MP_REQUIRE(h, MP_ERR_INVALID, "mp_attn_alloc: null handle");
    MP_REQUIRE(!h->allocated, MP_ERR_STATE, "mp_attn_alloc: already allocated");
    MP_REQUIRE(attn_supported_head_dim(head_dim), MP_ERR_UNSUPPORTED,
               "mp_attn_alloc: head_dim must be 64 or 128");
    MP_REQUIRE(num_layers >= 1 && batch_size >= 1 && num_key_value_heads >= 1 &&
                   num_attention_heads >= num_key_value_heads &&
                   num_attention_heads % num_key_value_heads == 0 && max_length >= 1,
               MP_ERR_INVALID, "mp_attn_alloc: bad head/layer/batch counts");
    MP_REQUIRE((int64_t)batch_size * num_attention_heads <= 16384, MP_ERR_UNSUPPORTED,
               "mp_attn_alloc: more than 16384 query heads per call");
    // ONE bound, the real one: the gathers (attn_head.h: attn_head_fold, `ro[u]`) address a KV group's rows as one base +
    // a 32-bit byte offset, a token's K | V rows are 4 x head_dim bytes -> max_length x 4 head_dim <= 2^32 (2^23 tokens at
    // head_dim 128, 2^24 at 64).  Stores that pair with an LSH handle are further limited by mp_lsh_alloc's 2^22.
    MP_REQUIRE((int64_t)max_length * 4 * head_dim <= (1ll << 32), MP_ERR_INVALID,
               "mp_attn_alloc: max_length x 4 x head_dim bytes must fit 32-bit row offsets (<= 2^32)");
    h->device = current_device();
    h->layers = num_layers; h->H = num_attention_heads; h->Hkv = num_key_value_heads;
    h->D = head_dim; h->B = batch_size; h->G = h->H / h->Hkv; h->M = max_length;

    const size_t groups = (size_t)h->B * h->Hkv, BH = (size_t)h->B * h->H;
    int rc = MP_OK;
    for (int i = 0; i < num_layers && rc == MP_OK; ++i) {
        void* a = nullptr; void* b = nullptr;
        rc = alloc_zero(&a, groups * (size_t)h->M * 2 * h->D * 2);
        if (rc == MP_OK) rc = alloc_zero(&b, groups * (size_t)h->M * 4);
        h->kv.push_back((uint16_t*)a);
        h->kn.push_back((float*)b);
    }
    const size_t ms = (size_t)BH * attn_slices_per_head(h->M);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->score, BH * (size_t)h->M * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->part_o, ms * h->D * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->part_ml, ms * sizeof(float2));
    if (rc == MP_OK) rc = alloc_zero((void**)&h->head_mz, BH * sizeof(float2));
    if (rc == MP_OK) rc = alloc_zero((void**)&h->last_nnz, BH * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->head_cnt, BH * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->err, 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->part_cnt, BH * (size_t)MAX_CLUSTER * 4);
    if (rc == MP_OK) rc = alloc_zero((void**)&h->kn_ver_dev, (size_t)num_layers * groups * 4);
    if (rc == MP_OK) {   // every slot's norms (zeros) get a version no table can carry yet
        const uint32_t v = next_kn_version();
        h->kn_ver.assign((size_t)num_layers, std::vector<uint32_t>((size_t)batch_size, v));
        if (hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(h->kn_ver_dev), (int)v, (size_t)num_layers * groups) != hipSuccess)
            rc = MP_ERR_HIP;
    }
    // scratch of mp_attn_fill_offload (8 MB at Llama shapes): here, not lazily -- a first call under stream capture
    // could not allocate
    if (rc == MP_OK) rc = alloc_zero((void**)&h->colsum, (size_t)FILL_BLOCKS * h->Hkv * h->D * sizeof(double));
    if (rc != MP_OK) { attn_free(h); return rc; }
    // grid.x of the attention kernel: B*H * GX workgroups of 4 waves should fill the chip exactly
    // once (4 resident workgroups per CU at 120 VGPRs): no second dispatch round with a ragged tail.
    {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        h->cus = cus;
        h->xcd_rr = xcd_round_robin_verified();
        // with a head (or more) per CU the split-KV machinery only costs: one workgroup per head
        h->head_kernel = (BH * 2 >= cus) && (h->D == 64 || h->D == 128);
        h->grid = (int)((size_t)cus * 4 / BH);
        if (h->grid < 1) h->grid = 1;
        const int cap = (int)((h->M + 255) / 256);          // never more waves than 64-entry slices
        if (h->grid > cap) h->grid = cap;
    }
    h->allocated = true;
    return MP_OK;
}

int mp_attn_fill(mp_attn_t* h, int layer_id, int request_id, const uint16_t* k, const uint16_t* v,
                 const float* kn, int64_t n, int mem, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_attn_fill: not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, "mp_attn_fill: layer_id out of range");
    MP_REQUIRE(request_id >= 0 && request_id < h->B, MP_ERR_INVALID, "mp_attn_fill: request_id out of range");
    MP_REQUIRE(n >= 0 && n <= h->M, MP_ERR_INVALID, "mp_attn_fill: sequence longer than max_length");
    MP_REQUIRE(k && v && kn, MP_ERR_INVALID, "mp_attn_fill: null argument");
    if (n == 0) return MP_OK;
    hipStream_t st = (hipStream_t)stream;
    if (int vrc = attn_new_version(h, layer_id, request_id, st)) return vrc;
    DevBuf dk, dv, dn;
    const void *kd, *vd, *nd;
    const size_t eb = (size_t)h->Hkv * n * h->D * 2;
    int rc = stage_in(k, eb, mem, dk, &kd);
    if (rc == MP_OK) rc = stage_in(v, eb, mem, dv, &vd);
    if (rc == MP_OK) rc = stage_in(kn, (size_t)h->Hkv * n * 4, mem, dn, &nd);
    if (rc) return rc;
    uint16_t* kv = h->kv[layer_id] + (size_t)request_id * h->Hkv * h->M * 2 * h->D;
    float* knd = h->kn[layer_id] + (size_t)request_id * h->Hkv * h->M;
    MP_HIP_CHECK(launch_attn_fill((const uint16_t*)kd, (const uint16_t*)vd, (const float*)nd, h->Hkv,
                                  n, h->D, h->M, kv, knd, st));
    if (mem == MP_MEM_HOST) MP_HIP_CHECK(hipStreamSynchronize(st));
    return MP_OK;
}

// models/attnserver.py:126-175 (sparse-layer branch of fill) for one request, device buffers only

int mp_attn_fill_offload(mp_attn_t* h, mp_simhash_t* s, int layer_id, int request_id, const uint16_t* key_cache,
                         const uint16_t* value_cache, int64_t seq_len, int num_sink, int num_local,
                         uint16_t* avg_k, int16_t* codes, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_attn_fill_offload: not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, "mp_attn_fill_offload: layer_id out of range");
    MP_REQUIRE(request_id >= 0 && request_id < h->B, MP_ERR_INVALID, "mp_attn_fill_offload: request_id out of range");
    MP_REQUIRE(key_cache && value_cache && avg_k, MP_ERR_INVALID, "mp_attn_fill_offload: null argument");
    MP_REQUIRE(num_sink >= 0 && num_local >= 0 && seq_len > (int64_t)num_sink + num_local, MP_ERR_INVALID,
               "mp_attn_fill_offload: nothing to offload (seq_len <= sink + local)");
    const int64_t n = seq_len - num_sink - num_local;
    MP_REQUIRE(n <= h->M, MP_ERR_INVALID, "mp_attn_fill_offload: more offloaded tokens than max_length");
    MP_REQUIRE((h->Hkv * h->D) % 4 == 0, MP_ERR_UNSUPPORTED, "mp_attn_fill_offload: Hkv * head_dim must be a multiple of 4");
    if (codes != nullptr) {
        MP_REQUIRE(s && s->Wt, MP_ERR_STATE, "mp_attn_fill_offload: SimHash planes not set");
        MP_REQUIRE(s->D == h->D && s->device == h->device, MP_ERR_INVALID,
                   "mp_attn_fill_offload: hasher disagrees on head_dim / device");
    }
    hipStream_t st = (hipStream_t)stream;
    if (int vrc = attn_new_version(h, layer_id, request_id, st)) return vrc;
    int nblk = (int)((n + 63) / 64);
    if (nblk > FILL_BLOCKS) nblk = FILL_BLOCKS;
    uint16_t* kv = h->kv[layer_id] + (size_t)request_id * h->Hkv * h->M * 2 * h->D;
    float* knd = h->kn[layer_id] + (size_t)request_id * h->Hkv * h->M;
    MP_HIP_CHECK(launch_key_centre_fill(key_cache, value_cache, num_sink, n, h->Hkv, h->D, h->M, h->colsum, nblk,
                                        avg_k, kv, knd, st));
    if (codes != nullptr)   // key SimHash straight from the store's (centred) K rows: row stride 2D, head stride M*2D
        // (the rows' norms are in the store already -- kn: the bf16 rounding of the exact norm of the stored row -- so the
        // kernel's guard band takes them from there instead of summing every row's squares again)
        MP_HIP_CHECK(launch_simhash_keys_strided(kv, h->M * 2 * h->D, 2 * h->D, s->Wt, s->wnorm, h->Hkv, n, s->D,
                                                 s->K, s->L, codes, knd, h->M, st));
    return MP_OK;
}

static int attn_append(mp_attn_t* h, int layer_id, const uint16_t* k, const uint16_t* v, const int32_t* pos,
                       int pos_delta, const uint16_t* centre, hipStream_t st, const char* who) {
    MP_ON_DEVICE(h);
    const std::string w(who);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, w + ": not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, w + ": layer_id out of range");
    MP_REQUIRE(k && v && pos, MP_ERR_INVALID, w + ": null argument");
    // the norms of every request's KV groups change at a caller-chosen position (possibly one the LSH tables index):
    // the version of the layer's norms becomes "unknown" -- on the device by the kernel itself, in stream order and
    // in every replay of a captured graph; on the host here, so that the decode entry does not re-pack table words
    // from norms that keep changing (it packs again after the next fill)
    for (int b = 0; b < h->B; ++b) h->kn_ver[layer_id][b] = KN_VERSION_UNKNOWN;
    MP_HIP_CHECK(launch_attn_append(k, v, pos, pos_delta, centre, h->B, h->Hkv, h->D, h->M, h->kv[layer_id],
                                    h->kn[layer_id], h->kn_ver_dev + (size_t)layer_id * h->B * h->Hkv, h->err, st));
    return MP_OK;
}

int mp_attn_append(mp_attn_t* h, int layer_id, const uint16_t* k, const uint16_t* v,
                   const int32_t* pos, mp_stream_t stream) {
    return attn_append(h, layer_id, k, v, pos, 0, nullptr, (hipStream_t)stream, "mp_attn_append");
}

int mp_attn_append_centred(mp_attn_t* h, int layer_id, const uint16_t* k, const uint16_t* v,
                           const uint16_t* centre, const int32_t* pos, int pos_delta, mp_stream_t stream) {
    MP_REQUIRE(centre != nullptr, MP_ERR_INVALID, "mp_attn_append_centred: null centre");
    return attn_append(h, layer_id, k, v, pos, pos_delta, centre, (hipStream_t)stream, "mp_attn_append_centred");
}

int mp_attn_check(mp_attn_t* h, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_attn_check: not allocated");
    hipStream_t st = (hipStream_t)stream;
    int flag = 0;
    // arrival tickets must be zero between launches: a counter left standing means a cluster's tickets were lost
    // (its members ran on different XCDs and drew from different L2s) -- bit 8; the counters are reset
    MP_HIP_CHECK(launch_attn_ticket_check(h->head_cnt, h->B * h->H, h->err, st));
    MP_HIP_CHECK(hipMemcpyAsync(&flag, h->err, 4, hipMemcpyDeviceToHost, st));
    MP_HIP_CHECK(hipStreamSynchronize(st));
    if (flag) {
        MP_HIP_CHECK(hipMemsetAsync(h->err, 0, 4, st));
        if (flag & 8)
            return fail(MP_ERR_STATE, "mp_attn_check: arrival tickets of an in-launch merge were lost (the workgroups of "
                                      "a decode cluster ran on different XCDs: stream with a CU mask / partition "
                                      "change): outputs of those heads were not written; set the decode_agent_scope option");
        if (flag & 4)
            return fail(MP_ERR_STATE, "mp_attn_check: a decode cluster ran on another XCD than the placement "
                                      "observed at alloc (stream with a CU mask / partition change): its hand-off "
                                      "may have read stale partials; set the decode_agent_scope option");
        return fail(MP_ERR_DATA, "mp_attn_check: an append hit a full store (position >= max_length)");
    }
    return MP_OK;
}

// shared by sparse / full / the fused decode entry; every pointer is a device pointer here
static int attn_run(mp_attn_t* h, int layer_id, bool dense, int K, int L, uint16_t* output,
                    float* mve, const void* query, int query_dtype, const float* qn,
                    const int32_t* ind, const int32_t* nnz, hipStream_t st) {
    const int BH = h->B * h->H;
    bool head_kernel = h->head_kernel;                       // A/B overrides (mp_debug_set_option)
    if (const int o = g_opt.attn_head_kernel.load(); o >= 0) head_kernel = o != 0 && (h->D == 64 || h->D == 128);
    int grid = h->grid;
    if (const int o = g_opt.attn_gx.load(); o >= 1) grid = o;
    if (dense && g_opt.attn_dense_grouped.load() != 0) {     // K/V once per kv group (G = 1, 2, 4, 8)
        hipError_t e = hipSuccess;
        if (launch_attn_dense(h->D, h->G, query_dtype == MP_DTYPE_BF16, h->kv[layer_id], query, nnz, h->part_o,
                              h->part_ml, h->head_cnt, output, mve, h->head_mz, h->score, BH, h->M, h->cus, st, &e)) {
            MP_HIP_CHECK(e);
            h->lastz = nnz;
            h->score_state = 1;
            h->seg_cnt = nullptr;
            h->seg_R = 1;
            return MP_OK;
        }
    }
    MP_HIP_CHECK(launch_attn_sparse(h->D, dense, query_dtype == MP_DTYPE_BF16, h->kv[layer_id],
                                    h->kn[layer_id], query, qn, ind, nnz, h->part_o, h->part_ml,
                                    h->head_cnt, output, mve, h->head_mz, h->score, BH, h->G, h->M, K, L,
                                    grid, head_kernel, st));
    h->lastz = nnz;
    h->score_state = 1;
    h->seg_cnt = nullptr;
    h->seg_R = 1;
    return MP_OK;
}

static int attn_entry(mp_attn_t* h, int layer_id, bool dense, int K, int L, uint16_t* output,
                      float* mve, const void* query, int query_dtype, const float* qn,
                      const int32_t* ind, const int32_t* nnz, int mem, hipStream_t st,
                      const char* who) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, std::string(who) + ": not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, std::string(who) + ": layer_id out of range");
    MP_REQUIRE(output && mve && query && nnz && (dense || (qn && ind)), MP_ERR_INVALID,
               std::string(who) + ": null argument");
    MP_REQUIRE(query_dtype == MP_DTYPE_BF16 || query_dtype == MP_DTYPE_F32, MP_ERR_INVALID,
               std::string(who) + ": query dtype must be bf16 or f32");
    MP_REQUIRE(dense || (K >= 1 && L >= 2), MP_ERR_INVALID, std::string(who) + ": need K >= 1, L >= 2");
    const int BH = h->B * h->H;
    if (mem == MP_MEM_DEVICE)
        return attn_run(h, layer_id, dense, K, L, output, mve, query, query_dtype, qn, ind, nnz, st);
    // host buffers: ONE block of small arguments up (q | qn | nnz | offsets), the index rows packed to their first
    // nnz[h] entries in ONE copy, unpacked on device into the handle's [BH][M] rows; (out | mve) come back in ONE
    // copy.  All staging is pinned and owned by the handle.
    const AttnHostLayout lo = attn_host_layout(BH, h->D, query_dtype);
    const size_t qbytes = lo.qbytes, o_q = lo.o_q, o_qn = lo.o_qn, o_nnz = lo.o_nnz, o_offs = lo.o_offs, o_out = lo.o_out,
                 o_mve = lo.o_mve, o_sqn = lo.o_sqn, o_end = lo.o_end;
    int rc = h->small.reserve(o_end);
    if (rc) return rc;
    const unsigned long long seq_at_entry = h->host_seq++;      // this call owns the pinned block from here on
    char* hp = reinterpret_cast<char*>(h->small.hp);
    char* dp = reinterpret_cast<char*>(h->small.dp);
    memcpy(hp + o_q, query, qbytes);
    if (!dense) memcpy(hp + o_qn, qn, (size_t)BH * 4);
    memcpy(hp + o_nnz, nnz, (size_t)BH * 4);
    int32_t* offs = reinterpret_cast<int32_t*>(hp + o_offs);
    size_t total = 0;
    for (int i = 0; i < BH; ++i) {
        offs[i] = (int32_t)total;
        int64_t z = nnz[i];
        z = z < 0 ? 0 : (z > h->M ? h->M : z);
        total += dense ? 0 : (size_t)z;
    }
    offs[BH] = (int32_t)total;
    MP_REQUIRE(total <= (size_t)INT32_MAX, MP_ERR_UNSUPPORTED, std::string(who) + ": more than 2^31 index entries in one call");
    // The rows a paired LSH handle has just handed out (round 4).  The reference's caller passes the `results` /
    // `nnz` of batch_retrieve straight on as `ind` / `nnz` (models/attnserver.py:299-300); the handle that produced them
    // still holds the same rows in HBM (in buffers only that retrieve writes).  They are recognised by the caller's
    // pointers, the counts and a comparison of every live row -- exact where the handle kept the rows it wrote (pageable
    // caller rows: its pinned mirror), two checksums where the kernel wrote the caller's pinned rows -- a caller that
    // edited `ind` in between is served its edit through the upload below -- and then only (q | qn) cross PCIe: no
    // index upload, no second copy of the rows.
    if (!dense && g_opt.host_zero_copy.load() != 0 && h->small.hd != nullptr) {
        // The pairing is looked up under the mutex and USED under a use count taken there: the paired handle's destroy / next
        // host-mode retrieve on another thread (host_ret_forget) waits until this call no longer reads its state or its HBM
        // rows (`l` used to be dereferenced after the lock was released; r05: the mutex itself used to be held
        // over the launch and the wait, serialising every host-mode call of the process).
        mp_lsh_t* l = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_host_ret_mu);
            l = g_host_ret_lsh;
            if (l != nullptr && !(l->host_ret.valid && l->allocated && l->device == h->device && l->B * l->H == BH &&
                                  l->M == h->M && l->hr_rows != nullptr && l->host_ret.results == ind && l->host_ret.nnz == nnz &&
                                  memcmp(l->host_ret.nnzv.data(), nnz, (size_t)BH * 4) == 0))
                l = nullptr;
            if (l != nullptr) ++l->ret_users;          // the handle's destroy / next host-mode retrieve wait for this call
        }
        struct Release {                               // ... until here, on every way out of the block
            mp_lsh_t* l;
            ~Release() {
                if (l == nullptr) return;
                std::lock_guard<std::mutex> lock(g_host_ret_mu);
                --l->ret_users;
                g_host_ret_cv.notify_all();
            }
        } release{l};
        // the rows are those the retrieve handed out?  exact where the handle kept them, checksums where the kernel wrote
        // the caller's pinned rows
        auto rows_untouched = [&]() {
            bool same = true;
            const int32_t* kept = l->host_ret.kept;
            for (int i = 0; i < BH && same; ++i) {
                int64_t z = nnz[i];
                z = z < 0 ? 0 : (z > h->M ? h->M : z);
                if (kept != nullptr) {
                    same = z == 0 || memcmp(ind + (size_t)i * h->M, kept + (size_t)i * h->M, (size_t)z * 4) == 0;
                } else {
                    uint32_t s1, s2;
                    host_row_sum(ind + (size_t)i * h->M, z, &s1, &s2);
                    same = s1 == l->host_ret.sums[2 * i] && s2 == l->host_ret.sums[2 * i + 1];
                }
            }
            return same;
        };
        // what this call looks like, for the paired handle's next retrieve (mp_lsh::Spec): only with a pinned query tensor
        auto remember_call = [&]() {
            if (g_opt.host_speculate.load() == 0 || h->hostmap.resolve(query, qbytes) == nullptr) return;
            std::lock_guard<std::mutex> lock(g_host_ret_mu);
            if (l->spec.attn != h) {
                if (l->spec.attn != nullptr) {
                    auto& ow = l->spec.attn->spec_owners;
                    for (size_t i = 0; i < ow.size(); ++i)
                        if (ow[i] == l) { ow.erase(ow.begin() + i); break; }
                }
                l->spec.attn = h;
                h->spec_owners.push_back(l);
            }
            l->spec.q_host = query;
            l->spec.q_dtype = query_dtype;
            l->spec.K = K;
            l->spec.L = L;
        };
        if (l != nullptr && l->spec.launched) {
            // The retrieve that handed these rows out has already run this store's attention launch behind its own kernel.
            // Is this the call it assumed?  Same store, nothing else on the store's pinned block since, same layer / K / L /
            // dtype, the same query tensor holding the same bytes, the caller's ||q|| within rounding of the kernel's.
            l->spec.launched = false;                              // (a launch serves one call)
            bool hit = l->spec.attn == h && l->spec.attn_seq == seq_at_entry && l->spec.layer == layer_id && l->spec.K == K &&
                       l->spec.L == L && l->spec.q_dtype == query_dtype && l->spec.q_host == query &&
                       l->spec.q_bytes == qbytes && memcmp(query, hp + lo.o_qsnap, qbytes) == 0;
            if (hit && rows_untouched()) {
                if (!h->hostflag.reached(l->spec.done_flag)) {
                    if (l->spec.done_flag != 0u) g_opt.host_flag_timeouts.fetch_add(1, std::memory_order_relaxed);
                    MP_HIP_CHECK(hipStreamSynchronize(l->spec.stream));
                }
                const float* sq = reinterpret_cast<const float*>(hp + o_sqn);        // (the launch's norms: written behind the retrieve)
                for (int i = 0; i < BH && hit; ++i) hit = fabsf(qn[i] - sq[i]) <= 2e-6f * fabsf(sq[i]);
            } else {
                hit = false;
            }
            if (hit) {
                g_opt.host_spec_hits.fetch_add(1, std::memory_order_relaxed);
                g_opt.host_fast_hits.fetch_add(1, std::memory_order_relaxed);
                std::swap(h->score, h->score_alt);                 // the launch's logits and (max, Z) are the last call's now
                std::swap(h->head_mz, h->head_mz_alt);
                h->score_state = 1;
                h->seg_cnt = nullptr;
                h->seg_R = 1;
                h->lastz_host.assign(nnz, nnz + BH);
                h->lastz = nullptr;
                memcpy(output, hp + o_out, (size_t)BH * h->D * 2);
                memcpy(mve, hp + o_mve, (size_t)2 * BH * 4);
                remember_call();
                return MP_OK;
            }
            g_opt.host_spec_misses.fetch_add(1, std::memory_order_relaxed);
            // the launch's outputs are dropped: what this call enqueues overwrites them in stream order -- on another
            // stream only behind a synchronisation
            if (l->spec.stream != st) MP_HIP_CHECK(hipStreamSynchronize(l->spec.stream));
        }
        if (l != nullptr) {
            // Launched BEFORE the rows are verified: the attention kernel works on the rows + counts the retrieve kernel left
            // in HBM (l->hr_rows / hr_nnz: written by that retrieve and by nothing else) while the host compares the
            // caller's rows with what the retrieve handed out -- exactly (memcmp against the handle's pinned mirror, which
            // still holds the rows as the kernel wrote them) where the caller's rows are pageable, by the two checksums
            // where the kernel wrote the caller's pinned rows directly.  A row that was edited (or a stale pairing) is
            // found before anything is handed back: the launch is waited for, its outputs dropped, and the upload path
            // below serves the caller's rows.  Few heads (B*H <= 64: cfg 1 / 4): ONE launch, every workgroup reads its
            // (q | qn) straight from the pinned block, two PCIe reads under its index loads (-5 us per call at cfg 1
            // against a relay launch in front).  Many heads (cfg 2 / 3: 256 heads x the workgroups of a head, 256 bytes
            // each) would queue on PCIe -- measured +23 us -- so there a one-workgroup relay brings (q | qn) into HBM first.
            char* hd = reinterpret_cast<char*>(h->small.hd);
            const bool direct = BH <= 64;
            if (!direct) MP_HIP_CHECK(launch_relay(hd, dp, o_nnz, st));
            const char* qsrc = direct ? hd : dp;
            rc = attn_run(h, layer_id, false, K, L, reinterpret_cast<uint16_t*>(hd + o_out),
                          reinterpret_cast<float*>(hd + o_mve), qsrc + o_q, query_dtype,
                          reinterpret_cast<const float*>(qsrc + o_qn), l->hr_rows, l->hr_nnz, st);
            if (rc) return rc;
            const bool same = rows_untouched();
            if ((rc = h->hostflag.wait(st, g_opt.host_flag_wait.load() != 0)) != MP_OK) return rc;
            if (same) {
                g_opt.host_fast_hits.fetch_add(1, std::memory_order_relaxed);
                h->lastz_host.assign(nnz, nnz + BH);
                h->lastz = nullptr;
                memcpy(output, hp + o_out, (size_t)BH * h->D * 2);
                memcpy(mve, hp + o_mve, (size_t)2 * BH * 4);
                remember_call();
                return MP_OK;
            }
            g_opt.host_fast_edited.fetch_add(1, std::memory_order_relaxed);
        } else {
            g_opt.host_fast_unpaired.fetch_add(1, std::memory_order_relaxed);
        }
    }
    // Zero copy: ONE launch brings (q | qn | nnz) and the first nnz[h] entries of every index row into HBM with
    // coalesced reads over PCIe -- straight from the caller's rows where they are pinned (HostMap),
    // else from the handle's pinned block, into which the host packs the live entries -- and the attention
    // kernel writes (out | mve) straight into the pinned block: two or three launches, ONE synchronisation, no copy engine.
    if (g_opt.host_zero_copy.load() != 0 && h->small.hd != nullptr) {
        char* hd = reinterpret_cast<char*>(h->small.hd);
        bool ok = true;
        if (dense) {
            MP_HIP_CHECK(launch_relay(hd, dp, o_offs, st));
        } else {
            if (h->ind_rows == nullptr) MP_HIP_CHECK(hipMalloc((void**)&h->ind_rows, (size_t)BH * h->M * 4));
            const void* ind_dev = h->hostmap.resolve(ind, (size_t)BH * h->M * 4);
            if (ind_dev != nullptr) {
                int64_t longest = 0;                                     // one PCIe round trip per thread: blocks by the longest row
                for (int i = 0; i < BH; ++i) longest = nnz[i] > longest ? nnz[i] : longest;
                longest = longest > h->M ? h->M : longest;
                int gx = (int)((longest + 255) / 256);
                gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
                MP_HIP_CHECK(launch_host_rows(reinterpret_cast<const int32_t*>(ind_dev),
                                              reinterpret_cast<const int32_t*>(hd + o_nnz), h->ind_rows, h->M, BH, hd, dp,
                                              o_offs, gx, st));
            } else {                                                     // pageable rows: packed into the pinned block by the host
                rc = h->big.reserve(total ? total * 4 : 4);
                if (rc) return rc;
                ok = h->big.hd != nullptr;
                if (ok) {
                    int32_t* packed = reinterpret_cast<int32_t*>(h->big.hp);
                    for (int i = 0; i < BH; ++i)
                        if (offs[i + 1] > offs[i])
                            memcpy(packed + offs[i], ind + (size_t)i * h->M, (size_t)(offs[i + 1] - offs[i]) * 4);
                    MP_HIP_CHECK(launch_relay(hd, dp, o_out, st));       // (q | qn | nnz | offsets)
                    if (total > 0)
                        MP_HIP_CHECK(launch_ragged_copy(false, h->ind_rows, reinterpret_cast<int32_t*>(h->big.hd),
                                                        reinterpret_cast<const int32_t*>(dp + o_offs), BH, h->M, st));
                }
            }
        }
        if (ok) {
            rc = attn_run(h, layer_id, dense, K, L, reinterpret_cast<uint16_t*>(hd + o_out),
                          reinterpret_cast<float*>(hd + o_mve), dp + o_q, query_dtype,
                          reinterpret_cast<const float*>(dp + o_qn), dense ? nullptr : h->ind_rows,
                          reinterpret_cast<const int32_t*>(dp + o_nnz), st);
            if (rc) return rc;
            if ((rc = h->hostflag.wait(st, g_opt.host_flag_wait.load() != 0)) != MP_OK) return rc;
            memcpy(output, hp + o_out, (size_t)BH * h->D * 2);
            memcpy(mve, hp + o_mve, (size_t)2 * BH * 4);
            return MP_OK;                                         // (lastz = dp + o_nnz: valid until the next host call)
        }
    }
    MP_HIP_CHECK(hipMemcpyAsync(dp, hp, o_out, hipMemcpyHostToDevice, st));
    MP_HIP_CHECK(hipMemcpyAsync(h->last_nnz, dp + o_nnz, (size_t)BH * 4, hipMemcpyDeviceToDevice, st));   // outlives the call (get_score)
    const int32_t* indd = nullptr;
    if (!dense) {
        if (h->ind_rows == nullptr) MP_HIP_CHECK(hipMalloc((void**)&h->ind_rows, (size_t)BH * h->M * 4));
        if (total > 0) {
            rc = h->big.reserve(total * 4);
            if (rc) return rc;
            int32_t* packed = reinterpret_cast<int32_t*>(h->big.hp);
            for (int i = 0; i < BH; ++i)
                if (offs[i + 1] > offs[i])
                    memcpy(packed + offs[i], ind + (size_t)i * h->M, (size_t)(offs[i + 1] - offs[i]) * 4);
            MP_HIP_CHECK(hipMemcpyAsync(h->big.dp, h->big.hp, total * 4, hipMemcpyHostToDevice, st));
            MP_HIP_CHECK(launch_ragged_copy(false, h->ind_rows, reinterpret_cast<int32_t*>(h->big.dp),
                                            reinterpret_cast<const int32_t*>(dp + o_offs), BH, h->M, st));
        }
        indd = h->ind_rows;
    }
    rc = attn_run(h, layer_id, dense, K, L, reinterpret_cast<uint16_t*>(dp + o_out), reinterpret_cast<float*>(dp + o_mve),
                  dp + o_q, query_dtype, reinterpret_cast<const float*>(dp + o_qn), indd, h->last_nnz, st);
    if (rc) return rc;
    MP_HIP_CHECK(hipMemcpyAsync(hp + o_out, dp + o_out, o_end - o_out, hipMemcpyDeviceToHost, st));
    MP_HIP_CHECK(hipStreamSynchronize(st));
    memcpy(output, hp + o_out, (size_t)BH * h->D * 2);
    memcpy(mve, hp + o_mve, (size_t)2 * BH * 4);
    return MP_OK;
}

int mp_attn_sparse(mp_attn_t* h, int layer_id, int K, int L, uint16_t* output,
                   float* max_value_expsum, const void* query, int query_dtype,
                   const float* query_norm, const int32_t* ind, const int32_t* nnz, int mem,
                   mp_stream_t stream) {
    return attn_entry(h, layer_id, false, K, L, output, max_value_expsum, query, query_dtype,
                      query_norm, ind, nnz, mem, (hipStream_t)stream, "mp_attn_sparse");
}

int mp_attn_full(mp_attn_t* h, int layer_id, uint16_t* output, float* max_value_expsum,
                 const void* query, int query_dtype, const int32_t* nnz, int mem,
                 mp_stream_t stream) {
    return attn_entry(h, layer_id, true, 0, 0, output, max_value_expsum, query, query_dtype, nullptr,
                      nullptr, nnz, mem, (hipStream_t)stream, "mp_attn_full");
}

int mp_attn_clear(mp_attn_t* h, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_attn_clear: not allocated");
    hipStream_t st = (hipStream_t)stream;
    const size_t groups = (size_t)h->B * h->Hkv, BH = (size_t)h->B * h->H;
    for (int i = 0; i < h->layers; ++i) {
        MP_HIP_CHECK(hipMemsetAsync(h->kv[i], 0, groups * (size_t)h->M * 2 * h->D * 2, st));
        MP_HIP_CHECK(hipMemsetAsync(h->kn[i], 0, groups * (size_t)h->M * 4, st));
    }
    MP_HIP_CHECK(hipMemsetAsync(h->score, 0, BH * (size_t)h->M * 4, st));
    {
        const uint32_t nv = next_kn_version();
        for (auto& v : h->kn_ver) std::fill(v.begin(), v.end(), nv);
        MP_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->kn_ver_dev), (int)nv,
                                       (size_t)h->layers * h->B * h->Hkv, st));
    }
    h->score_state = 0;
    return MP_OK;
}

// Debug: device buffer (>= 64 u64) receiving 100 MHz wall-clock stamps at the phase boundaries of
// workgroup 0 of the hot kernels (slots: simhash 0-3, retrieve 16-21, attention 32-38); NULL = off.
int mp_debug_set_stamp_buffer(void* dev_u64x64) {
    g_stamp = reinterpret_cast<unsigned long long*>(dev_u64x64);
    return MP_OK;
}

int mp_debug_xcd_round_robin(void) { return xcd_round_robin_verified() ? 1 : 0; }

int mp_debug_set_option(const char* name, int value) {
    if (name && !strcmp(name, "stamp_stride")) {      // every workgroup of the decode kernel records (lsh.hip)
        MP_HIP_CHECK(set_stamp_stride(value));
        return MP_OK;
    }
    if (name && !strcmp(name, "simhash_exact_norm")) {   // 1: the fused hash normalises the query row by the exact sequence always
        set_exact_norm(value);
        return MP_OK;
    }
    if (name && !strcmp(name, "decode_slot_log2")) {     // 3 / 4 / 5: 32- / 64- / 128-byte direct slots whatever the mean piece; 0 = auto.
        set_slot_log2(value);                            // Process-wide and read at alloc, build and launch: set it before the handles exist
        return MP_OK;
    }
    std::atomic<int>* o = debug_option(name);
    MP_REQUIRE(o != nullptr, MP_ERR_INVALID, std::string("mp_debug_set_option: unknown option '") + (name ? name : "(null)") + "'");
    o->store(value);
    return MP_OK;
}

int mp_debug_get_option(const char* name, int* value) {
    if (name && value && !strcmp(name, "decode_slot_log2")) {
        *value = get_slot_log2();
        return MP_OK;
    }
    std::atomic<int>* o = debug_option(name);
    MP_REQUIRE(o != nullptr && value != nullptr, MP_ERR_INVALID, "mp_debug_get_option: unknown option or null value");
    *value = o->load();
    return MP_OK;
}

int mp_attn_get_kv(mp_attn_t* h, int layer_id, void** key_dev, void** value_dev,
                   int64_t* row_stride_elems) {
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_attn_get_kv: not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, "mp_attn_get_kv: layer_id out of range");
    if (key_dev) *key_dev = h->kv[layer_id];
    if (value_dev) *value_dev = h->kv[layer_id] + h->D;
    if (row_stride_elems) *row_stride_elems = 2 * (int64_t)h->D;
    return MP_OK;
}

int mp_attn_get_footprint(mp_attn_t* h, int64_t* bytes2) {
    MP_REQUIRE(h && h->allocated && bytes2, MP_ERR_STATE, "mp_attn_get_footprint: not allocated / null argument");
    const int64_t groups = (int64_t)h->B * h->Hkv;
    bytes2[0] = groups * h->M * 2 * h->D * 2;
    bytes2[1] = groups * h->M * 4;
    return MP_OK;
}

int mp_attn_get_key_norm(mp_attn_t* h, int layer_id, void** kn_dev) {
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_attn_get_key_norm: not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, "mp_attn_get_key_norm: layer_id out of range");
    if (kn_dev) *kn_dev = h->kn[layer_id];
    return MP_OK;
}

int mp_attn_invalidate_norms(mp_attn_t* h, int layer_id, int request_id, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_attn_invalidate_norms: not allocated");
    MP_REQUIRE(layer_id >= 0 && layer_id < h->layers, MP_ERR_INVALID, "mp_attn_invalidate_norms: layer_id out of range");
    MP_REQUIRE(request_id >= 0 && request_id < h->B, MP_ERR_INVALID, "mp_attn_invalidate_norms: request_id out of range");
    return attn_new_version(h, layer_id, request_id, (hipStream_t)stream);
}

int mp_attn_get_score(mp_attn_t* h, void** score_dev, mp_stream_t stream) {
    MP_ON_DEVICE(h);
    MP_REQUIRE(h && h->allocated, MP_ERR_STATE, "mp_attn_get_score: not allocated");
    MP_REQUIRE(score_dev, MP_ERR_INVALID, "mp_attn_get_score: null argument");
    hipStream_t st = (hipStream_t)stream;
    MP_REQUIRE(h->score_state != 3, MP_ERR_STATE, "mp_attn_get_score: the last call was mp_decode_*_ex with "
                                                  "MP_DECODE_NO_BYPRODUCTS: it left no logits");
    if (h->score_state == 1) {
        if (h->seg_cnt != nullptr && h->seg_R > 1) {   // one-launch decode: R per-member segments -> one list
            MP_HIP_CHECK(launch_lsh_compact(reinterpret_cast<uint32_t*>(h->score), h->seg_cnt, h->B * h->H,
                                            h->seg_R, h->M, st));
            h->seg_cnt = nullptr;
            h->seg_R = 1;
        }
        if (h->lastz == nullptr) {                      // host-buffer fast path: the counts are still on the host
            MP_REQUIRE((int)h->lastz_host.size() == h->B * h->H, MP_ERR_STATE, "mp_attn_get_score: no counts of the last call");
            MP_HIP_CHECK(hipMemcpyAsync(h->last_nnz, h->lastz_host.data(), h->lastz_host.size() * 4, hipMemcpyHostToDevice, st));
            MP_HIP_CHECK(hipStreamSynchronize(st));     // (the pageable source may be rewritten by the next call)
            h->lastz = h->last_nnz;
        }
        MP_HIP_CHECK(launch_attn_normalize(h->score, h->lastz, h->head_mz, h->B * h->H, h->M, st));
        h->score_state = 2;
    }
    *score_dev = h->score;
    return MP_OK;
}

// =================================================================== table build (with the store's norms)

static int lsh_build_entry(mp_lsh_t* h, mp_attn_t* attn, int layer_id, int request_id, const int16_t* codes, int64_t n,
                           int mem, hipStream_t st, const char* who) {
    MP_ON_DEVICE(h);
    int rc = lsh_check_slot(h, layer_id, request_id, n, who);
    if (rc) return rc;
    MP_REQUIRE(codes || n == 0, MP_ERR_INVALID, std::string(who) + ": null argument");
    const int rows = h->Hkv * h->L;
    DevBuf dc;
    const void* c = nullptr;
    if (n > 0) {           // (an empty request -- n = 0: empty tables -- has no codes to stage)
        rc = stage_in(codes, (size_t)rows * n * 2, mem, dc, &c);
        if (rc) return rc;
    }
    int32_t* b = h->bounds[layer_id] + (size_t)request_id * rows * h->NB * (h->R + 1);
    int32_t* t = h->table[layer_id] + (size_t)request_id * rows * h->M;
    if ((rc = lsh_set_version(h, layer_id, request_id, 0, st)) != MP_OK) return rc;   // the rows are being rewritten
    if (n > (1 << 17) && (rc = lsh_widen(h, layer_id, request_id, st)) != MP_OK) return rc;
    // Packed build: the store already holds this request's key norms (the reference's order: norms and K/V at
    // models/attnserver.py:146, 174, the tables at :178-193) -> the sort writes  id | norm << 17  itself, the direct slots
    // are built ONCE from the packed words, and the first decode of the layer finds nothing to do (round 3: a second
    // sweep over the table + a second build of the slots, +1.4 ms per layer at cfg 1 and a first-token latency spike).
    const float* kn = nullptr;
    int* flag = h->pay_bad + ((size_t)layer_id * h->B + request_id) * h->Hkv;
    uint32_t ver = 0;
    if (attn != nullptr) {
        MP_REQUIRE(attn->allocated && attn->device == h->device && attn->B == h->B && attn->Hkv == h->Hkv &&
                       attn->M == h->M && layer_id < attn->layers,
                   MP_ERR_INVALID, std::string(who) + ": the attention store disagrees on device / B / Hkv / max_length / layers");
        ver = attn->kn_ver[layer_id][request_id];
        if (h->idbits_of[layer_id] != 0 && ver != KN_VERSION_UNKNOWN && g_opt.decode_kn_payload.load() != 0)
            kn = attn->kn[layer_id] + (size_t)request_id * attn->Hkv * attn->M;
    }
    // The build ranks the tokens of a bucket by the order in which the LDS serves the lanes of one atomic instruction -- lane
    // order on gfx950, observed, not promised -- and VERIFIES every bucket run it writes (err bit 64).  A run that does not ascend
    // means: rebuild the request with the exact ranking (once; counted in `build_rank_fallbacks`).
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool exact = attempt == 1 || g_opt.build_rank_exact.load() != 0;
        bool packed = false, cut = false, misranked = false;
        if (kn != nullptr) MP_HIP_CHECK(hipMemsetAsync(flag, 0, (size_t)h->Hkv * 4, st));
        MP_HIP_CHECK(launch_lsh_build((const int16_t*)c, rows, n, h->NB, h->M, h->R, b, t, h->err, kn, h->L, 17, flag, &packed, &cut,
                                      exact, st));
        // (the staged build writes the sub-bounds itself, round 5; the direct variant for K >= 14 leaves them to the search)
        if (!cut) MP_HIP_CHECK(launch_lsh_subbounds(t, b, rows, h->NB, h->R, h->M, packed ? 17 : 0, st));
        if (!h->slots.empty())
            MP_HIP_CHECK(launch_lsh_slots(t, b, h->slots[layer_id] + (size_t)request_id * rows * h->NB * h->R * h->slot_words, rows,
                                          h->NB, h->R, h->M, h->slot_log2, st));
        if (packed && (rc = lsh_set_version(h, layer_id, request_id, ver, st)) != MP_OK) return rc;
        rc = lsh_read_err(h, st, who, nullptr, nullptr, &misranked);
        if (rc == MP_OK && !exact && !misranked) {      // test hook: behave as if the check had failed (the rebuild cannot be
            int left = g_opt.build_rank_inject.load();  // provoked on gfx950, where the LDS does serve the lanes in order)
            while (left > 0 && !g_opt.build_rank_inject.compare_exchange_weak(left, left - 1)) {}
            misranked = left > 0;
        }
        if (rc != MP_OK || !misranked) return rc;
        MP_REQUIRE(!exact, MP_ERR_DATA, std::string(who) + ": the exact table build reported a mis-ranked bucket");
        g_opt.build_rank_fallbacks.fetch_add(1, std::memory_order_relaxed);
        if ((rc = lsh_set_version(h, layer_id, request_id, 0, st)) != MP_OK) return rc;     // rewritten again
    }
    return MP_OK;
}

int mp_lsh_build_with_norms(mp_lsh_t* h, mp_attn_t* attn, int layer_id, int request_id, const int16_t* codes, int64_t n,
                            int mem, mp_stream_t stream) {
    MP_REQUIRE(attn != nullptr, MP_ERR_INVALID, "mp_lsh_build_with_norms: null attention store");
    return lsh_build_entry(h, attn, layer_id, request_id, codes, n, mem, (hipStream_t)stream, "mp_lsh_build_with_norms");
}

// =================================================================== fused decode step

// shared by mp_decode_sparse_layer (win == nullptr) and mp_decode_layer_window
static int decode_layer(mp_simhash_t* s, mp_lsh_t* lsh, mp_attn_t* attn, mp_attn_t* win, int layer_id,
                        const uint16_t* q, const int32_t* win_len, uint16_t* output, float* max_value_expsum,
                        int32_t* nnz_out, unsigned int flags, hipStream_t st, const char* who) {
    const std::string w(who);
    MP_REQUIRE((flags & ~(unsigned int)MP_DECODE_NO_BYPRODUCTS) == 0u, MP_ERR_INVALID, w + ": unknown flag");
    MP_ON_DEVICE(attn);
    MP_REQUIRE(s && s->Wt, MP_ERR_STATE, w + ": SimHash planes not set");
    MP_REQUIRE(lsh && lsh->allocated && attn && attn->allocated, MP_ERR_STATE, w + ": handles not allocated");
    MP_REQUIRE(s->device == attn->device && lsh->device == attn->device && (!win || win->device == attn->device),
               MP_ERR_INVALID, w + ": the handles live on different devices");
    MP_REQUIRE(q && output && max_value_expsum, MP_ERR_INVALID, w + ": null argument");
    MP_REQUIRE(lsh->B == attn->B && lsh->H == attn->H && lsh->Hkv == attn->Hkv && lsh->M == attn->M &&
                   s->K == lsh->K && s->L == lsh->L && s->D == attn->D,
               MP_ERR_INVALID, w + ": handles disagree on B/H/Hkv/M/K/L/D");
    MP_REQUIRE(layer_id >= 0 && layer_id < lsh->layers && layer_id < attn->layers, MP_ERR_INVALID,
               w + ": layer_id out of range");
    const int BH = lsh->B * lsh->H;
    // (the rows a host-mode batch_retrieve handed out live in buffers of their own, hr_rows / hr_nnz: this entry's
    // by-products -- eager or replayed from a graph -- no longer touch them)
    lsh->lastq = lsh->codes;
    lsh->last_layer = layer_id;
    lsh->last_lean = false;
    const bool two_launch = g_opt.decode_two_launch.load() != 0;                // A/B switch
    const bool fused = !two_launch && lsh_decode_supported(lsh->M, lsh->L, s->D, lsh->R);
    if (win != nullptr) {
        MP_REQUIRE(win->allocated && win_len, MP_ERR_INVALID, w + ": window store not allocated / null win_len");
        MP_REQUIRE(win->B == attn->B && win->H == attn->H && win->Hkv == attn->Hkv && win->D == attn->D &&
                       layer_id < win->layers,
                   MP_ERR_INVALID, w + ": window store disagrees on B/H/Hkv/D/layers");
        MP_REQUIRE(fused, MP_ERR_UNSUPPORTED,
                   w + ": the one-launch form is not available for this shape; use full_attention + "
                       "mp_decode_sparse_layer + mp_merge_state");
    }
    if (fused) {
        // a-1 .. a-12 (models/attnserver.py:264-300) in ONE launch: hash -> retrieve -> attention of a
        // head inside a cluster of lsh->R workgroups, each owning one token range of the head's tables;
        // the selected ids stay in LDS.
        // A/B (north_star names MFMA for the query projection): the hash as simhash_query_kernel's own launch
        const bool mfma_hash = win == nullptr && g_opt.decode_mfma_hash.load() != 0;
        // the planes split over the members of a cluster, sign bits exchanged through the XCD's L2 (lsh.hip); the
        // launcher keeps it to clusters that share an XCD.  -1 = auto: on
        int xmode = g_opt.decode_split_hash.load();
        if (xmode < 0) xmode = 1;
        if (mfma_hash)
            MP_HIP_CHECK(launch_simhash_query(q, s->Wt, s->wnorm, BH, s->D, s->K, s->L, lsh->codes, lsh->qnorm,
                                              nullptr, st));
        // Key norms as a payload of the table entries (while the layer's ids fit 17 bits; A/B: decode_kn_payload = 0): the
        // first decode of a layer after its tables or its store's norms changed packs them (two kernels per request,
        // once; never under stream capture).  Whether a KV group's payload is USED is decided by the kernel from device
        // words written in stream order (the version the rows carry == the version the store holds, no norm refused):
        // a graph captured before the packing uses it afterwards, one replayed after a refill does not.
        const bool kn_payload = lsh->idbits_of[layer_id] != 0 && g_opt.decode_kn_payload.load() != 0;
        if (kn_payload) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
            for (int b = 0; b < lsh->B && !capturing; ++b) {
                if (lsh->att_ver[layer_id][b] == attn->kn_ver[layer_id][b]) continue;
                if (attn->kn_ver[layer_id][b] == KN_VERSION_UNKNOWN) continue;   // norms changed outside a fill: never packed
                int rc = lsh_attach_norms(lsh, layer_id, b, attn->kn[layer_id] + (size_t)b * attn->Hkv * attn->M, st);
                if (rc == MP_OK) rc = lsh_set_version(lsh, layer_id, b, attn->kn_ver[layer_id][b], st);
                if (rc) return rc;
            }
        }
        const size_t goff = (size_t)layer_id * lsh->B * lsh->Hkv;
        // MP_DECODE_NO_BYPRODUCTS: the launch writes `output`, `max_value_expsum` and the counts, nothing else (no codes, no
        // ||q||, no result rows, no logits) and hands its selected ids to the gather unordered (lsh.hip: LEAN).  The one-launch
        // form only (the other forms ARE the by-products); the MFMA-hash A/B keeps them too.
        bool lean = (flags & MP_DECODE_NO_BYPRODUCTS) != 0u && !mfma_hash;   // (what ran comes back: the form needs more LDS)
        MP_HIP_CHECK(launch_lsh_decode(lsh->bounds[layer_id], lsh->table[layer_id], q, s->Wk, s->wnorm, s->D,
                                       s->K, s->KLpad, lsh->codes, lsh->qnorm, lsh->results, lsh->nnz,
                                       attn->kv[layer_id], attn->kn[layer_id], attn->part_o, attn->part_ml,
                                       attn->part_cnt, attn->head_cnt, output, max_value_expsum, attn->head_mz,
                                       lsh->slots.empty() ? nullptr : lsh->slots[layer_id], lsh->slot_log2, attn->score, attn->err, attn_slices_per_head(attn->M), lsh->R,
                                       attn->xcd_rr && g_opt.decode_agent_scope.load() == 0,
                                       win ? win->kv[layer_id] : nullptr, win_len, win ? win->M : 0, BH, lsh->G,
                                       lsh->L, lsh->NB, lsh->M, mfma_hash, lsh->xw, lsh->xseq, lsh->xwords, xmode, lsh->idbits_of[layer_id], lsh->idbits_dev + layer_id,
                                       kn_payload ? lsh->pay_bad + goff : nullptr, lsh->att_ver_dev + goff,
                                       attn->kn_ver_dev + goff, lean, &lean, s->Wt, g_opt.decode_quad_hash.load(), st));
        attn->lastz = lsh->nnz;
        attn->score_state = lean ? 3 : 1;               // 3: the last call left no logits (mp_attn_get_score says so)
        attn->seg_cnt = lsh->R > 1 ? attn->part_cnt : nullptr;
        attn->seg_R = lsh->R;
        lsh->last_lean = lean;                          // ... and no codes (mp_lsh_get_mask says so)
    } else {
        // two launches: (hash + retrieve), then attention (models/attnserver.py:264-299, :300)
        MP_HIP_CHECK(launch_lsh_hash_retrieve(lsh->bounds[layer_id], lsh->table[layer_id], q, s->Wk,
                                              s->wnorm, s->D, s->K, s->KLpad, lsh->codes, lsh->qnorm,
                                              lsh->results, lsh->nnz, BH, lsh->G, lsh->L, lsh->NB, lsh->M,
                                              lsh->R, lsh->idbits_dev + layer_id, st));
        int rc = attn_run(attn, layer_id, false, s->K, s->L, output, max_value_expsum, q, MP_DTYPE_BF16,
                          lsh->qnorm, lsh->results, lsh->nnz, st);
        if (rc) return rc;
    }
    if (nnz_out)
        MP_HIP_CHECK(hipMemcpyAsync(nnz_out, lsh->nnz, (size_t)BH * 4, hipMemcpyDeviceToDevice, st));
    return MP_OK;
}

int mp_decode_sparse_layer(mp_simhash_t* s, mp_lsh_t* lsh, mp_attn_t* attn, int layer_id,
                           const uint16_t* q, uint16_t* output, float* max_value_expsum,
                           int32_t* nnz_out, mp_stream_t stream) {
    return decode_layer(s, lsh, attn, nullptr, layer_id, q, nullptr, output, max_value_expsum, nnz_out, 0u,
                        (hipStream_t)stream, "mp_decode_sparse_layer");
}

int mp_decode_sparse_layer_ex(mp_simhash_t* s, mp_lsh_t* lsh, mp_attn_t* attn, int layer_id,
                              const uint16_t* q, uint16_t* output, float* max_value_expsum,
                              int32_t* nnz_out, unsigned int flags, mp_stream_t stream) {
    return decode_layer(s, lsh, attn, nullptr, layer_id, q, nullptr, output, max_value_expsum, nnz_out, flags,
                        (hipStream_t)stream, "mp_decode_sparse_layer_ex");
}

int mp_decode_layer_window(mp_simhash_t* s, mp_lsh_t* lsh, mp_attn_t* attn, mp_attn_t* window,
                           int layer_id, const uint16_t* q, const int32_t* window_len, uint16_t* output,
                           float* max_value_expsum, int32_t* nnz_out, mp_stream_t stream) {
    MP_REQUIRE(window != nullptr, MP_ERR_INVALID, "mp_decode_layer_window: null window store");
    return decode_layer(s, lsh, attn, window, layer_id, q, window_len, output, max_value_expsum, nnz_out, 0u,
                        (hipStream_t)stream, "mp_decode_layer_window");
}

int mp_decode_layer_window_ex(mp_simhash_t* s, mp_lsh_t* lsh, mp_attn_t* attn, mp_attn_t* window,
                              int layer_id, const uint16_t* q, const int32_t* window_len, uint16_t* output,
                              float* max_value_expsum, int32_t* nnz_out, unsigned int flags, mp_stream_t stream) {
    MP_REQUIRE(window != nullptr, MP_ERR_INVALID, "mp_decode_layer_window_ex: null window store");
    return decode_layer(s, lsh, attn, window, layer_id, q, window_len, output, max_value_expsum, nnz_out, flags,
                        (hipStream_t)stream, "mp_decode_layer_window_ex");
}

// =================================================================== LSE merge

int mp_merge_state(const uint16_t* va, const float* sa, const uint16_t* vb, const float* sb, int R,
                   int D, uint16_t* v, float* s, mp_stream_t stream) {
    MP_REQUIRE(va && sa && vb && sb && v && R >= 1 && D >= 1, MP_ERR_INVALID, "mp_merge_state: bad argument");
    MP_HIP_CHECK(launch_merge_state(va, sa, vb, sb, R, D, v, s, (hipStream_t)stream));
    return MP_OK;
}

}  // extern "C"
