// capi_launchers.h -- the kernels' host launchers (simhash.hip, lsh.hip, attention.hip) as the C ABI's entry points call them
// (one of the pieces capi.hip is made of: included there, once, in this order; not a header for other translation units)
#pragma once

namespace mp {


// ---- kernels' host launchers (simhash.hip, lsh.hip, attention.hip)
int simhash_padded_cols(int K, int L);
int simhash_supported(int D, int K);
hipError_t launch_simhash_prepare(const uint16_t*, int, int, int, uint16_t*, uint16_t*, float*, hipStream_t);
hipError_t launch_simhash_query(const uint16_t*, const uint16_t*, const float*, int, int, int, int,
                                int32_t*, float*, float*, hipStream_t);
hipError_t launch_simhash_keys(const uint16_t*, const uint16_t*, const float*, int, int64_t, int,
                               int, int, int16_t*, hipStream_t);
size_t retrieve_lds_bytes(int64_t M, int L);
size_t lsh_lds_limit();
int lsh_range_len(int64_t M, int R);
bool lsh_decode_supported(int64_t M, int L, int D, int R);
bool xcd_round_robin_verified();
hipError_t launch_lsh_decode(const int32_t*, const int32_t*, const uint16_t*, const uint16_t*, const float*, int,
                             int, int, int32_t*, float*, int32_t*, int32_t*, const uint16_t*, const float*,
                             float*, float2*, int*, int*, uint16_t*, float*, float2*, const int32_t*, int, float*, int*,
                             int, int, bool, const uint16_t*, const int32_t*, int64_t, int, int, int, int, int64_t,
                             bool, unsigned long long*, unsigned int*, int, int, int, const int*, const int*, const unsigned int*,
                             const unsigned int*, bool, bool*, const uint16_t*, int, hipStream_t);
hipError_t set_stamp_stride(int);
void set_exact_norm(int);
void set_slot_log2(int);
hipError_t launch_lsh_slots(const int32_t*, const int32_t*, int32_t*, int, int, int, int64_t, int, hipStream_t);
int get_slot_log2();
int lsh_slot_log2(int64_t M, int NB, int R);
hipError_t launch_lsh_fill(const int16_t*, const int32_t*, int, int64_t, int, int64_t, int, int32_t*,
                           int32_t*, int*, hipStream_t);
hipError_t launch_lsh_unsort(const int16_t*, const int32_t*, int, int64_t, int16_t*, int*, hipStream_t);
hipError_t launch_lsh_subbounds(const int32_t*, int32_t*, int, int, int, int64_t, int, hipStream_t);
hipError_t launch_lsh_build(const int16_t*, int, int64_t, int, int64_t, int, int32_t*, int32_t*, int*, const float*, int,
                            int, int*, bool*, bool*, bool, hipStream_t);
hipError_t launch_lsh_retrieve(const int32_t*, const int32_t*, const int32_t*, int32_t*, int32_t*, int,
                               int, int, int, int64_t, int, const int*, int32_t*, int32_t*, uint32_t*, hipStream_t);
hipError_t launch_lsh_attach_norms(int32_t*, const float*, int, int, int64_t, int, int*, hipStream_t);
hipError_t launch_lsh_hash_retrieve(const int32_t*, const int32_t*, const uint16_t*, const uint16_t*,
                                    const float*, int, int, int, int32_t*, float*, int32_t*, int32_t*,
                                    int, int, int, int, int64_t, int, const int*, hipStream_t);
hipError_t launch_lsh_compact(uint32_t*, const int*, int, int, int64_t, hipStream_t);
hipError_t launch_lsh_hash_only(const uint16_t*, const uint16_t*, const float*, int, int, int, int32_t*, float*, int, int,
                                hipStream_t);
hipError_t launch_lsh_mask(const int32_t*, const int32_t*, const int32_t*, int8_t*, int, int, int, int,
                           int64_t, int, int, hipStream_t);
int attn_slices_per_head(int64_t M);
int attn_supported_head_dim(int D);
hipError_t launch_attn_sparse(int, bool, bool, const uint16_t*, const float*, const void*, const float*,
                              const int32_t*, const int32_t*, float*, float2*, int*, uint16_t*, float*,
                              float2*, float*, int, int, int64_t, int, int, int, bool, hipStream_t);
bool launch_attn_dense(int, int, bool, const uint16_t*, const void*, const int32_t*, float*, float2*, int*, uint16_t*,
                       float*, float2*, float*, int, int64_t, int, hipStream_t, hipError_t*);
hipError_t launch_attn_normalize(float*, const int32_t*, const float2*, int, int64_t, hipStream_t);
hipError_t launch_attn_fill(const uint16_t*, const uint16_t*, const float*, int, int64_t, int,
                            int64_t, uint16_t*, float*, hipStream_t);
hipError_t launch_key_centre_fill(const uint16_t*, const uint16_t*, int64_t, int64_t, int, int, int64_t, double*, int,
                                  uint16_t*, uint16_t*, float*, hipStream_t);
hipError_t launch_simhash_keys_strided(const uint16_t*, int64_t, int64_t, const uint16_t*, const float*, int, int64_t,
                                       int, int, int, int16_t*, const float*, int64_t, hipStream_t);
hipError_t launch_ragged_offsets(const int32_t*, int, int64_t, int32_t*, hipStream_t);
hipError_t launch_ragged_copy(bool, int32_t*, int32_t*, const int32_t*, int, int64_t, hipStream_t);
hipError_t launch_merge_state(const uint16_t*, const float*, const uint16_t*, const float*, int,
                              int, uint16_t*, float*, hipStream_t);
hipError_t launch_attn_append(const uint16_t*, const uint16_t*, const int32_t*, int, const uint16_t*, int, int, int, int64_t,
                              uint16_t*, float*, unsigned int*, int*, hipStream_t);
bool lsh_hash_only_supported(int L);
hipError_t launch_attn_ticket_check(int*, int, int*, hipStream_t);
hipError_t launch_relay(const void*, void*, size_t, hipStream_t);
hipError_t launch_row_norm(const void*, bool, int, int, float*, float*, void*, hipStream_t);
hipError_t launch_host_flag(unsigned int*, unsigned int, hipStream_t);
hipError_t launch_host_rows(const int32_t*, const int32_t*, int32_t*, int64_t, int, const void*, void*, size_t, int,
                            hipStream_t);

extern unsigned long long* g_stamp;


}  // namespace mp
