/* A plain-C client of the C ABI (no HIP headers, no torch): what a maintainer binding
 * libmagicpig_hip.so from cgo / JNI / a C++ pybind11 module would write.  Host buffers only
 * (MP_MEM_HOST): the library stages them through HBM, as the reference's callers hand it pinned
 * CPU tensors (models/attnserver.py:59-66).
 *
 *   client <in.bin> <out.bin>
 * in.bin : int32 header {B, H, Hkv, D, K, L, n, M} then
 *          hash_func  bf16  [D][K*L]
 *          keys       bf16  [B][Hkv][n][D]      (centred)
 *          values     bf16  [B][Hkv][n][D]
 *          key_norm   f32   [B][Hkv][n]
 *          query      bf16  [B*H][D]
 * out.bin: codes int32 [B*H][L], nnz int32 [B*H], results int32 [B*H][M],
 *          output bf16 [B*H][D], max_value_expsum f32 [2][B*H]
 * Steps: key SimHash -> device table build -> KV fill -> query SimHash -> batch_retrieve ->
 * attention_wrapper, i.e. models/attnserver.py:159-193 and :264-300 through the three-call API. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "magicpig_hip.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != MP_OK) {                                                            \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mp_last_error());      \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

static void* xread(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "short read (%zu bytes)\n", bytes);
        exit(3);
    }
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
        return 1;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hd[8];
    if (fread(hd, 4, 8, f) != 8) return 3;
    const int B = hd[0], H = hd[1], Hkv = hd[2], D = hd[3], K = hd[4], L = hd[5];
    const int64_t n = hd[6], M = hd[7];
    const int BH = B * H;
    uint16_t* hash_func = xread(f, (size_t)D * K * L * 2);
    uint16_t* keys = xread(f, (size_t)B * Hkv * n * D * 2);
    uint16_t* vals = xread(f, (size_t)B * Hkv * n * D * 2);
    float* kn = xread(f, (size_t)B * Hkv * n * 4);
    uint16_t* q = xread(f, (size_t)BH * D * 2);
    fclose(f);

    mp_simhash_t* sh;
    mp_lsh_t* lsh;
    mp_attn_t* attn;
    CHECK(mp_simhash_create(&sh));
    CHECK(mp_simhash_set_planes(sh, D, K, L, hash_func, MP_MEM_HOST, NULL));
    CHECK(mp_lsh_create(&lsh));
    CHECK(mp_lsh_alloc(lsh, K, L, 1, H, Hkv, B, (int)M));
    CHECK(mp_attn_create(&attn));
    CHECK(mp_attn_alloc(attn, 1, H, Hkv, D, B, (int)M));

    int16_t* kcodes = malloc((size_t)Hkv * L * n * 2);
    for (int b = 0; b < B; ++b) {
        const size_t off = (size_t)b * Hkv * n * D;
        CHECK(mp_simhash_keys(sh, keys + off, Hkv, n, kcodes, MP_MEM_HOST, NULL));
        CHECK(mp_lsh_build(lsh, 0, b, kcodes, n, MP_MEM_HOST, NULL));
        CHECK(mp_attn_fill(attn, 0, b, keys + off, vals + off, kn + (size_t)b * Hkv * n, n, MP_MEM_HOST, NULL));
    }

    int32_t* codes = calloc((size_t)BH * L, 4);
    float* qn = calloc((size_t)BH, 4);
    int32_t* results = calloc((size_t)BH * M, 4);
    int32_t* nnz = calloc((size_t)BH, 4);
    uint16_t* out = calloc((size_t)BH * D, 2);
    float* mve = calloc((size_t)2 * BH, 4);
    CHECK(mp_simhash_query(sh, q, BH, codes, qn, MP_MEM_HOST, NULL));
    CHECK(mp_lsh_batch_retrieve(lsh, 0, codes, results, nnz, MP_MEM_HOST, NULL));
    CHECK(mp_attn_sparse(attn, 0, K, L, out, mve, q, MP_DTYPE_BF16, qn, results, nnz, MP_MEM_HOST, NULL));

    FILE* g = fopen(argv[2], "wb");
    if (!g) return 1;
    fwrite(codes, 4, (size_t)BH * L, g);
    fwrite(nnz, 4, (size_t)BH, g);
    fwrite(results, 4, (size_t)BH * M, g);
    fwrite(out, 2, (size_t)BH * D, g);
    fwrite(mve, 4, (size_t)2 * BH, g);
    fclose(g);
    CHECK(mp_attn_destroy(attn));
    CHECK(mp_lsh_destroy(lsh));
    CHECK(mp_simhash_destroy(sh));
    printf("ok arch=%s version=%d\n", mp_arch(), mp_version());
    return 0;
}
