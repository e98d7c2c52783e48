// common.h -- shared device/host helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/magicpig_hip.h"

namespace mp {

constexpr int WAVE = 64;
#define MP_CLUSTER_MAX 32   // workgroups per query head of the one-launch decode, at most (one XCD's 32 CUs)

// ---------------------------------------------------------------- error plumbing (host)
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define MP_HIP_CHECK(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return ::mp::fail(_e == hipErrorOutOfMemory ? MP_ERR_NOMEM : MP_ERR_HIP,        \
                              std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

#define MP_REQUIRE(cond, code, msg)                      \
    do {                                                 \
        if (!(cond)) return ::mp::fail((code), (msg));   \
    } while (0)

// ---------------------------------------------------------------- bf16 (device)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
    return __uint_as_float(((uint32_t)h) << 16);
}
// low / high bf16 of a packed dword -> f32 (one VALU op each)
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// round-to-nearest-even, NaN preserved (torch semantics)
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// ---------------------------------------------------------------- bf16 dot products (v_dot2c_f32_bf16)
// acc += sum of 8 bf16 products: four chained v_dot2c_f32_bf16 (D += a.lo*b.lo + a.hi*b.hi).
// Inline asm on purpose: with ROCm 7.2's hipcc, __builtin_amdgcn_fdot2_f32_bf16 fed from ext-vector
// element extracts selects element 0 for EVERY call (scripts/probe_dot2.hip); the asm form is
// correct.  hipcc pads nothing inside asm, so the gfx940-class DOT hazards are handled by hand: a
// DOT result may feed the next same-opcode DOT as its accumulator with 0 wait states, but any other
// VALU read / write of it needs 3 / 4 wait states (LLVM GCNHazardRecognizer,
// DotWriteDifferentVALURead / Write) -> call dot_settle(acc) once after the last link of a chain.
__device__ __forceinline__ void dot8_bf16_chain(float& acc, const u32x4& a, const u32x4& b) {
    asm("v_dot2c_f32_bf16 %0, %1, %5\n\t"
        "v_dot2c_f32_bf16 %0, %2, %6\n\t"
        "v_dot2c_f32_bf16 %0, %3, %7\n\t"
        "v_dot2c_f32_bf16 %0, %4, %8"
        : "+v"(acc)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
__device__ __forceinline__ void dot_settle(float& acc) { asm("s_nop 3" : "+v"(acc)); }

// ---------------------------------------------------------------- row checksum (host-buffer mode)
// One word of the per-row checksum the stand-alone retrieve leaves for rows it writes straight into a caller's PINNED
// `results` (capi.hip: HostRetrieve): a non-linear 32-bit mix (lowbias32) of (entry, position).  Summed with wrap-around
// next to the position-weighted linear sum: an edit of the row that keeps both needs to be constructed for it (the
// linear pair alone was kept by e.g. +1, -2, +1 on three neighbours -- ADVICE r04).
__host__ __device__ inline uint32_t row_mix(uint32_t v, uint32_t pos) {
    uint32_t x = v ^ (pos * 0x9E3779B1u);
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// ---------------------------------------------------------------- debug phase timestamps
// Compiled in only with -DMP_STAMPS=1 (scripts/build_variant.py stamps -DMP_STAMPS=1; the measurement scripts load that
// build): the product's kernels carry no stamp code at all.  There: stamp == nullptr unless set through
// mp_debug_set_stamp_buffer (scripts/phase_times.py); one lane of workgroup 0 records the 100 MHz wall clock at phase
// boundaries.  A translation unit that defines MP_STAMP_STRIDE (an int lvalue in device memory) lets EVERY workgroup
// record: workgroup b writes at [b * stride + slot] when the stride is > 0 (scripts/phase_spread.py).
#ifndef MP_STAMPS
#define MP_STAMPS 0
#endif
#if !MP_STAMPS
#define MP_STAMP(stamp, slot) do { } while (0)
#elif defined(MP_STAMP_STRIDE)
#define MP_STAMP(stamp, slot)                                                               \
    do {                                                                                    \
        if ((stamp) != nullptr && threadIdx.x == 0) {                                       \
            const int _ss = MP_STAMP_STRIDE;                                                \
            if (_ss > 0) (stamp)[(size_t)blockIdx.x * _ss + (slot)] = wall_clock64();       \
            else if ((blockIdx.x | blockIdx.y | blockIdx.z) == 0) (stamp)[(slot)] = wall_clock64(); \
        }                                                                                   \
    } while (0)
#else
#define MP_STAMP(stamp, slot)                                                               \
    do {                                                                                    \
        if ((stamp) != nullptr && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) \
            (stamp)[(slot)] = wall_clock64();                                               \
    } while (0)
#endif

// ---------------------------------------------------------------- wave / block primitives
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// Wave-wide reductions: four DPP row rotations leave every lane of a 16-lane row with the row's
// result, four v_readlane + three scalar-operand ops join the rows.  The shuffle form is six dependent
// ds_bpermute round trips (~0.25 us on a critical path).
template <int N>
__device__ __forceinline__ float dpp_ror_f(float v) {
    const int x = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x120 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_ror_f<8>(v));
    v = fmaxf(v, dpp_ror_f<4>(v));
    v = fmaxf(v, dpp_ror_f<2>(v));
    v = fmaxf(v, dpp_ror_f<1>(v));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_ror_f<8>(v);
    v += dpp_ror_f<4>(v);
    v += dpp_ror_f<2>(v);
    v += dpp_ror_f<1>(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}
template <int N>
__device__ __forceinline__ double dpp_ror_d(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const int lo = (int)(uint32_t)u, hi = (int)(uint32_t)(u >> 32);
    const uint32_t rl = (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, 0x120 + N, 0xf, 0xf, true);
    const uint32_t rh = (uint32_t)__builtin_amdgcn_update_dpp(hi, hi, 0x120 + N, 0xf, 0xf, true);
    return __longlong_as_double((long long)((unsigned long long)rl | ((unsigned long long)rh << 32)));
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const unsigned long long u = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
    return __longlong_as_double((long long)((unsigned long long)lo | ((unsigned long long)hi << 32)));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_ror_d<8>(v);
    v += dpp_ror_d<4>(v);
    v += dpp_ror_d<2>(v);
    v += dpp_ror_d<1>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
// Reductions over the 16 lanes of a DPP row (lanes 16r .. 16r+15) by row rotation: four VALU
// instructions with a DPP operand instead of four ds_bpermute round trips; every lane of the
// row ends up with the result.
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {
    const int x = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x120 + N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ uint32_t dpp_row_ror(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x120 + N, 0xf, 0xf, true);
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<1>(v);
    return v;
}
// max of NON-NEGATIVE floats (ordered like their bit patterns; integer max needs no NaN canonicalisation)
__device__ __forceinline__ float row16_max_nonneg(float f) {
    uint32_t v = (uint32_t)__float_as_int(f);
    v = max(v, dpp_row_ror<8>(v));
    v = max(v, dpp_row_ror<4>(v));
    v = max(v, dpp_row_ror<2>(v));
    v = max(v, dpp_row_ror<1>(v));
    return __int_as_float((int)v);
}
// Inclusive prefix sum over the 64 lanes with DPP operands (six VALU adds; the shuffle form costs
// six ds_bpermute round trips): Hillis-Steele inside each row of 16 by row_shr 1/2/4/8 (lanes
// shifted in from outside the row read the identity), then row_bcast:15 carries a row's total into
// rows 1 and 3 and row_bcast:31 carries lanes 0..31's total into rows 2 and 3.
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// wave-wide sum of u32 values with wrap-around (checksums): the same DPP ladder on unsigned arithmetic
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// A workgroup barrier that orders LDS traffic only: global loads and stores stay in flight across it (__syncthreads() also
// waits for vmcnt(0): every outstanding global access of the wave).  Use where the barrier protects LDS data and nothing
// read back from global memory (HIP guide, "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Exclusive prefix sum over the block's threads; `total` gets the block sum.
// s_tmp: LDS scratch of >= blockDim.x/64 + 1 ints.  Contains two workgroup barriers (LDS_ONLY: lds_barrier()).
template <bool LDS_ONLY = false>
__device__ __forceinline__ int block_excl_scan(int v, int* s_tmp, int& total) {
    const int l = lane_id(), w = threadIdx.x >> 6, nw = (blockDim.x + WAVE - 1) >> 6;
    const int inc = wave_incl_scan(v);
    if (l == WAVE - 1) s_tmp[w] = inc;
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    if (w == 0) {
        int x = (l < nw) ? s_tmp[l] : 0;
        int xi = wave_incl_scan(x);
        if (l < nw) s_tmp[l] = xi - x;
        if (l == nw - 1) s_tmp[nw] = xi;
    }
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    total = s_tmp[nw];
    return s_tmp[w] + inc - v;
}

}  // namespace mp
