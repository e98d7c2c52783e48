// capi_support.h -- what every entry point leans on: error text, the debug / A-B options, the device guard, staging buffers, the host-buffer mode's completion word (HostFlag) and pinned-memory map (HostMap)
// (one of the pieces capi.hip is made of: included there, once, in this order; not a header for other translation units)
#pragma once

namespace mp {

// ---- error text (thread local)
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// ---- debug / A-B options (mp_debug_set_option): process-wide, read at call time
struct DebugOptions {
    std::atomic<int> decode_two_launch{0};   // 1: hash+retrieve launch, then attention launch
    std::atomic<int> decode_cluster{0};      // 0 = auto, else workgroups per head (clamped to [1, min(32, slices)])
    std::atomic<int> decode_agent_scope{0};  // 1: cluster hand-off through memory even when the XCD placement was observed
    std::atomic<int> decode_mfma_hash{0};    // 1: query SimHash by the MFMA kernel in a launch of its own, then the decode
    std::atomic<int> decode_split_hash{-1};  // -1 = auto, 0 = never, 1 = always (clusters on one XCD), 2 = split but nobody publishes (test)
    std::atomic<int> decode_quad_hash{-1};   // one workgroup per head: -1 = auto, 0 = never, 1 = the heads of an XCD residue hash in quads on the matrix pipe
    std::atomic<int> decode_direct{-1};      // -1 = auto, 0 = never, 1 = always (when R > 1) keep direct piece slots
    std::atomic<int> attn_head_kernel{-1};   // -1 = auto, 0 = split-KV kernel, 1 = one workgroup per head
    std::atomic<int> attn_gx{0};             // 0 = auto, else split-KV workgroups per head
    std::atomic<int> attn_dense_grouped{1};  // full_attention: 1 = K/V read once per kv group, 0 = once per query head
    std::atomic<int> decode_kn_payload{1};   // 1: use the key norms attached to the table entries (where attached), 0: one HBM access per token
    std::atomic<int> host_zero_copy{1};      // MP_MEM_HOST calls: 1 = kernels work on pinned memory in place (the caller's, or the handle's mirror), 0 = staged copies
    std::atomic<int> host_flag_wait{0};      // MP_MEM_HOST calls: 1 = wait for the stream by spinning on a word a one-thread kernel
                                             // writes to pinned memory instead of hipStreamSynchronize (A/B, EXPERIMENTS.md R4-5)
    // counters (read with mp_debug_get_option, reset with mp_debug_set_option(name, 0)): how the MP_MEM_HOST attention
    // entry served its calls -- a fast path that silently stops hitting shows here (fallbacks must be observable)
    std::atomic<int> build_rank_exact{0};    // 1: the table build ranks by match-any ballots always (A/B, tests); 0: by the LDS's lane order, verified
    std::atomic<int> build_rank_fallbacks{0};// counter: builds redone with the exact ranking because a bucket run did not ascend
    std::atomic<int> build_rank_inject{0};   // test hook: n > 0 = the next n table builds behave as if the fast ranking's check had failed
    std::atomic<int> host_fast_hits{0};      // the rows batch_retrieve had just handed out were recognised: no index upload
    std::atomic<int> host_fast_edited{0};    // pairing found, but a row differed from what was handed out: launch dropped, upload path
    std::atomic<int> host_fast_unpaired{0};  // no pairing (other buffers, other counts, another handle in between): upload path
    std::atomic<int> host_speculate{1};      // MP_MEM_HOST batch_retrieve enqueues the paired store's attention launch behind its own
                                             // kernel when the last attention call came with a pinned query tensor (see mp_lsh::Spec)
    std::atomic<int> host_spec_hits{0};      // counter: attention calls served by the launch the retrieve had issued
    std::atomic<int> host_spec_misses{0};    // counter: such a launch existed but the call's arguments were not what it had assumed
    std::atomic<int> host_flag_timeouts{0};  // counter: a completion word did not arrive within ~5 ms (the stream was synchronised instead)
    // where a MP_MEM_HOST batch_retrieve spends its time, ns summed over the calls since the last reset (scripts/host_mode_times.py):
    // up to the last launch, waiting for its completion word, copying counts and rows out + bookkeeping
    std::atomic<int> host_ret_calls{0}, host_ret_ns_enqueue{0}, host_ret_ns_wait{0}, host_ret_ns_copy{0};
    std::atomic<int> host_copy_prefetch{48}; // the copy of the handed-out rows asks for the NEXT row while it copies one: the whole row up to
                                             // this many 64-byte lines, the first 8 lines of a longer one; 0 = off (A/B: R6-2)
};
static DebugOptions g_opt;

static std::atomic<int>* debug_option(const char* name) {
    if (!name) return nullptr;
    if (!strcmp(name, "decode_two_launch")) return &g_opt.decode_two_launch;
    if (!strcmp(name, "decode_cluster")) return &g_opt.decode_cluster;
    if (!strcmp(name, "decode_agent_scope")) return &g_opt.decode_agent_scope;
    if (!strcmp(name, "decode_direct")) return &g_opt.decode_direct;
    if (!strcmp(name, "decode_split_hash")) return &g_opt.decode_split_hash;
    if (!strcmp(name, "decode_quad_hash")) return &g_opt.decode_quad_hash;
    if (!strcmp(name, "decode_mfma_hash")) return &g_opt.decode_mfma_hash;
    if (!strcmp(name, "attn_head_kernel")) return &g_opt.attn_head_kernel;
    if (!strcmp(name, "attn_gx")) return &g_opt.attn_gx;
    if (!strcmp(name, "attn_dense_grouped")) return &g_opt.attn_dense_grouped;
    if (!strcmp(name, "decode_kn_payload")) return &g_opt.decode_kn_payload;
    if (!strcmp(name, "host_zero_copy")) return &g_opt.host_zero_copy;
    if (!strcmp(name, "host_flag_wait")) return &g_opt.host_flag_wait;
    if (!strcmp(name, "build_rank_exact")) return &g_opt.build_rank_exact;
    if (!strcmp(name, "build_rank_fallbacks")) return &g_opt.build_rank_fallbacks;
    if (!strcmp(name, "build_rank_inject")) return &g_opt.build_rank_inject;
    if (!strcmp(name, "host_fast_hits")) return &g_opt.host_fast_hits;
    if (!strcmp(name, "host_fast_edited")) return &g_opt.host_fast_edited;
    if (!strcmp(name, "host_fast_unpaired")) return &g_opt.host_fast_unpaired;
    if (!strcmp(name, "host_speculate")) return &g_opt.host_speculate;
    if (!strcmp(name, "host_spec_hits")) return &g_opt.host_spec_hits;
    if (!strcmp(name, "host_spec_misses")) return &g_opt.host_spec_misses;
    if (!strcmp(name, "host_flag_timeouts")) return &g_opt.host_flag_timeouts;
    if (!strcmp(name, "host_copy_prefetch")) return &g_opt.host_copy_prefetch;
    if (!strcmp(name, "host_ret_calls")) return &g_opt.host_ret_calls;
    if (!strcmp(name, "host_ret_ns_enqueue")) return &g_opt.host_ret_ns_enqueue;
    if (!strcmp(name, "host_ret_ns_wait")) return &g_opt.host_ret_ns_wait;
    if (!strcmp(name, "host_ret_ns_copy")) return &g_opt.host_ret_ns_copy;
    return nullptr;
}

// ---- every handle remembers the device that was current when its state was allocated; entry points
// switch to it for the duration of the call (allocations, launches and memsets then target the owning
// device whatever the caller's current device is) and restore the caller's device on return
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define MP_ON_DEVICE(h) ::mp::DeviceGuard _device_guard((h) ? (h)->device : -1)

static int current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}

// ---- small RAII device buffer for staging host arguments
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T>
    T* as() { return reinterpret_cast<T*>(p); }
};

// Stage `bytes` of a caller buffer into HBM when it lives on the host; device buffers pass through.
static int stage_in(const void* src, size_t bytes, int mem, DevBuf& buf, const void** out) {
    if (mem == MP_MEM_DEVICE) {
        *out = src;
        return MP_OK;
    }
    MP_HIP_CHECK(buf.alloc(bytes));
    MP_HIP_CHECK(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice));
    *out = buf.p;
    return MP_OK;
}

// Persistent staging of the host-buffer mode: a pinned host block and a device block of the same size, grown on
// demand and kept by the handle (no hipMalloc / hipFree per call).
struct Stage {
    void* hp = nullptr;   // pinned host
    void* hd = nullptr;   // the same block as the device sees it (kernels read / write it over PCIe)
    void* dp = nullptr;   // device
    size_t cap = 0;
    // host_only: the block is a pinned MIRROR the kernels write over PCIe (hp / hd); no device twin is allocated
    int reserve(size_t bytes, bool host_only = false) {
        if (bytes <= cap && (host_only || dp != nullptr)) return MP_OK;
        size_t want = cap ? cap : 4096;
        while (want < bytes) want *= 2;
        release();
        MP_HIP_CHECK(hipHostMalloc(&hp, want, hipHostMallocMapped));
        if (!host_only) MP_HIP_CHECK(hipMalloc(&dp, want));
        if (hipHostGetDevicePointer(&hd, hp, 0) != hipSuccess) {
            (void)hipGetLastError();
            hd = nullptr;                      // no alias: callers fall back to copies
        }
        cap = want;
        return MP_OK;
    }
    void release() {
        if (hp) (void)hipHostFree(hp);
        if (dp) (void)hipFree(dp);
        hp = dp = hd = nullptr;
        cap = 0;
    }
};

// Host-buffer mode: waiting for the stream.  hipStreamSynchronize costs ~10 us from the kernel's end to the caller's next
// instruction; the `host_flag_wait` option instead has a one-thread kernel write a sequence number to a pinned word behind
// the call's launches and spins on it (PCIe posted writes of one device arrive in order: what the launches wrote to pinned
// memory is there when the word is).  Falls back to the synchronisation after 5 ms.
struct HostFlag {
    unsigned int* hp = nullptr;   // pinned word
    unsigned int* hd = nullptr;   // as the device sees it
    unsigned int seq = 0;
    // arm: a one-thread kernel on `st` writes the next sequence number to the pinned word (0 = could not be armed);
    // reached: spin until the word has got there (false after ~5 ms: the caller synchronises the stream instead).  A call may
    // arm a word in the MIDDLE of what it enqueues and wait for that point only (capi.hip: the speculative attention launch).
    unsigned int arm(hipStream_t st) {
        if (hp == nullptr) {
            void* p = nullptr;
            if (hipHostMalloc(&p, 64, hipHostMallocMapped) == hipSuccess) {
                hp = reinterpret_cast<unsigned int*>(p);
                *hp = 0u;
                void* d = nullptr;
                if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess) hd = reinterpret_cast<unsigned int*>(d);
            }
            if (hd == nullptr) (void)hipGetLastError();
        }
        if (hd == nullptr) return 0u;
        if (++seq == 0u) ++seq;
        if (launch_host_flag(hd, seq, st) != hipSuccess) {
            (void)hipGetLastError();
            return 0u;
        }
        return seq;
    }
    bool reached(unsigned int want) const {
        if (want == 0u || hp == nullptr) return false;
        volatile unsigned int* f = hp;
        for (long it = 0; it < 5000000L; ++it) {          // ~5 ms
            if ((int)(*f - want) >= 0) return true;       // (a later number has passed it)
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        return false;
    }
    int wait(hipStream_t st, bool spin) {
        if (spin && reached(arm(st))) return MP_OK;
        MP_HIP_CHECK(hipStreamSynchronize(st));
        return MP_OK;
    }
    void release() {
        if (hp) (void)hipHostFree(hp);
        hp = hd = nullptr;
    }
};

// Host-buffer mode without copies: the address at which a KERNEL can read / write a caller's host buffer in place.
// The reference's callers hold pinned tensors (models/attnserver.py:59-66: hipHostMalloc through torch's pin_memory):
// those are mapped already.  A PAGEABLE buffer (results_lsh_cpu and nnz, :59-60) is never touched by a kernel: the kernels
// work on a pinned mirror owned by the handle and the host copies the live entries across.  (Rounds 2-3 could also
// REGISTER a pageable buffer -- hipHostRegister, the opt-in `host_register` mode; the full GPU suite aborted twice inside
// the ROCm runtime with it, and round 4's hunt -- scripts/experiments/stress_host_register.py: registrations that outlive,
// or are outlived by, their buffers, heap-resident and really unmapped ones, 150 iterations each, under rocgdb -- reproduced
// neither the aborts nor a wrong result.  A mode whose failure cannot be explained does not ship: removed, EXPERIMENTS.md
// R4-6.)  nullptr = not mapped.
struct HostMap {
    const void* last_pageable = nullptr;   // the last pointer found to be plain pageable memory (negative results only
                                           // are remembered: treating pinned memory as pageable is merely slower)
    void* resolve(const void* ptr, size_t /*bytes*/) {
        if (ptr == last_pageable) return nullptr;   // (the failing lookup below costs microseconds per call)
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, ptr) == hipSuccess) {
            if (a.type == hipMemoryTypeHost && a.devicePointer != nullptr) return a.devicePointer;
            if (a.type != hipMemoryTypeUnregistered) return nullptr;   // device / managed memory passed as "host"
        } else {
            (void)hipGetLastError();                               // pageable memory: "invalid value" on older runtimes
        }
        last_pageable = ptr;
        return nullptr;
    }
    void release() { last_pageable = nullptr; }
};

constexpr int FILL_BLOCKS = 1024;   // row blocks of mp_attn_fill_offload's column sums
constexpr int MAX_CLUSTER = MP_CLUSTER_MAX;   // workgroups per query head of the one-launch decode, at most

static int alloc_zero(void** p, size_t bytes) {
    MP_HIP_CHECK(hipMalloc(p, bytes ? bytes : 1));
    MP_HIP_CHECK(hipMemset(*p, 0, bytes));
    return MP_OK;
}


}  // namespace mp
