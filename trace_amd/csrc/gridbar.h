// Grid barrier for the persistent decode kernels (decode_b1.hip, decode_wide.hip; round 6).  MI355X_MICROARCH.md "barrier-xcd", placement-independent form:
// workgroup w belongs to group w % 8 (= its XCD when the dispatcher deals workgroups round-robin; correctness does not depend on that), arrives with a
// fire-and-forget atomic on its group's counter; workgroups 0..7 collect their group, meet on a top counter and publish their group's generation word, which the
// other members poll (one lane, relaxed agent-scope loads, s_sleep).  Counters are monotonic across launches, compared modulo 2^32 (below); every spin
// is bounded — a timeout raises an error word and the launch drains without hanging the device.  What crosses a barrier must have been stored WRITE-THROUGH
// (st_wt16 / agent-scope atomic stores) and drained by the storing wave (bar_arrive waits vmcnt(0) first): there is no release fence (buffer_wbl2 measured
// 1 us per barrier); the consumer side is one acquire fence by the polling lane followed by a workgroup barrier.
#pragma once
#include "common.h"

namespace gridbar {
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ void st_wt16(void* base, size_t byte_off, u32x4_t v) {         // base is wave-uniform
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, 16);                  // aux 16 = sc1
}
__device__ __forceinline__ void st_wt8(void* base, size_t byte_off, unsigned int lo, unsigned int hi) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{lo, hi}, rs, (int)byte_off, 0, 16);
}
constexpr int BAR_STRIDE = 32;                 // words between two barrier counters (128 bytes: one L2 line each)
constexpr int BAR_CNT = 0, BAR_TOP = 8, BAR_GEN = 9, BAR_ERR = 17, BAR_BASE = 18, BAR_WORDS = 19 * BAR_STRIDE;
constexpr unsigned SPIN_LIMIT = 4u << 20;       // polls (each followed by s_sleep): a few seconds; then the launch gives up

struct GridBar {
    unsigned* w;
    unsigned* err;         // sticky error word (not reset per launch)
    unsigned epoch;        // barriers passed so far (same in every workgroup)
    int nwg;
    bool dead;             // a spin timed out somewhere: skip all further work
    int opt;               // A/B bits (DecodeB1Args::prefetch >> 2): 1 = wave 0 polls before it requests its own weights, 2 = + a release fence by the arriving lane, 4 = no acquire fence (TIMING ONLY), 8 = every wave fences, 16 = no waiting at all (TIMING ONLY)
};

__device__ __forceinline__ unsigned bar_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// end of a phase: every wave's stores have left (vmcnt), the workgroup meets, one lane makes them visible to the other XCDs and checks in
__device__ __forceinline__ void bar_arrive(GridBar& gb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (gb.opt & 8) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // A/B: every wave releases
    __syncthreads();
    if (threadIdx.x == 0) {
        // (every cross-workgroup payload store is write-through and has drained: no release fence; opt bit 2 adds one — A/B)
        if (gb.opt & 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_fetch_add(gb.w + (BAR_CNT + (blockIdx.x & 7)) * BAR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gb.epoch += 1;
}
// wait until every workgroup has checked in for the current epoch; returns with the workgroup synchronised and its caches acquired.
// Two levels, no top counter: collector g (workgroup g < 8) waits for its group's arrivals, then stores the epoch into slot g of EVERY group's generation line
// (8 stores); a workgroup of group h polls line h — one 32-byte load, eight slots — until all eight slots have reached the epoch.  After the last arrival:
// one atomic + one poll round (collector) + one store + one poll round (member) + the acquire.
__device__ __forceinline__ void bar_wait(GridBar& gb, int* s_flag) {
    if (threadIdx.x == 0) {
        const int g = blockIdx.x & 7;
        const unsigned members = (unsigned)((gb.nwg - g + 7) >> 3);
        const int groups = gb.nwg < 8 ? gb.nwg : 8;
        bool ok = true;
        unsigned spins = 0;
        if (gb.opt & 16) ok = true;          // TIMING ONLY: no waiting at all (what a free barrier would give)
        else {
            if ((int)blockIdx.x < 8) {
                while ((int)(bar_load(gb.w + (BAR_CNT + g) * BAR_STRIDE) - members * gb.epoch) < 0) {
                    if (++spins > SPIN_LIMIT || bar_load(gb.w + BAR_ERR * BAR_STRIDE)) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (ok) {
                    for (int h = 0; h < groups; ++h)
                        __hip_atomic_store(gb.w + (BAR_GEN + h) * BAR_STRIDE + g, gb.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const unsigned* line = gb.w + (BAR_GEN + g) * BAR_STRIDE;
            while (ok) {
                bool all = true;
                for (int h = 0; h < groups; ++h) all &= (int)(bar_load(line + h) - gb.epoch) >= 0;
                if (all) break;
                if (++spins > SPIN_LIMIT || bar_load(gb.w + BAR_ERR * BAR_STRIDE)) { ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (!ok) {
            __hip_atomic_store(gb.w + BAR_ERR * BAR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // sticky record (the host reads word 0 in trace_decode_read): first failing workgroup leaves [1, epoch, workgroup, polls]
            if (__hip_atomic_exchange(gb.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __hip_atomic_store(gb.err + 1, gb.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gb.err + 2, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gb.err + 3, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!(gb.opt & 4)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_flag = ok ? 0 : 1;
    }
    __syncthreads();
    if (gb.opt & 8) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // A/B: every wave acquires
    if (*s_flag) gb.dead = true;
    __syncthreads();
}

// The counters are NEVER reset between launches (a memset node in front of a replayed graph's kernel was seen to run while the PREVIOUS replay's kernel was
// still counting: barriers passed early or never, profiles/r06_decode_b1_persistent_ab.txt): they run on, modulo 2^32, and every launch continues at the epoch
// the previous one left in the BASE word.  bar_begin: every workgroup reads BASE before its first barrier (nobody writes it during the launch); bar_end:
// workgroup 0, after the launch's LAST bar_wait (every workgroup has long read BASE by then), stores the epoch the next launch starts from.  Every launch
// that shares a barrier block must use the same grid size.  The host zeroes the block only while nothing runs (context creation, trace_decode_begin).
__device__ __forceinline__ void bar_begin(GridBar& gb, int* s_flag) {
    if (threadIdx.x == 0) *s_flag = (int)bar_load(gb.w + BAR_BASE * BAR_STRIDE);
    __syncthreads();
    gb.epoch = (unsigned)*s_flag;
    __syncthreads();
}
__device__ __forceinline__ void bar_end(GridBar& gb, unsigned final_epoch) {          // final_epoch: gb.epoch after the launch's last bar_arrive, the same in every workgroup
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(gb.w + BAR_BASE * BAR_STRIDE, final_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
inline size_t bar_bytes() { return (size_t)BAR_WORDS * 4; }
}  // namespace gridbar
