// common.h — shared device/host declarations for libmillion_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/million_hip.h"

namespace million {

typedef _Float16 f16;

constexpr int kMaxSplits = 64;      // code splits per (b, hk); workspace is sized for this
constexpr int kMaxG = 8;            // q heads per kv head handled by one workgroup of the tile and scalar kernels
constexpr int kMaxGMfma = 16;       // ... of the MFMA kernels (attn_mfma.hip): the 16 columns of the score tile
constexpr int kCntBytes = 1024;     // granule of the counter / flag blocks at the start of the workspace
// Workspace head: one 128-byte record per (b, kv head), then one int per batch item (second-level ticket), then
// 2 x kFlagWords words per (b, kv head) (split flags, XCD census line), then the split partials.  Record words (u32):
//   [0..1]  u64 arrival ticket of the round-2 hand-off (ticket_and_merge below: tile and scalar kernels)
//   [2]     ticket word     } L2 hand-off of the MFMA kernels (attn_mfma.hip, "Tail"): arrival count of the (b, kv head)'s
//   [3]     generation      } workgroups in bits 31:8 (never reset: a launch counts from [4]), give-up bits of the merge helpers
//   [4]     count base      } in bits 7:0; flags of launch n carry generation + 1
// Every word is at rest when a launch ends (ticket [0..1] zero, give-up bits zero, generation and base advanced).
constexpr int kRecWords = 32;
constexpr int kFlagWords = 64;      // one flag per split (the MFMA kernels use at most kMaxSplits = 64 slots)

// Kernel parameter block (passed by value).
struct AttnParams {
    const f16 *q;
    const uint8_t *k_codes;
    const uint8_t *v_codes;
    // page ids, typed + restrict (read-only in every kernel): uniform indices become scalar loads
    const int *__restrict__ k_ids32;
    const int *__restrict__ v_ids32;
    const long long *__restrict__ k_ids64;
    const long long *__restrict__ v_ids64;
    const f16 *k_tab;      // prepared K table, row image [m][c][dm]
    const f16 *k_tab_col;  // prepared K table, col image [c][m][dm]
    const f16 *v_tab;      // prepared V table, row image
    const f16 *v_tab_col;  // prepared V table, col image [c][m][dm]
    const f16 *k_res;
    const f16 *v_res;
    const f16 *k_new;      // fused append: the new token's K/V rows (bs, nh_k, 1, d), or null
    const f16 *v_new;
    f16 *k_res_w;          // writable aliases of k_res / v_res for the fused append
    f16 *v_res_w;
    int *dev_lengths_w;    // writable alias of dev_lengths (r += 1 after a fused append)
    int *ws_cnt2;          // per-batch second-level ticket
    f16 *out;
    float *ws_part;
    unsigned long long *ws_cnt;   // records of kRecWords u32 per (b, kv head), see above
    unsigned *ws_flags;           // 2 * kFlagWords per (b, kv head): split flags, then the XCD census line
    const int *__restrict__ dev_lengths;
    int bs, nh, nh_k, G, d, M, C, dm;   // G: query heads of a kv head served by THIS launch (<= kMaxG)
    int Gt, g0;                         // nh / nh_k, and the first of them this launch serves (query-head groups > kMaxG: several launches)
    // Query-head parts (lean kernel at d = 64, streaming kernel at d = 128 / M = 16; 5 .. 16 query heads per kv head): the launch treats each
    // kv head as hparts_m1 + 1 = ceil(heads / 4) VIRTUAL kv heads of G = ceil(heads / parts) query heads each (the last part: what is left) (virtual kv head hk = part * real nh_k + real hk: the parts of a real head
    // share an XCD), every tail structure indexed by the virtual pair.  No split: nhk_real = 2^28 (no hk reaches it: part = 0 without a
    // branch or a division), nhk_mul = 0, hparts_m1 = 0
    int nhk_real, nhk_mul, hparts_m1;
    int G_all;                          // query heads per REAL kv head of this launch when it runs as parts (the last part may hold fewer than G)
    int T, r, rstart, rcap;
    long long res_sb, res_sh, k_sb, k_sh, v_sb, v_sh;
    int k_paged, v_paged, page_size, ps_shift, n_pages_cap, ids64;
    int v_identity;  // V pages are a dense scratch pool: page id = bh * n_pages_cap + page (no id table)
    int nsplit;      // code splits per (b, hk)
    int nslots;      // partial slots per (b, hk)
    int tail_test;   // diagnostics (million_set_force_generic(4 / 8)): the helpers of the merge give up at once (1: preset bits, 2: real path)
    int nmerge;      // MFMA kernels: workgroups per (b, kv head) that merge heads (the last arriver + nmerge - 1 helpers); 1 when the grid does not fit the chip
    int split_len;   // tokens per split
    int slot_floats; // floats per partial slot = G*d + 2*G, padded to whole 128-byte lines (no line is shared by two slots)
    float scale_log2e;
    unsigned long long *dbg;   // diagnostic stamp buffer (million_debug_set_stamp_buffer), normally null
    int k_pool_pages, v_pool_pages;   // pages in the pools (0 = not given): read only under MILLION_DEBUG_CHECK_IDS
    int *bad_ids;                     // MILLION_DEBUG_CHECK_IDS: device counter of out-of-range page ids, else null
};

// Page ids are trusted in the product build, as in the reference.  A library built with -DMILLION_DEBUG_CHECK_IDS routes
// every id the three attention kernels read through checked_page_id: an id outside the pool becomes page 0 (a valid
// address) and is counted.  `live`: the id is really used (the streaming kernel preloads table entries beyond the
// context, which may hold anything); dead ids are clamped without being counted.
#ifdef MILLION_DEBUG_CHECK_IDS
template <class T>
__device__ __forceinline__ T checked_page_id(const AttnParams &p, T id, int pool, bool live = true) {
    if (pool > 0 && (unsigned long long)(long long)id >= (unsigned long long)pool) {
        if (live && p.bad_ids) atomicAdd(p.bad_ids, 1);
        return (T)0;
    }
    return id;
}
#define MILLION_CHECK_KID(p, id, ...) ::million::checked_page_id((p), (id), (p).k_pool_pages, ##__VA_ARGS__)
#define MILLION_CHECK_VID(p, id, ...) ::million::checked_page_id((p), (id), (p).v_pool_pages, ##__VA_ARGS__)
#else
#define MILLION_CHECK_KID(p, id, ...) (id)
#define MILLION_CHECK_VID(p, id, ...) (id)
#endif

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

typedef unsigned v2u __attribute__((ext_vector_type(2)));

// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with the gfx950 row swaps:
// pure VALU.  (ds_bpermute-based __shfl_xor goes through the LDS queue, which the code-byte gathers of the
// other waves keep saturated: a dependent shuffle then costs a microsecond.)
// NOTE (hipcc, ROCm 7.2): `__builtin_bit_cast(float, v[1])` on an element of an ext_vector (here the pair a
// permlane swap builtin returns) is mis-lowered to element 0 — seen in the ISA: the row maximum became
// "row 0" and the row sum 4 x row 0.  The elements are therefore copied to scalars and converted with
// __uint_as_float.  tests: test_rows_reduce_selfcheck.
__device__ __forceinline__ unsigned opaque_copy(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ v2u swap16_self(unsigned x) {      // rows {0,0,2,2} of x / rows {1,1,3,3} of x
    return __builtin_amdgcn_permlane16_swap(x, opaque_copy(x), false, false);
}
__device__ __forceinline__ v2u swap32_self(unsigned x) {      // lower half twice / upper half twice
    return __builtin_amdgcn_permlane32_swap(x, opaque_copy(x), false, false);
}
__device__ __forceinline__ float rows_max(float x) {
    const v2u a = swap16_self(__float_as_uint(x));
    const unsigned a0 = a[0], a1 = a[1];
    const float m1 = fmaxf(__uint_as_float(a0), __uint_as_float(a1));
    const v2u b = swap32_self(__float_as_uint(m1));
    const unsigned b0 = b[0], b1 = b[1];
    return fmaxf(__uint_as_float(b0), __uint_as_float(b1));
}
__device__ __forceinline__ float rows_sum(float x) {
    const v2u a = swap16_self(__float_as_uint(x));
    const unsigned a0 = a[0], a1 = a[1];
    const float s1 = __uint_as_float(a0) + __uint_as_float(a1);
    const v2u b = swap32_self(__float_as_uint(s1));
    const unsigned b0 = b[0], b1 = b[1];
    return __uint_as_float(b0) + __uint_as_float(b1);
}
// first query head (row of q / out) of kv head hk in this launch
__device__ __forceinline__ int head_part(const AttnParams &p, int hk) {      // which part of its real kv head a (virtual) kv head is: three compares
    return (hk >= p.nhk_real) + (hk >= 2 * p.nhk_real) + (hk >= 3 * p.nhk_real);
}
__device__ __forceinline__ int head0(const AttnParams &p, int hk) {
    const int part = head_part(p, hk);
    return (hk - part * p.nhk_mul) * p.Gt + p.g0 + part * p.G;
}

// Device-resident lengths are not trusted: T is clamped to the host bound the grid was sized for, r to the window
// capacity (minus the row a fused append is about to add), the ring start to [0, cap).  Out-of-range values become a
// shorter context / window, never an out-of-bounds read.
__device__ __forceinline__ void clamp_lengths(const AttnParams &p, int &T, int &r, int &rstart) {
    T = T < 0 ? 0 : (T > p.T ? p.T : T);
    const int rmax = p.rcap - (p.k_new ? 1 : 0);
    r = r < 0 ? 0 : (r > rmax ? rmax : r);
    rstart = (unsigned)rstart < (unsigned)p.rcap ? rstart : 0;
}
__device__ __forceinline__ void load_lengths(const AttnParams &p, int b, int &T, int &r, int &rstart) {
    if (p.dev_lengths) {
        T = p.dev_lengths[b * 4 + 0];
        r = p.dev_lengths[b * 4 + 1];
        rstart = p.dev_lengths[b * 4 + 2];
        clamp_lengths(p, T, r, rstart);
    } else {
        T = p.T; r = p.r; rstart = p.rstart;
    }
}

__device__ __forceinline__ long long k_page_id(const AttnParams &p, int bh, int page) {
    const long long idx = (long long)bh * p.n_pages_cap + page;
    return MILLION_CHECK_KID(p, p.ids64 ? p.k_ids64[idx] : (long long)p.k_ids32[idx]);
}
__device__ __forceinline__ long long v_page_id(const AttnParams &p, int bh, int page) {
    const long long idx = (long long)bh * p.n_pages_cap + page;
    if (p.v_identity) return idx;
    return MILLION_CHECK_VID(p, p.ids64 ? p.v_ids64[idx] : (long long)p.v_ids32[idx]);
}

// Diagnostic stamps (off unless a buffer is set): lane 0 of each of the first 8 waves of a workgroup stores the
// 100 MHz realtime counter at phase boundaries; 32 slots per wave, layout [workgroup][wave][slot].  Never
// read by any kernel.
constexpr int kStampWaves = 8, kStampSlots = 32;
#define MILLION_STAMP(p, i)                                                                              \
    do {                                                                                                 \
        if ((p).dbg && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < kStampWaves)                      \
            (p).dbg[(((long long)blockIdx.y * gridDim.x + blockIdx.x) * kStampWaves + (threadIdx.x >> 6)) * kStampSlots + (i)] = \
                __builtin_amdgcn_s_memrealtime();                                                        \
    } while (0)

// ---- inter-workgroup hand-off of split partials (cdna_hip_programming.md, Guideline 16, R1 counter form) ----
// Producer: agent-scope relaxed (sc1, write-through) stores of the partial, every storing wave drains
// vmcnt, workgroup barrier, ONE lane adds to the (b,hk) ticket.  The workgroup whose add returns
// nslots-1 merges: its other waves join a workgroup barrier after the add has returned, then EVERY load
// of handed-off bytes is an agent-scope relaxed (sc1) load.  One workgroup per CU in the fast kernel.
__device__ __forceinline__ void st_agent(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// part_lds: this workgroup's partial in LDS: O[G][d] (unnormalised, relative to m), then m[G] (log2
// domain), then l[G].  scratch: >= (2*nslots*G + 2*G) floats of LDS nobody else uses any more.
// flag_lds: one int of LDS scratch.  All threads of the workgroup call this.
// Last-arriver combine with 16-byte sc1 loads: the scalar form below issues one 4-byte agent-scope load per
// (element, slot) - 256 wave-level requests of 256 B for 32 slots x 512 elements, ~1.9 us of load-path time on
// the critical path of the whole launch; here a thread owns 4 consecutive output elements of nthr / (G*d/4)
// interleaved slot subsets: 64 requests of 1 KiB, then one LDS reduction over the subsets.
// Returns false (nothing done) for shapes it does not cover.
typedef unsigned mv4u __attribute__((ext_vector_type(4)));
typedef float mv4f __attribute__((ext_vector_type(4)));
// AUX = cache-policy bits of the loads of handed-off bytes: 16 (sc1, agent scope).  (Round 2 switched to sc0 loads when
// the ticket said every producer had run on the merger's XCD; an sc0 load may be served by the CU's L1, which nothing
// refreshes - tools/micro/l2_handoff.hip shows polls with sc0 loads never seeing a later store - so that path is gone.)
template <int G_, int AUX, int kThreads = 512, int kD = 128>
__device__ __forceinline__ void merge_vec4(const AttnParams &p, int b, int hk, const float *src, int ns, float *scratch) {
    constexpr int nq = G_ * kD / 4;            // float4 groups of the output
    constexpr int nsg = kThreads / nq;         // slot subsets = threads per group
    constexpr int kPer = (32 + nsg - 1) / nsg; // loads per thread (ns <= 32)
    const int tid = threadIdx.x;
    const int q = tid % nq, sg = tid / nq;
    const int g = (4 * q) / kD;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
    mv4u v[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int slot = sg + k * nsg;
        const int sl = slot < ns ? slot : ns - 1;                      // clamped: never a conditional load
        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (sl * p.slot_floats + 4 * q) * 4, 0, AUX);
    }
    // softmax weights of the slots (lane = slot), as in the scalar form
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int gg = wv; gg < G_; gg += kThreads / 64) {
            const bool on = lane < ns;
            const int sl = on ? lane : 0;                                  // clamped: never a conditional load
            const float m1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (sl * p.slot_floats + G_ * kD + gg) * 4, 0, AUX));
            const float l1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (sl * p.slot_floats + G_ * kD + G_ + gg) * 4, 0, AUX));
            const float m0 = on ? m1 : -INFINITY;
            const float l0 = on ? l1 : 0.f;
            const float mx = wave_max(m0);
            const float ms_ = mx > -INFINITY ? mx : 0.f;
            const float w0 = exp2f(m0 - ms_);
            const float den = wave_sum(w0 * l0);
            const float inv = den > 0.f ? 1.0f / den : 0.f;
            if (on) scratch[lane * G_ + gg] = w0 * inv;
        }
    }
    __syncthreads();
    mv4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int slot = sg + k * nsg;
        const float w = slot < ns ? scratch[slot * G_ + g] : 0.f;
        acc += w * __builtin_bit_cast(mv4f, v[k]);
    }
    mv4f *red = (mv4f *)(scratch + 512);       // behind the (<= 32 * 8) weights
    red[sg * nq + q] = acc;
    __syncthreads();
    if (sg == 0) {
#pragma unroll
        for (int j = 1; j < nsg; ++j) acc += red[j * nq + q];
        typedef f16 h4 __attribute__((ext_vector_type(4)));
        const h4 o = {(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
        *(h4 *)(p.out + ((long long)b * p.nh + head0(p, hk)) * kD + 4 * q) = o;
    }
}

// ticket_and_merge: the caller has ISSUED the sc1 stores of its partial into slot_ptr(p, b, hk, slot) (every
// storing thread of the workgroup, straight from registers if it likes); this drains them, takes the ticket and,
// in the last-arriving workgroup, merges.  publish_and_merge: the same with the partial staged in LDS.
__device__ __forceinline__ float *slot_ptr(const AttnParams &p, int b, int hk, int slot) {
    return p.ws_part + ((long long)(b * p.nh_k + hk) * p.nslots + slot) * p.slot_floats;
}
__device__ __forceinline__ void ticket_and_merge(const AttnParams &p, int b, int hk, float *scratch, int *flag_lds);

__device__ __forceinline__ void publish_and_merge(const AttnParams &p, int b, int hk, int slot,
                                                  const float *part_lds, float *scratch, int *flag_lds) {
    const int nvals = p.G * p.d + 2 * p.G;
    float *dst = slot_ptr(p, b, hk, slot);
    for (int i = threadIdx.x; i < nvals; i += blockDim.x) st_agent(dst + i, part_lds[i]);
    ticket_and_merge(p, b, hk, scratch, flag_lds);
}

__device__ __forceinline__ void ticket_and_merge(const AttnParams &p, int b, int hk, float *scratch, int *flag_lds) {
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int bh = b * p.nh_k + hk;
    const int G = p.G, d = p.d;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned long long now = __hip_atomic_fetch_add(p.ws_cnt + (long long)bh * (kRecWords / 2), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        *flag_lds = now == (unsigned long long)p.nslots ? 1 : 0;
    }
    __syncthreads();
    MILLION_STAMP(p, 10);
    const int arrival = *flag_lds;      // 1 = this workgroup arrived last
    if (!arrival) return;

    // ---- last arriver: merge all slots of (b, hk); every load of handed-off bytes is an sc1 load ----
    // One memory round trip: each thread first requests the values of its output element from up to
    // kMergeBatch slots; meanwhile wave w turns the (m, l) pairs of head g = w, w + nwaves, ... into
    // softmax weights (lane = slot, wave max / sum) and leaves them in LDS; one barrier; every thread
    // then combines its values with the weights of its head.
    constexpr int kMergeBatch = 32;
    const int ns = p.nslots;
    const float *src = p.ws_part + (long long)bh * ns * p.slot_floats;
    if ((nthr == 512 || nthr == 1024) && (d == 128 || d == 64) && ns <= 32 && (G == 1 || G == 2 || G == 4 || G == 8)) {      // workgroup-uniform
        MILLION_STAMP(p, 11);
#define MILLION_MV4(G_) \
        { if (nthr == 512) { if (d == 128) merge_vec4<G_, 16, 512, 128>(p, b, hk, src, ns, scratch); else merge_vec4<G_, 16, 512, 64>(p, b, hk, src, ns, scratch); } \
          else { if (d == 128) merge_vec4<G_, 16, 1024, 128>(p, b, hk, src, ns, scratch); else merge_vec4<G_, 16, 1024, 64>(p, b, hk, src, ns, scratch); } }
        if (G == 4) MILLION_MV4(4)
        else if (G == 8) MILLION_MV4(8)
        else if (G == 2) MILLION_MV4(2)
        else MILLION_MV4(1)
#undef MILLION_MV4
        goto merged;
    }
    {
    const bool fast = (G * d <= nthr) && (ns <= kMergeBatch);      // workgroup-uniform: straight-line path
    float v[kMergeBatch];
    if (fast) {
        const int e0 = tid < G * d ? tid : 0;
#pragma unroll
        for (int k = 0; k < kMergeBatch; ++k) v[k] = ld_agent(src + (long long)(k < ns ? k : ns - 1) * p.slot_floats + e0);
    }
    {
        const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
        for (int g = wv; g < G; g += nwv) {
            const bool on = lane < ns;      // ns <= kMaxSplits + 1 = 65 > 64: slot 64 handled by lane 63 below
            float m0 = on ? ld_agent(src + (long long)lane * p.slot_floats + G * d + g) : -INFINITY;
            float l0 = on ? ld_agent(src + (long long)lane * p.slot_floats + G * d + G + g) : 0.f;
            float m1 = -INFINITY, l1 = 0.f;
            if (ns > 64 && lane == 63) {
                m1 = ld_agent(src + 64ll * p.slot_floats + G * d + g);
                l1 = ld_agent(src + 64ll * p.slot_floats + G * d + G + g);
            }
            const float mx = wave_max(fmaxf(m0, m1));
            const float ms_ = mx > -INFINITY ? mx : 0.f;
            const float w0 = exp2f(m0 - ms_), w1 = exp2f(m1 - ms_);       // m = -inf -> 0
            const float den = wave_sum(w0 * l0 + w1 * l1);
            const float inv = den > 0.f ? 1.0f / den : 0.f;
            if (on) scratch[lane * G + g] = w0 * inv;
            if (ns > 64 && lane == 63) scratch[64 * G + g] = w1 * inv;
        }
    }
    __syncthreads();
    MILLION_STAMP(p, 11);
    if (fast) {
        if (tid < G * d) {
            const int g = tid / d;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < kMergeBatch; ++k) {
                const float w = k < ns ? scratch[k * G + g] : 0.f;
                acc = fmaf(w, v[k], acc);
            }
            p.out[((long long)b * p.nh + head0(p, hk) + g) * d + (tid - g * d)] = (f16)acc;
        }
    } else {
        for (int e = tid; e < G * d; e += nthr) {
            const int g = e / d;
            float acc = 0.f;
            for (int s0 = 0; s0 < ns; s0 += kMergeBatch) {
#pragma unroll
                for (int k = 0; k < kMergeBatch; ++k)
                    v[k] = ld_agent(src + (long long)(s0 + k < ns ? s0 + k : ns - 1) * p.slot_floats + e);
#pragma unroll
                for (int k = 0; k < kMergeBatch; ++k)
                    if (s0 + k < ns) acc = fmaf(scratch[(s0 + k) * G + g], v[k], acc);
            }
            p.out[((long long)b * p.nh + head0(p, hk) + g) * d + (e - g * d)] = (f16)acc;
        }
    }
    }
merged:
    if (tid == 0) {
        __hip_atomic_store(p.ws_cnt + (long long)bh * (kRecWords / 2), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // fused append with device-resident lengths: every workgroup of batch b has read its lengths once
        // all nh_k heads have been merged; the last merger advances r
        if (p.k_new && p.dev_lengths_w) {
            const int t2 = __hip_atomic_fetch_add(p.ws_cnt2 + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t2 == p.nh_k - 1) {
                __hip_atomic_store(p.ws_cnt2 + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p.dev_lengths_w[b * 4 + 1] += 1;
            }
        }
    }
}

struct EncParams {
    const f16 *x;
    const f16 *cents;
    const float *cents32;   // fp32 row image [m][c][dm] of a prepared codebook, or null
    uint8_t *dst;
    const int *page_ids;
    int bs, nh_k, n, d, M, C, dm;
    long long xsb, xsh, xsn;
    int xrow_start, xrow_mod;
    int layout, tok0;
    long long dsb, dsh;
    int page_size, n_pages_cap;
    const int *dev_lengths;
};

// host-side launchers (defined in the .hip files)
int launch_attn_generic(const AttnParams &p, hipStream_t s);
constexpr int kAttnNotHandled = 1;      // launch_attn_mfma: shape fine in principle, this call is for the generic kernel
int launch_attn_mfma(const AttnParams &p, hipStream_t s);
int launch_attn_tile(const AttnParams &p, hipStream_t s);
bool attn_tile_supported(const AttnParams &p);   // shape taken by the tile kernel and V already in transposed pages
bool attn_tile_shape_ok(const AttnParams &p);    // shape taken by the tile kernel once V is transposed
int launch_encode(const EncParams &p, hipStream_t s);
int launch_decode(const void *codes, const f16 *cents, f16 *out, long long n_rows, int M, int C, int dm, hipStream_t s);
struct FlushLayers { int n_layers; long long x_ls, ids_ls, len_ls; int advance; };      // one layer: {1, 0, 0, 0, 1}
int launch_flush(const EncParams &k, const EncParams &v, int *dev_lengths_w, int rcap, int min_r, const FlushLayers &ly, hipStream_t s);
bool attn_mfma_shape_ok(const AttnParams &p);
int launch_rows_reduce_check(const float *in, float *out_max, float *out_sum, hipStream_t s);
bool attn_mfma_supported(const AttnParams &p);
bool attn_mfma_handles(const AttnParams &p);     // supported AND launch_attn_mfma will not hand the call back (kAttnNotHandled)
bool attn_mfma_streams(const AttnParams &p);     // ... and the streaming kernel runs (not the grouped one: T = 0, > 1M tokens)
void set_error(const char *fmt, ...);

// Per-device facts and one-time per-device setup (one process may drive several devices): CU count, and a set of
// "dynamic-LDS attribute already raised on this device" bits, one per kernel family.  Guarded by a mutex.
int device_cus();
bool device_once(int family);      // true exactly once per (current device, family 0..7)

int read_tail_faults();             // attn_mfma.hip: merges that ran out of their poll bound since the last call (and clears)
void set_prefill_policy(int plain);  // prefill.hip: million_set_force_generic(64) = the plain form of the prompt-attention kernel at d = 128 (A/B, tests)
void set_mfma_policy(int policy);   // attn_mfma.hip: A/B and test knob behind million_set_force_generic(2 / 4 / 8): bit 0 = grouped kernel only, bits 2:1 = merge-helper test mode

}  // namespace million
