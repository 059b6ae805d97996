// common.h — shared device/host declarations for libmillion_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/million_hip.h"

namespace million {

typedef _Float16 f16;

constexpr int kMaxSplits = 64;      // code splits per (b, hk); workspace is sized for this
constexpr int kMaxG = 8;            // q heads per kv head handled by one workgroup
constexpr int kCntBytes = 1024;     // counter block at the start of the workspace (multiple of 16)

// Kernel parameter block (passed by value).
struct AttnParams {
    const f16 *q;
    const uint8_t *k_codes;
    const uint8_t *v_codes;
    const void *k_page_ids;
    const void *v_page_ids;
    const f16 *k_tab;      // prepared K table, row image [m][c][dm]
    const f16 *k_tab_col;  // prepared K table, col image [c][m][dm]
    const f16 *v_tab;      // prepared V table, row image
    const f16 *v_tab_col;  // prepared V table, col image [c][m][dm]
    const f16 *k_res;
    const f16 *v_res;
    f16 *out;
    float *ws_part;
    int *ws_cnt;
    const int *dev_lengths;
    int bs, nh, nh_k, G, d, M, C, dm;
    int T, r, rstart, rcap;
    long long res_sb, res_sh, k_sb, k_sh, v_sb, v_sh;
    int k_paged, v_paged, page_size, n_pages_cap, ids64;
    int nsplit;      // code splits per (b, hk)
    int nslots;      // partial slots per (b, hk)
    int split_len;   // tokens per split
    int slot_floats; // floats per partial slot = G*d + 2*G (padded to 4)
    float scale_log2e;
};

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

__device__ __forceinline__ void load_lengths(const AttnParams &p, int b, int &T, int &r, int &rstart) {
    if (p.dev_lengths) {
        T = p.dev_lengths[b * 4 + 0];
        r = p.dev_lengths[b * 4 + 1];
        rstart = p.dev_lengths[b * 4 + 2];
        if (T > p.T) T = p.T;   // host value is the bound the grid was sized for
    } else {
        T = p.T; r = p.r; rstart = p.rstart;
    }
}

__device__ __forceinline__ long long page_id_at(const AttnParams &p, const void *ids, int bh, int page) {
    const long long idx = (long long)bh * p.n_pages_cap + page;
    return p.ids64 ? ((const long long *)ids)[idx] : (long long)((const int *)ids)[idx];
}

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
// domain), then l[G].  flag_lds: one int of LDS scratch.  All threads of the workgroup call this.
__device__ __forceinline__ void publish_and_merge(const AttnParams &p, int b, int hk, int slot,
                                                  const float *part_lds, int *flag_lds) {
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int bh = b * p.nh_k + hk;
    const int G = p.G, d = p.d;
    const int nvals = G * d + 2 * G;
    float *dst = p.ws_part + ((long long)bh * p.nslots + slot) * p.slot_floats;
    for (int i = tid; i < nvals; i += nthr) st_agent(dst + i, part_lds[i]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(p.ws_cnt + bh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag_lds = (t == p.nslots - 1);
    }
    __syncthreads();
    if (!*flag_lds) return;

    // ---- last arriver: merge all slots of (b, hk) ----
    const float *src = p.ws_part + (long long)bh * p.nslots * p.slot_floats;
    for (int i = tid; i < G * d; i += nthr) {
        const int g = i / d;
        float mx = -INFINITY;
        for (int s = 0; s < p.nslots; ++s) mx = fmaxf(mx, ld_agent(src + (long long)s * p.slot_floats + G * d + g));
        float num = 0.f, den = 0.f;
        if (mx > -INFINITY) {
            for (int s = 0; s < p.nslots; ++s) {
                const float *sp = src + (long long)s * p.slot_floats;
                const float ms = ld_agent(sp + G * d + g);
                if (ms > -INFINITY) {
                    const float w = exp2f(ms - mx);
                    num += w * ld_agent(sp + i);
                    den += w * ld_agent(sp + G * d + G + g);
                }
            }
        }
        const float o = den > 0.f ? num / den : 0.f;
        p.out[((long long)b * p.nh + hk * G + g) * d + (i - g * d)] = (f16)o;
    }
    if (tid == 0) __hip_atomic_store(p.ws_cnt + bh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct EncParams {
    const f16 *x;
    const f16 *cents;
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
int launch_attn_mfma(const AttnParams &p, hipStream_t s);
int launch_encode(const EncParams &p, hipStream_t s);
bool attn_mfma_supported(const AttnParams &p);
void set_error(const char *fmt, ...);

}  // namespace million
