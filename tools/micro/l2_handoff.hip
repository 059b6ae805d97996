// Microbenchmark + correctness probe: the split-partial hand-off of the fused attention kernel, through the XCD's L2
// instead of through memory.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/micro/l2_handoff tools/micro/l2_handoff.hip && build/micro/l2_handoff
//
// 256 workgroups (one per CU: 130 KiB of LDS each), groups of NS = 32 with equal id % 8 (observed: one XCD per group; the
// XCC ids are recorded and checked).  Every workgroup, after a pseudo-random delay:
//   takes the group's ticket EARLY (agent-scope atomic, before its data is out: it only elects the merger),
//   wave 0 stores a 2 KiB payload (values depend on launch number, group, split) with store flavour ST,
//   waits vmcnt(0), stores its flag (= launch number) with the same flavour.
// The workgroup whose ticket was last polls the 32 flags (lane = slot) with load flavour LD until all carry the launch
// number (bounded), then loads all 64 KiB of payload with LD and checks every word.  Stamps (100 MHz realtime counter):
// flag issued (per producer), polls done, payload checked (merger).
//
// Flavours (cache-policy bits of the buffer intrinsics): 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1.
// Question 1: which (ST, LD) pairs are CORRECT when all producers share the merger's XCD (no stale word, no timeout)?
// Question 2: how long from the last producer's flag to "payload checked", per pair?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int NS = 32, NG = 8, SLOT_WORDS = 512 + 32;      // 2 KiB payload + one 128-byte line of padding per slot
constexpr int FLAG_STRIDE = 64;                            // flags of a group: 64 words = two 128-byte lines

__device__ __forceinline__ unsigned payload_word(unsigned epoch, int g, int s, int i) {
    return (epoch * 2654435761u) ^ (unsigned)(g * 7919 + s * 104729 + i * 31);
}

// ELECT: 0 = the last ticket merges (first poll nearly always succeeds), 1 = the FIRST ticket merges (long polling: do
// repeated polls see later stores?).  MIXED: 1 = groups of consecutive ids (every group spans all 8 XCDs).
// NMERGE: the last NMERGE tickets each check a 1/NMERGE share of every slot (the per-head mergers of the attention tail).
template <int ST, int LD, int ELECT = 0, int MIXED = 0, int NMERGE = 1>
__global__ __launch_bounds__(512) void handoff(unsigned *part, unsigned *flags, unsigned *tick, unsigned long long *stamps,
                                               unsigned *report, unsigned epoch, int spin_bound) {
    extern __shared__ char smem[];
    int *flag_lds = (int *)smem;
    const int id = blockIdx.x, g = MIXED ? id / NS : id % NG, s = MIXED ? id % NS : id / NG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pseudo-random skew, 0 .. ~1.5 us
    unsigned h = (unsigned)id * 2246822519u + epoch * 3266489917u;
    h ^= h >> 15;
    for (unsigned i = 0; i < (h & 31); ++i) __builtin_amdgcn_s_sleep(8);
    if (tid == 0) stamps[id * 4 + 3] = __builtin_amdgcn_s_getreg(6164) & 15;      // XCC id
    unsigned my_ticket = 0;
    if (tid == 0) my_ticket = __hip_atomic_fetch_add(tick + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned *slot = part + (size_t)(g * NS + s) * SLOT_WORDS;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)slot, 0, 0x7fffffff, 0x00020000);
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i0 = (k * 64 + lane) * 4;
            v4u v = {payload_word(epoch, g, s, i0), payload_word(epoch, g, s, i0 + 1), payload_word(epoch, g, s, i0 + 2),
                     payload_word(epoch, g, s, i0 + 3)};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, i0 * 4, 0, ST);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            stamps[id * 4 + 0] = __builtin_amdgcn_s_memrealtime();
            __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)(flags + g * FLAG_STRIDE), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(epoch, rf, s * 4, 0, ST);
            *flag_lds = ELECT ? (my_ticket == 0 ? 1 : 0) : ((int)my_ticket >= NS - NMERGE ? (int)my_ticket - (NS - NMERGE) + 1 : 0);
        }
    }
    __syncthreads();
    if (!*flag_lds) return;
    // ---- merger ----
    const int share = *flag_lds - 1;
    if (share < 0 || share >= NMERGE) return;
    __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)(flags + g * FLAG_STRIDE), 0, 0x7fffffff, 0x00020000);
    int spins = 0;
    bool ok = false;
    while (spins < spin_bound) {
        const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(rf, (lane < NS ? lane : 0) * 4, 0, LD);
        ok = __all(f == epoch);
        if (ok) break;
        ++spins;
        __builtin_amdgcn_s_sleep(2);
    }
    if (tid == 0) stamps[id * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(part + (size_t)g * NS * SLOT_WORDS), 0, 0x7fffffff, 0x00020000);
    unsigned bad = 0;
    constexpr int KL = 8 / NMERGE, QS = 128 / NMERGE;      // float4 per slot in a share
    v4u v[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) {      // 32 slots x (128 / NMERGE) float4 over 512 threads
        const int q = k * 512 + tid, sl = q / QS, i0 = (share * QS + q % QS) * 4;
        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rp, (sl * SLOT_WORDS + i0) * 4, 0, LD);
    }
#pragma unroll
    for (int k = 0; k < KL; ++k) {
        const int q = k * 512 + tid, sl = q / QS, i0 = (share * QS + q % QS) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) bad += v[k][j] != payload_word(epoch, g, sl, i0 + j);
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if (lane == 0 && bad) atomicAdd(report + g * 4 + 0, bad);
    __syncthreads();
    if (tid == 0) {
        stamps[id * 4 + 2] = __builtin_amdgcn_s_memrealtime();
        if (!ok) atomicAdd(report + g * 4 + 1, 1u);      // timeout
        if (ELECT || my_ticket == NS - 1) {
            report[g * 4 + 2] = (unsigned)id;            // whose stamps the host reads
            report[g * 4 + 3] = (unsigned)spins;
        }
        if (my_ticket == NS - 1 && !ELECT) __hip_atomic_store(tick + g * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int ST, int LD, int ELECT = 0, int MIXED = 0, int NMERGE = 1>
void run(const char *name, unsigned *part, unsigned *flags, unsigned *tick, unsigned long long *stamps, unsigned *report,
         int launches, unsigned &epoch) {
    auto k = handoff<ST, LD, ELECT, MIXED, NMERGE>;
    const int lds = 130 * 1024;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    std::vector<unsigned long long> hs(NS * NG * 4);
    std::vector<unsigned> hr(NG * 4);
    std::vector<double> lat_poll, lat_done;
    long long stale = 0, timeouts = 0, mixed = 0, spins = 0;
    CK(hipMemset(tick, 0, NG * 32 * 4));      // a run that elects the FIRST ticket leaves the counters at NS
    for (int it = 0; it < launches; ++it) {
        ++epoch;
        CK(hipMemset(report, 0, NG * 4 * sizeof(unsigned)));
        CK(hipMemset(stamps, 0, NS * NG * 4 * sizeof(unsigned long long)));
        if (ELECT) CK(hipMemset(tick, 0, NG * 32 * 4));
        hipLaunchKernelGGL(k, dim3(NS * NG), dim3(512), lds, 0, part, flags, tick, stamps, report, epoch, 20000);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), report, hr.size() * 4, hipMemcpyDeviceToHost));
        for (int g = 0; g < NG; ++g) {
            stale += hr[g * 4 + 0];
            timeouts += hr[g * 4 + 1];
            spins += hr[g * 4 + 3];
            const int mid = (int)hr[g * 4 + 2];
            unsigned long long last_flag = 0;
            bool mix = false;
            for (int s = 0; s < NS; ++s) {
                const int id = MIXED ? g * NS + s : s * NG + g;
                last_flag = std::max(last_flag, hs[id * 4 + 0]);
                mix |= hs[id * 4 + 3] != hs[(MIXED ? g * NS : g) * 4 + 3];
            }
            mixed += mix;
            if (it >= 4) {
                lat_poll.push_back(((double)hs[mid * 4 + 1] - (double)last_flag) * 0.01);
                lat_done.push_back(((double)hs[mid * 4 + 2] - (double)last_flag) * 0.01);
            }
        }
    }
    std::sort(lat_poll.begin(), lat_poll.end());
    std::sort(lat_done.begin(), lat_done.end());
    auto q = [](std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
    printf("%-34s stale words %lld  timeouts %lld  groups on >1 XCD %lld  polls/merge %.1f | last flag -> polls ok: med %.2f p90 %.2f us | -> payload checked: med %.2f p90 %.2f us\n",
           name, stale, timeouts, mixed, (double)spins / (launches * NG), q(lat_poll, 0.5), q(lat_poll, 0.9), q(lat_done, 0.5), q(lat_done, 0.9));
    fflush(stdout);
}

int main() {
    unsigned *part, *flags, *tick, *report;
    unsigned long long *stamps;
    CK(hipMalloc(&part, (size_t)NG * NS * SLOT_WORDS * 4));
    CK(hipMalloc(&flags, NG * FLAG_STRIDE * 4));
    CK(hipMalloc(&tick, NG * 32 * 4));
    CK(hipMalloc(&report, NG * 4 * 4));
    CK(hipMalloc(&stamps, NS * NG * 4 * 8));
    CK(hipMemset(part, 0, (size_t)NG * NS * SLOT_WORDS * 4));
    CK(hipMemset(flags, 0, NG * FLAG_STRIDE * 4));
    CK(hipMemset(tick, 0, NG * 32 * 4));
    unsigned epoch = 0;
    const int L = 200;
    run<16, 16>("store sc1 / load sc1 (round 2)", part, flags, tick, stamps, report, L, epoch);
    run<0, 16>("store plain / load sc1", part, flags, tick, stamps, report, L, epoch);
    run<0, 2>("store plain / load nt", part, flags, tick, stamps, report, L, epoch);
    run<0, 1>("store plain / load sc0", part, flags, tick, stamps, report, L, epoch);
    run<0, 3>("store plain / load sc0 nt", part, flags, tick, stamps, report, L, epoch);
    run<0, 17>("store plain / load sc0 sc1", part, flags, tick, stamps, report, L, epoch);
    run<2, 2>("store nt / load nt", part, flags, tick, stamps, report, L, epoch);
    run<1, 16>("store sc0 / load sc1", part, flags, tick, stamps, report, L, epoch);
    run<16, 16>("store sc1 / load sc1 (again)", part, flags, tick, stamps, report, L, epoch);
    printf("-- four mergers (the last four tickets), a quarter of every slot each --\n");
    run<16, 16, 0, 0, 4>("sc1 / sc1, 4 mergers", part, flags, tick, stamps, report, L, epoch);
    run<0, 16, 0, 0, 4>("plain / sc1, 4 mergers", part, flags, tick, stamps, report, L, epoch);
    printf("-- the FIRST ticket merges: long polling --\n");
    run<0, 16, 1>("plain / sc1, first merges", part, flags, tick, stamps, report, L, epoch);
    run<0, 2, 1>("plain / nt, first merges", part, flags, tick, stamps, report, L, epoch);
    run<0, 1, 1>("plain / sc0, first merges", part, flags, tick, stamps, report, 20, epoch);
    run<16, 16, 1>("sc1 / sc1, first merges", part, flags, tick, stamps, report, L, epoch);
    printf("-- groups spread over all 8 XCDs --\n");
    run<16, 16, 0, 1>("MIXED sc1 / sc1", part, flags, tick, stamps, report, L, epoch);
    run<16, 16, 1, 1>("MIXED sc1 / sc1, first merges", part, flags, tick, stamps, report, L, epoch);
    run<0, 16, 0, 1>("MIXED plain / sc1 (must fail)", part, flags, tick, stamps, report, 20, epoch);
    return 0;
}
