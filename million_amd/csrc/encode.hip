// encode.hip — PQ encode (argmin over centroids per sub-vector), gfx950.
//
// Replaces sa_encode_4d_keops (reference scripts/utils/pq_utils.py:451-499: fp32 upcast :483-484,
// ((x-c)**2).sum(-1).argmin :491-494) and the permute/cat that stores the codes (:497-499,
// paged_pq_utils.py:162,173-175): codes are written straight into their final place — row-major K
// store, K page pool, or transposed V page pool.
//
// Arithmetic contract (bit-exact with oracle/pq_oracle.c:pq_encode_direct): e = x - c, sq = e * e,
// acc = sq_0 + sq_1 + ... sequentially, every operation one IEEE fp32 round-to-nearest, no FMA
// contraction; strict '<' scan over increasing c, so the lowest index wins exact ties.
//
// Mapping: wave = one subspace m, lane = one token.  The centroid row of m (C*d_m halfs) is wave-uniform
// and comes through the scalar cache; the per-lane state is d_m fp32 values, the running best distance
// and its index.
#include "common.h"

#pragma clang fp contract(off)

namespace million {


constexpr int kEncBlock = 256;

template <int DM>
__global__ __launch_bounds__(kEncBlock) void pq_encode_kernel(EncParams p) {
    const int lane = threadIdx.x & 63;
    const int m = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * (kEncBlock / 64) + (threadIdx.x >> 6)));
    const int bh = blockIdx.z;
    const int b = bh / p.nh_k, hk = bh % p.nh_k;
    const int t = blockIdx.x * 64 + lane;
    if (m >= p.M) return;
    const bool valid = t < p.n;
    const int tc = valid ? t : p.n - 1;
    int tok0 = p.tok0, xrow_start = p.xrow_start;
    if (p.dev_lengths) { tok0 = p.dev_lengths[b * 4 + 0]; xrow_start = p.dev_lengths[b * 4 + 2]; }
    const int xrow = p.xrow_mod > 0 ? (xrow_start + tc) % p.xrow_mod : tc;
    const f16 *xp = p.x + b * p.xsb + hk * p.xsh + (long long)xrow * p.xsn + m * DM;
    float x[DM];
#pragma unroll
    for (int k = 0; k < DM; ++k) x[k] = (float)xp[k];

    const f16 *cm = p.cents + (long long)m * p.C * DM;   // wave-uniform
    float best = INFINITY;
    int best_c = 0;
    for (int c = 0; c < p.C; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            const float e = x[k] - (float)cm[c * DM + k];
            const float sq = e * e;
            acc = (k == 0) ? sq : acc + sq;
        }
        if (acc < best) { best = acc; best_c = c; }
    }
    if (!valid) return;
    const int tok = tok0 + t;
    if (p.layout == MILLION_CODES_ROWMAJOR) {
        p.dst[b * p.dsb + hk * p.dsh + (long long)tok * p.M + m] = (uint8_t)best_c;
    } else {
        const long long pid = p.page_ids[(long long)bh * p.n_pages_cap + tok / p.page_size];
        const int off = tok % p.page_size;
        if (p.layout == MILLION_CODES_KPAGES)
            p.dst[(pid * p.page_size + off) * p.M + m] = (uint8_t)best_c;
        else
            p.dst[(pid * p.M + m) * p.page_size + off] = (uint8_t)best_c;
    }
}

int launch_encode(const EncParams &p, hipStream_t s) {
    if (p.n <= 0 || p.bs * p.nh_k <= 0) return MILLION_OK;
    dim3 grid((p.n + 63) / 64, (p.M + kEncBlock / 64 - 1) / (kEncBlock / 64), p.bs * p.nh_k);
    switch (p.dm) {
        case 1: hipLaunchKernelGGL(pq_encode_kernel<1>, grid, dim3(kEncBlock), 0, s, p); break;
        case 2: hipLaunchKernelGGL(pq_encode_kernel<2>, grid, dim3(kEncBlock), 0, s, p); break;
        case 4: hipLaunchKernelGGL(pq_encode_kernel<4>, grid, dim3(kEncBlock), 0, s, p); break;
        case 8: hipLaunchKernelGGL(pq_encode_kernel<8>, grid, dim3(kEncBlock), 0, s, p); break;
        default: set_error("encode: d/M=%d unsupported (1,2,4,8)", p.dm); return MILLION_ERR_SHAPE;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("encode launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

}  // namespace million
