// attn_generic.hip — shape-generic fused decode attention (LUT-gather form), gfx950.
//
// Computes what the reference does with cuBLAS LUT matmul + flash_decoding_split_kernel
// (Kernel.cuh:11-166) + flash_decoding_residual_kernel (:1038-1209) + flash_decoding_reduce_kernel
// (:1211-1270), in ONE launch and with fp32 accumulation: per (b, hk, split) workgroup a per-query LUT
// of q.centroid dot products is built in LDS, scores are gathered from it per code byte, softmax is
// done with wave/block reductions (no serial thread-0 loops), values are reconstructed on the fly from
// the value codebook, and the last-arriving workgroup of each (b, hk) merges the split partials and the
// residual-window partial.  This kernel handles every (d, M, C) of the binding surface; the MFMA
// kernel (attn_mfma.hip) takes the headline shapes.
#include "common.h"

namespace million {

constexpr int kGenBlock = 256;
constexpr int kGenChunk = 2048;   // tokens scored per pass (S buffer in LDS)

__device__ __forceinline__ const uint8_t *k_row_ptr(const AttnParams &p, int b, int hk, int bh, int t) {
    if (p.k_paged) {
        const long long pid = k_page_id(p, bh, t / p.page_size);
        return p.k_codes + (pid * p.page_size + (t % p.page_size)) * p.M;
    }
    return p.k_codes + b * p.k_sb + hk * p.k_sh + (long long)t * p.M;
}
__device__ __forceinline__ uint8_t v_code_at(const AttnParams &p, int b, int hk, int bh, int t, int m) {
    if (p.v_paged) {
        const long long pid = v_page_id(p, bh, t / p.page_size);
        return p.v_codes[(pid * p.M + m) * p.page_size + (t % p.page_size)];
    }
    return p.v_codes[b * p.v_sb + hk * p.v_sh + (long long)t * p.M + m];
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float *red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

__global__ __launch_bounds__(kGenBlock) void attn_generic_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int slot = blockIdx.x;               // [0, nsplit) code splits, nsplit = residual window
    const int bh = blockIdx.y;
    const int b = bh / p.nh_k, hk = bh % p.nh_k;
    const int G = p.G, d = p.d, M = p.M, C = p.C, dm = p.dm;
    int T, r, rstart;
    load_lengths(p, b, T, r, rstart);
    const int r_old = r;
    if (p.k_new) r += 1;                      // fused append: the new token is window row r_old

    float *lut = (float *)smem;                          // [M*C]
    float *S = lut + M * C;                              // [kGenChunk]
    float *part = S + kGenChunk;                         // [G*d + 2G]
    float *obuf = part + (G * d + 2 * G + 3) / 4 * 4;    // [kGenBlock]
    float *red = obuf + kGenBlock;                       // [8]
    float *qs = red + 8;                                 // [d]
    int *flag = (int *)(qs + d);

    const bool is_resid = (slot == p.nsplit);
    const int t0 = is_resid ? 0 : min(slot * p.split_len, T);
    const int t1 = is_resid ? r : min(t0 + p.split_len, T);
    const int nph = kGenBlock / d;                       // token phases of the V pass
    const int vdim = tid % d, vph = tid / d;
    const int vm = vdim / dm, vk = vdim % dm;

    if (p.k_new && is_resid && tid < d) {           // fused append: park the new row in the window
        const long long o = b * p.res_sb + hk * p.res_sh + (long long)((rstart + r_old) % p.rcap) * d + tid;
        p.k_res_w[o] = p.k_new[(long long)bh * d + tid];
        p.v_res_w[o] = p.v_new[(long long)bh * d + tid];
    }
    for (int g = 0; g < G; ++g) {
        const int h = head0(p, hk) + g;
        const f16 *qv = p.q + ((long long)b * p.nh + h) * d;
        __syncthreads();
        for (int i = tid; i < d; i += kGenBlock) qs[i] = (float)qv[i];
        __syncthreads();
        if (!is_resid) {
            // LUT (reference: at::matmul, Interface.template.cu:49-50): lut[m][c] = q[m,:] . k_cents[m,c,:]
            for (int i = tid; i < M * C; i += kGenBlock) {
                const int m = i / C;
                float a = 0.f;
                for (int k = 0; k < dm; ++k) a = fmaf(qs[m * dm + k], (float)p.k_tab[(long long)i * dm + k], a);
                lut[i] = a;
            }
        }
        __syncthreads();

        float m_run = -INFINITY, l_run = 0.f, acc = 0.f;
        for (int c0 = t0; c0 < t1; c0 += kGenChunk) {
            const int cn = min(kGenChunk, t1 - c0);
            // ---- scores ----
            float lmax = -INFINITY;
            for (int i = tid; i < cn; i += kGenBlock) {
                float s;
                if (!is_resid) {
                    const uint8_t *row = k_row_ptr(p, b, hk, bh, c0 + i);
                    float a = 0.f;
                    for (int m = 0; m < M; m += 4) {
                        const uint32_t w = *(const uint32_t *)(row + m);
                        a += lut[(m + 0) * C + (w & 0xff)];
                        a += lut[(m + 1) * C + ((w >> 8) & 0xff)];
                        a += lut[(m + 2) * C + ((w >> 16) & 0xff)];
                        a += lut[(m + 3) * C + (w >> 24)];
                    }
                    s = a * p.scale_log2e;
                } else {
                    const int row = (rstart + c0 + i) % p.rcap;
                    const f16 *kr = (p.k_new && c0 + i == r_old) ? p.k_new + (long long)bh * d
                                                                 : p.k_res + b * p.res_sb + hk * p.res_sh + (long long)row * d;
                    float a = 0.f;
                    for (int k = 0; k < d; ++k) a = fmaf(qs[k], (float)kr[k], a);
                    s = a * p.scale_log2e;
                }
                S[i] = s;
                lmax = fmaxf(lmax, s);
            }
            const float cmax = block_reduce(lmax, true, red);
            const float m_new = fmaxf(m_run, cmax);
            const float alpha = (m_run > -INFINITY) ? exp2f(m_run - m_new) : 0.f;
            float lsum = 0.f;
            for (int i = tid; i < cn; i += kGenBlock) {
                const float pe = exp2f(S[i] - m_new);
                S[i] = pe;
                lsum += pe;
            }
            const float csum = block_reduce(lsum, false, red);   // barriers inside publish S
            l_run = l_run * alpha + csum;
            m_run = m_new;
            // ---- values: thread = (dim, token phase) ----
            acc *= alpha;
            if (vph < nph) {
                if (!is_resid) {
                    for (int i = vph; i < cn; i += nph) {
                        const int code = v_code_at(p, b, hk, bh, c0 + i, vm);
                        acc = fmaf(S[i], (float)p.v_tab[((long long)vm * C + code) * dm + vk], acc);
                    }
                } else {
                    for (int i = vph; i < cn; i += nph) {
                        const int row = (rstart + c0 + i) % p.rcap;
                        const f16 vv = (p.v_new && c0 + i == r_old) ? p.v_new[(long long)bh * d + vdim]
                                                                    : p.v_res[b * p.res_sb + hk * p.res_sh + (long long)row * d + vdim];
                        acc = fmaf(S[i], (float)vv, acc);
                    }
                }
            }
            __syncthreads();
        }
        // combine token phases
        obuf[tid] = acc;
        __syncthreads();
        if (tid < d) {
            float o = 0.f;
            for (int ph = 0; ph < nph; ++ph) o += obuf[ph * d + tid];
            part[g * d + tid] = o;
        }
        if (tid == 0) {
            part[G * d + g] = m_run;
            part[G * d + G + g] = l_run;
        }
        __syncthreads();
    }
    __syncthreads();
    publish_and_merge(p, b, hk, slot, part, lut, flag);   // the LUT region is dead: merge scratch
}

int launch_attn_generic(const AttnParams &p, hipStream_t s) {
    const size_t lds = sizeof(float) * ((size_t)p.M * p.C + kGenChunk + (p.G * p.d + 2 * p.G + 3) / 4 * 4 +
                                        kGenBlock + 8 + p.d) + 16;
    if (kGenBlock % p.d != 0) { set_error("generic kernel: d=%d must divide %d", p.d, kGenBlock); return MILLION_ERR_SHAPE; }
    if (lds > 160 * 1024) { set_error("generic kernel: LUT of M*C=%d floats does not fit LDS", p.M * p.C); return MILLION_ERR_SHAPE; }
    if (device_once(0))
        (void)hipFuncSetAttribute((const void *)attn_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    dim3 grid(p.nsplit + 1, p.bs * p.nh_k);
    hipLaunchKernelGGL(attn_generic_kernel, grid, dim3(kGenBlock), lds, s, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("attn_generic launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

}  // namespace million
