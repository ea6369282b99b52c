// HBM-bound element-wise / reduction kernels of the MFCNet hot path (gfx950).
// All tensors NHWC with 16-byte granules; one thread moves one granule (coalesced 16-B accesses).
//   mfc_bn_finalize      statistics -> BatchNorm coefficients + running-stat update
//   mfc_combine_fwd      relu?(sum_k affine_k?(bilinear_k?(src_k)))      (BN-apply/ReLU/residual/fuse/up-sample)
//   mfc_bnbwd_*          BatchNorm+ReLU backward: reduce / finalize / apply
//   mfc_mask_add         gradient of identity / up-sampled combine terms (adjoint bilinear gather)
//   mfc_bias_grad        per-channel sum
//   layout bridges, head gather (x4 up-sample + temporal concat [+ flow warp]) forward/backward
// Reference operators replaced: see include/mfcnet_hip.h.
#include "common.h"

// event-profiler bracket of an element-wise launch: the row is named like the kernel rocprofv3 reports; BYTES = the tensors the launch
// must touch once (its algorithmic HBM traffic)
#define EW_PROF(st, KERNEL, DTYPE, BYTES) do { if (g_mfc_prof_on == 1) mfc_prof_before(st, (DTYPE) == MFC_BF16 ? KERNEL "<__bf16>" : ((DTYPE) == MFC_F16 ? KERNEL "<_Float16>" : KERNEL "<float>"), 0.0, (double)(BYTES)); } while (0)
static inline double view_bytes(const mfc_view& v, int N, int C, int esz) { return v.ptr ? (double)N * v.H * v.W * C * esz : 0.0; }

// ------------------------------------------------------------------ BN finalize
// sum of the R = MFC_STAT_REPLICAS replica rows of one (group, stat, channel) cell: all loads in flight at once
__device__ inline double replica_sum(const mfc_stat_t* base, size_t stride) {
    double p[MFC_R];
#pragma unroll
    for (int r = 0; r < MFC_R; ++r) p[r] = base[(size_t)r * stride];
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < MFC_R; ++r) s += p[r];
    return s;
}

// block = 64 channels x 4 group slots; grid = ceil(Cp / 64)
__device__ inline void bn_finalize_body(const mfc_bnfin_desc& d, int bxi, float (*sm)[64], float (*sv)[64]) {
    const size_t rstride = (size_t)d.G * 2 * d.Cp;
    const int cl = threadIdx.x & 63, gs = threadIdx.x >> 6;
    const int c = bxi * 64 + cl;
    if (c < d.Cp) {
        const bool real = c < d.C;
        const float gamma = real ? d.gamma[c] : 0.f, beta = real ? d.beta[c] : 0.f;
        for (int g = gs; g < d.G; g += 4) {
            float* cf = d.coef + (size_t)g * 4 * d.Cp + c;
            float mean = 0.f, rstd = 0.f, var = 0.f;
            if (real) {
                if (d.training) {
                    const double s = replica_sum(d.stats + ((size_t)g * 2 + 0) * d.Cp + c, rstride);
                    const double s2 = replica_sum(d.stats + ((size_t)g * 2 + 1) * d.Cp + c, rstride);
                    const double m = s / (double)d.count;
                    double v = s2 / (double)d.count - m * m;
                    if (v < 0.0) v = 0.0;
                    mean = (float)m; var = (float)v; rstd = (float)(1.0 / sqrt(v + (double)d.eps));
                } else {
                    mean = d.running_mean[c]; rstd = 1.0f / sqrtf(d.running_var[c] + d.eps);
                }
            }
            const float scale = gamma * rstd;
            cf[0] = scale; cf[d.Cp] = beta - mean * scale; cf[2 * d.Cp] = mean; cf[3 * d.Cp] = rstd;
            sm[g][cl] = mean; sv[g][cl] = var;
        }
    }
    if (!d.training) return;
    __syncthreads();
    // running statistics, group after group (the reference calls base_model once per frame)
    if (gs == 0 && c < d.C) {
        float rm = d.running_mean[c], rv = d.running_var[c];
        for (int g = 0; g < d.G; ++g) {
            const double var = (double)sv[g][cl];
            const double unb = d.count > 1.f ? var * (double)d.count / ((double)d.count - 1.0) : var;
            rm = (1.f - d.momentum) * rm + d.momentum * sm[g][cl];
            rv = (1.f - d.momentum) * rv + d.momentum * (float)unb;
        }
        d.running_mean[c] = rm; d.running_var[c] = rv;
    }
    if (d.num_batches_tracked && bxi == 0 && threadIdx.x == 0) *d.num_batches_tracked += d.G;
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(mfc_bnfin_desc d) {
    __shared__ float sm[8][64], sv[8][64];
    bn_finalize_body(d, blockIdx.x, sm, sv);
}
// one launch for a TABLE of finalizes (blockIdx.y = table row): the eval-mode records of a program depend on nothing but the
// running statistics, so the plan hoists all of them in front of the first convolution
__global__ __launch_bounds__(256) void bn_finalize_batch_kernel(const mfc_bnfin_desc* tab) {
    __shared__ float sm[8][64], sv[8][64];
    const mfc_bnfin_desc d = tab[blockIdx.y];
    if ((int)blockIdx.x * 64 >= d.Cp || d.G > 8 || d.G <= 0) return;      // (rows with more than 8 groups would overflow sm / sv: the host validates the table, mfc_bn_finalize rejects them)
    bn_finalize_body(d, blockIdx.x, sm, sv);
}

extern "C" int mfc_bn_finalize(const mfc_bnfin_desc* d, void* stream) {
    if (!d || !d->coef || !d->gamma || !d->beta || !d->running_mean || !d->running_var) return MFC_ERR_INVALID_ARG;
    if (d->training && !d->stats) return MFC_ERR_INVALID_ARG;
    if (d->C <= 0 || d->C > d->Cp || d->G <= 0) return MFC_ERR_INVALID_ARG;
    if (d->G > 8) return MFC_ERR_UNSUPPORTED;
    if (g_mfc_prof_on == 1) mfc_prof_before((hipStream_t)stream, "bn_finalize_kernel", 0.0, d->training ? (double)MFC_R * d->G * 2 * d->Cp * sizeof(mfc_stat_t) : 0.0);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((d->Cp + 63) / 64), dim3(256), 0, (hipStream_t)stream, *d);
    MFC_PROF_END((hipStream_t)stream);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

extern "C" int mfc_bn_finalize_batch(const mfc_bnfin_desc* table_dev, int32_t n, int32_t max_Cp, void* stream) {
    if (!table_dev || n <= 0 || max_Cp <= 0) return MFC_ERR_INVALID_ARG;
    if (n > 65535) return MFC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_finalize_batch_kernel, dim3((max_Cp + 63) / 64, n), dim3(256), 0, (hipStream_t)stream, table_dev);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// SiLU (nn.SiLU, resunet.py:67): x * sigmoid(x)
__device__ inline float silu_f(float x) { return x / (1.f + __expf(-x)); }

// ------------------------------------------------------------------ combine forward
template <typename T>
__device__ inline void load_gran_f(const mfc_view& v, int n, int h, int w, int ch, float* f) {
    const uint4 u = *(const uint4*)((const char*)v.ptr + ((((size_t)n * v.H + h) * v.W + w) * v.Cp + ch) * sizeof(T));
    Gran<T>::unpack(u, f);
}

template <typename T>
__global__ __launch_bounds__(256) void combine_fwd_kernel(mfc_combine_desc d, long total, int Cg, int wt) {
    constexpr int E = Gran<T>::E;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;      // (host guarantees total < 2^31: 32-bit index arithmetic)
    if (idx >= (unsigned)total) return;
    const int g = (int)(idx % (unsigned)Cg); unsigned pix = idx / (unsigned)Cg;
    const int w = (int)(pix % (unsigned)d.out.W); pix /= (unsigned)d.out.W;
    const int h = (int)(pix % (unsigned)d.out.H); const int n = (int)(pix / (unsigned)d.out.H);
    const int grp = n / d.images_per_group;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll 1
    for (int k = 0; k < d.nsrc; ++k) {
        const mfc_view& s = d.src[k];
        const int ch = s.c_off + g * E;
        float f[E];
        if (s.H == d.out.H && s.W == d.out.W) {
            load_gran_f<T>(s, n, h, w, ch, f);
        } else {
            int h0, h1, w0, w1; float lh, lw;
            bilin_src(h, s.H, d.out.H, h0, h1, lh);
            bilin_src(w, s.W, d.out.W, w0, w1, lw);
            float a[E], b[E], c[E], e4[E];
            load_gran_f<T>(s, n, h0, w0, ch, a); load_gran_f<T>(s, n, h0, w1, ch, b);
            load_gran_f<T>(s, n, h1, w0, ch, c); load_gran_f<T>(s, n, h1, w1, ch, e4);
#pragma unroll
            for (int e = 0; e < E; ++e)
                f[e] = (1.f - lh) * ((1.f - lw) * a[e] + lw * b[e]) + lh * ((1.f - lw) * c[e] + lw * e4[e]);
        }
        if (s.coef) {
            const float* cf = (const float*)s.coef + (size_t)grp * 4 * s.Cp + ch;
#pragma unroll
            for (int e = 0; e < E; ++e) f[e] = f[e] * cf[e] + cf[s.Cp + e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += f[e];
    }
    if (d.relu == 1) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = relu_nan(acc[e]);
    } else if (d.relu == 2) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = silu_f(acc[e]);
    }
    mfc_st16_if(((char*)d.out.ptr + ((((size_t)n * d.out.H + h) * d.out.W + w) * d.out.Cp + d.out.c_off + g * E) * sizeof(T)), Gran<T>::pack(acc), wt);
    if constexpr (E == 8) {
        if (d.maskbits) {
            unsigned b = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) b |= (acc[e] > 0.f ? 1u : 0u) << e;
            ((unsigned char*)d.maskbits)[((((size_t)n * d.out.H + h) * d.out.W + w) * d.out.Cp + d.out.c_off + g * E) >> 3] = (unsigned char)b;
        }
    }
}

// all sources at the output resolution (residual blocks, BN materialisation): linear pixel index, 2 granules per thread
// all sources at the output's resolution (residual adds, BatchNorm apply + ReLU): one-shot workgroups of 512 granules
template <typename T>
__global__ __launch_bounds__(256) void combine_same_kernel(mfc_combine_desc d, long total, int Cg, int wt) {
    constexpr int E = Gran<T>::E;
    constexpr int U = 2;
    const unsigned base = (blockIdx.x * 256u) * U + threadIdx.x;
    const unsigned ppg = (unsigned)d.images_per_group * d.out.H * d.out.W;
    uint4 r[U][4]; unsigned pix[U]; int gq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        unsigned idx = base + u * 256u;
        if (idx >= (unsigned)total) idx = (unsigned)total - 1;
        pix[u] = idx / (unsigned)Cg; gq[u] = (int)(idx - pix[u] * (unsigned)Cg);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < d.nsrc) r[u][k] = *(const uint4*)((const char*)d.src[k].ptr + ((size_t)pix[u] * d.src[k].Cp + d.src[k].c_off + gq[u] * E) * sizeof(T));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (base + u * 256u >= (unsigned)total) break;
        const int grp = (int)(pix[u] / ppg);
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < d.nsrc) {
                float f[E];
                Gran<T>::unpack(r[u][k], f);
                if (d.src[k].coef) {
                    const float* cf = (const float*)d.src[k].coef + (size_t)grp * 4 * d.src[k].Cp + d.src[k].c_off + gq[u] * E;
#pragma unroll
                    for (int e = 0; e < E; ++e) f[e] = f[e] * cf[e] + cf[d.src[k].Cp + e];
                }
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += f[e];
            }
        }
        if (d.relu == 1) {
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = relu_nan(acc[e]);
        } else if (d.relu == 2) {
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = silu_f(acc[e]);
        }
        mfc_st16_if(((char*)d.out.ptr + ((size_t)pix[u] * d.out.Cp + d.out.c_off + gq[u] * E) * sizeof(T)), Gran<T>::pack(acc), wt);
        if constexpr (E == 8) {
            if (d.maskbits) {       // the ReLU mask of the backward pass: one bit per element
                unsigned b = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) b |= (acc[e] > 0.f ? 1u : 0u) << e;
                ((unsigned char*)d.maskbits)[((size_t)pix[u] * d.out.Cp + d.out.c_off + gq[u] * E) >> 3] = (unsigned char)b;
            }
        }
    }
}

static bool view_ok(const mfc_view& v, int E) { return v.ptr && v.H > 0 && v.W > 0 && v.Cp % 8 == 0 && v.c_off % E == 0; }

extern "C" int mfc_combine_fwd(const mfc_combine_desc* d, void* stream) {
    if (!d || d->nsrc < 1 || d->nsrc > 4) return MFC_ERR_INVALID_ARG;
    const int E = mfc_is16(d->dtype) ? 8 : 4;
    if (!mfc_dtype_ok(d->dtype)) return MFC_ERR_INVALID_ARG;
    if (!view_ok(d->out, E) || d->C <= 0 || d->C % E || d->N <= 0 || d->images_per_group <= 0) return MFC_ERR_INVALID_ARG;
    if (d->out.c_off + d->C > d->out.Cp) return MFC_ERR_INVALID_ARG;
    if (!mfc_ptrs_ok(d->out.ptr, d->src[0].ptr, d->src[0].coef, d->src[1].ptr, d->src[1].coef, d->src[2].ptr, d->src[2].coef, d->src[3].ptr, d->src[3].coef,
                     d->maskbits)) return MFC_ERR_INVALID_ARG;
    for (int k = 0; k < d->nsrc; ++k)
        if (!view_ok(d->src[k], E) || d->src[k].c_off + d->C > d->src[k].Cp) return MFC_ERR_INVALID_ARG;
    const int Cg = d->C / E;
    const long total = (long)d->N * d->out.H * d->out.W * Cg;
    if (total >= (1L << 31) - 2048) return MFC_ERR_UNSUPPORTED;      // kernels index granules with 32 bits
    const int blocks = (int)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    bool same = true;
    for (int k = 0; k < d->nsrc; ++k) same = same && d->src[k].H == d->out.H && d->src[k].W == d->out.W;
    double cbytes = view_bytes(d->out, d->N, d->C, E == 8 ? 2 : 4) + (d->maskbits ? (double)total : 0.0);
    for (int k = 0; k < d->nsrc; ++k) cbytes += view_bytes(d->src[k], d->N, d->C, E == 8 ? 2 : 4);
    const int wt = mfc_wt_for((double)total * 16.0);                 // write-through stores for a large output (common.h)
    if (same) {
        const int b2 = (int)((total + 511) / 512);
        EW_PROF(st, "combine_same_kernel", d->dtype, cbytes);
        MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(combine_same_kernel<T_>, dim3(b2), dim3(256), 0, st, *d, total, Cg, wt));
        MFC_PROF_END(st);
        MFC_CHECK_LAUNCH();
        return MFC_OK;
    }
    EW_PROF(st, "combine_fwd_kernel", d->dtype, cbytes);
    MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(combine_fwd_kernel<T_>, dim3(blocks), dim3(256), 0, st, *d, total, Cg, wt));
    MFC_PROF_END(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// ------------------------------------------------------------------ BN backward
// linear-pixel granule access (all views of one BN-backward call share N, H, W, so a pixel index addresses them all)
template <typename T>
__device__ inline uint4 ld_lin(const mfc_view& v, long pix, int ch) {
    return *(const uint4*)((const char*)v.ptr + ((size_t)pix * v.Cp + ch) * sizeof(T));
}
// E consecutive floats of a coefficient row as 16-byte vector loads (rows are 32-byte aligned: Cp % 8 == 0, channel offsets % E == 0).
// Always loaded this way, unconditionally, ahead of the arithmetic: scalar reads of cf[e] inside the arms of a mode switch stop hipcc's
// vectoriser and make it branch and wait per element (bnbwd_apply_kernel went from 16 dwordx4 loads to 160 dword loads and 7x the time)
template <int E>
__device__ inline void ld_coef(const float* p, float* out) {
#pragma unroll
    for (int q = 0; q < E / 4; ++q) {
        const float4 v = *(const float4*)(p + 4 * q);
        out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
    }
}
// masked gradient of one granule: mode 0 -> g, mode 1 -> g * [mask_src > 0], mode 2 -> g * [y*scale+shift > 0]
// (cf = {scale[E], shift[E]} of the granule's channels, preloaded)
template <typename T>
__device__ inline void apply_mask(const mfc_bnbwd_desc& d, const uint4& mraw, const float* yv, const float* cf, float* gm) {
    constexpr int E = Gran<T>::E;
    if (d.mask_mode == 1) {
        float m[E];
        Gran<T>::unpack(mraw, m);
#pragma unroll
        for (int e = 0; e < E; ++e) gm[e] = m[e] > 0.f ? gm[e] : 0.f;
    } else if (d.mask_mode == 2) {
#pragma unroll
        for (int e = 0; e < E; ++e) gm[e] = (yv[e] * cf[e] + cf[E + e]) > 0.f ? gm[e] : 0.f;
    } else if (d.mask_mode == 3) {          // one bit per element, one byte per 8-channel granule (written by mfc_combine_fwd)
#pragma unroll
        for (int e = 0; e < E; ++e) gm[e] = ((mraw.x >> e) & 1u) ? gm[e] : 0.f;
    } else if (d.mask_mode == 4) {          // SiLU (resunet.py:67): d silu(z) / dz = s (1 + z (1 - s)), s = sigmoid(z), z = y*scale + shift
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float z = yv[e] * cf[e] + cf[E + e];
            const float sg = 1.f / (1.f + __expf(-z));
            gm[e] *= sg * (1.f + z * (1.f - sg));
        }
    }
}
// the byte of mask bits of granule (pixel, channel c) of a [.., Cp] tensor
__device__ inline unsigned ld_bits(const mfc_view& v, long pix, int c) { return ((const unsigned char*)v.ptr)[((size_t)pix * v.Cp + c) >> 3]; }

template <typename T, int BS>
__global__ __launch_bounds__(BS) void bnbwd_reduce_kernel(mfc_bnbwd_desc d, int Cg, int PPI, int pix_per_block, long pix_per_group, int ablate) {
    constexpr int E = Gran<T>::E;
    constexpr int U = 4;                       // pixels in flight per thread
    extern __shared__ float red[];             // [2 E][BS] partial sums of the block's threads
    const int grp = blockIdx.y;
    const int gi = threadIdx.x % Cg, prow = threadIdx.x / Cg;
    float s1[E], s2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const float* cfp = (const float*)d.y.coef + (size_t)grp * 4 * d.y.Cp + d.y.c_off + gi * E;
    float mean[E], rstd[E], cf[2 * E];
    ld_coef<E>(cfp, cf); ld_coef<E>(cfp + d.y.Cp, cf + E); ld_coef<E>(cfp + 2 * d.y.Cp, mean); ld_coef<E>(cfp + 3 * d.y.Cp, rstd);
    const long gbase = (long)grp * pix_per_group;
    const long p0 = (long)blockIdx.x * pix_per_block;
    long p1 = p0 + pix_per_block; if (p1 > pix_per_group) p1 = pix_per_group;
    const int ych = d.y.c_off + gi * E, gch = d.g.c_off + gi * E, mch = d.mask.c_off + gi * E;
    for (long pp = p0 + prow; pp < p1; pp += (long)U * PPI) {
        uint4 yr[U], gr[U], mr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long q = pp + (long)u * PPI;
            const long qq = q < p1 ? q : p0 + prow;      // clamp: branch-free loads, masked below
            yr[u] = ld_lin<T>(d.y, gbase + qq, ych);
            gr[u] = ld_lin<T>(d.g, gbase + qq, gch);
            if (d.mask_mode == 1) mr[u] = ld_lin<T>(d.mask, gbase + qq, mch);
            else if (d.mask_mode == 3) mr[u].x = (ablate & 4) ? 0xffu : ld_bits(d.mask, gbase + qq, mch);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pp + (long)u * PPI < p1) {
                float yv[E], gm[E];
                Gran<T>::unpack(yr[u], yv); Gran<T>::unpack(gr[u], gm);
                apply_mask<T>(d, mr[u], yv, cf, gm);
#pragma unroll
                for (int e = 0; e < E; ++e) { s1[e] += gm[e]; s2[e] += gm[e] * (yv[e] - mean[e]) * rstd[e]; }
                if (d.dy.ptr && !(ablate & 2)) {          // masked gradient for the identity branch of the same sum
                    char* o = (char*)d.dy.ptr + ((size_t)(gbase + pp + (long)u * PPI) * d.dy.Cp + d.dy.c_off + gi * E) * sizeof(T);
                    if (d.accumulate) {
                        float old[E];
                        Gran<T>::unpack(*(const uint4*)o, old);
#pragma unroll
                        for (int e = 0; e < E; ++e) gm[e] += old[e];
                    }
                    *(uint4*)(o) = Gran<T>::pack(gm);
                }
            }
        }
    }
    // block reduction over the pixel rows, then one atomic per (stat, channel): every thread parks its 2 E partial sums in LDS, ONE barrier,
    // and thread (k, g2) adds the PPI values of its column in a fixed order (round 3: the seven-barrier tree this replaces cost 3.5 us
    // of a 17-38 us launch, profiles/r03_bnbwd_reduce_ablation.txt)
    if (ablate & 1) { if (s1[0] == 123.f) d.bstats[0] = 1.0; return; }
#pragma unroll
    for (int e = 0; e < E; ++e) { red[(e * 2 + 0) * BS + threadIdx.x] = s1[e]; red[(e * 2 + 1) * BS + threadIdx.x] = s2[e]; }
    __syncthreads();
    if (ablate & 8) { if (red[threadIdx.x] == 123.f) d.bstats[0] = 1.0; return; }
    for (int i = threadIdx.x; i < 2 * E * Cg; i += blockDim.x) {
        const int k = i / Cg, g2 = i - k * Cg;        // k = e*2 + stat
        const float* col = red + k * BS + g2;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = 0;
        for (; r + 4 <= PPI; r += 4) { a0 += col[r * Cg]; a1 += col[(r + 1) * Cg]; a2 += col[(r + 2) * Cg]; a3 += col[(r + 3) * Cg]; }
        for (; r < PPI; ++r) a0 += col[r * Cg];
        const float tot = (a0 + a1) + (a2 + a3);
        const int G = gridDim.y;
        const int c = d.y.c_off + g2 * E + (k >> 1);
        const int rep = blockIdx.x % MFC_R;
        // (fp64 atomics execute at the memory side whatever their scope, MI355X_MICROARCH.md "Global float atomics": ~1.2 us of drain per
        //  launch plus the issue of 2 E Cg of them per workgroup; an XCD-local variant has nothing to gain)
        atomicAdd(d.bstats + (((size_t)rep * G + grp) * 2 + (k & 1)) * d.y.Cp + c, (mfc_stat_t)tot);
    }
}

int g_bnred_minpx = 8;               // fewest pixels per thread of a BN-backward reduce launch (fewer, longer workgroups on small tensors); mfc_set_flag(42, n)
int g_bnred_threads = 256;          // threads per workgroup of a BN-backward reduce launch (256; 512 / 1024 measured slower); tuning: mfc_set_flag(41, n)
int g_ew_ablate = 0;                 // tuning only: mfc_set_flag(40, mask) -- bnbwd_reduce: 1 no block reduction / atomics, 2 no masked-gradient store, 4 no mask-bit loads
int g_applyfin_blocks = 1024;        // workgroups of a fused finalize + apply launch; tuning: mfc_set_flag(39, n)
int g_bnred_blocks = 1024;           // workgroups of a BN-backward reduce launch; tuning: mfc_set_flag(27, n)
static int bnbwd_check(const mfc_bnbwd_desc* d, int& E) {
    if (!d || (!mfc_dtype_ok(d->dtype))) return MFC_ERR_INVALID_ARG;
    E = mfc_is16(d->dtype) ? 8 : 4;
    if (!view_ok(d->g, E) || !view_ok(d->y, E) || !d->y.coef || d->C <= 0 || d->C % E) return MFC_ERR_INVALID_ARG;
    if (d->g.H != d->y.H || d->g.W != d->y.W) return MFC_ERR_INVALID_ARG;
    if (d->mask_mode == 1 && (!view_ok(d->mask, E) || d->mask.H != d->y.H || d->mask.W != d->y.W)) return MFC_ERR_INVALID_ARG;
    if (d->mask_mode < 0 || d->mask_mode > 4) return MFC_ERR_INVALID_ARG;
    if (d->mask_mode == 3 && (!mfc_is16(d->dtype) || !d->mask.ptr || d->mask.Cp % 8 || d->mask.c_off % 8 || d->mask.H != d->y.H || d->mask.W != d->y.W)) return MFC_ERR_INVALID_ARG;
    if (d->N <= 0 || d->images_per_group <= 0 || d->N % d->images_per_group) return MFC_ERR_INVALID_ARG;
    return MFC_OK;
}

extern "C" int mfc_bnbwd_reduce(const mfc_bnbwd_desc* d, void* stream) {
    int E; int rc = bnbwd_check(d, E); if (rc < 0) return rc;
    if (!d->bstats) return MFC_ERR_INVALID_ARG;
    if (!mfc_ptrs_ok(d->g.ptr, d->y.ptr, d->y.coef, d->mask.ptr, d->dy.ptr, d->bstats)) return MFC_ERR_INVALID_ARG;
    if (d->dy.ptr && (!view_ok(d->dy, E) || d->dy.H != d->y.H || d->dy.W != d->y.W)) return MFC_ERR_INVALID_ARG;
    const int Cg = d->C / E;
    if (Cg > 256) return MFC_ERR_UNSUPPORTED;
    // The launch ends with one fp64 atomic per workgroup, statistic and channel, and the atomics on one cell run one after the other at the
    // memory side (the XCDs' L2s are not coherent): ablation (profiles/r03_bnbwd_reduce_ablation.txt) puts that tail at ~4 us of a 17-38 us
    // launch, next to ~3.5 us for the old seven-barrier LDS tree.  Larger workgroups (512 / 1024 threads, flag 41) cut the atomics but lose
    // more in the streaming phase (158 VGPRs: 1024 threads spill); small tensors get fewer, longer workgroups instead (flag 42)
    const int BS = g_bnred_threads >= 1024 ? 1024 : (g_bnred_threads >= 512 ? 512 : 256);
    const int PPI = BS / Cg;
    const int G = d->N / d->images_per_group;
    const long ppg = (long)d->images_per_group * d->y.H * d->y.W;
    const int nblk = g_bnred_blocks * 256 / BS;
    long want = ppg / ((long)g_bnred_minpx * PPI) + 1; if (want > nblk / G + 1) want = nblk / G + 1;   // blocks per group (>= g_bnred_minpx pixels per thread)
    int ppb = (int)((ppg + want - 1) / want);
    ppb = ((ppb + PPI - 1) / PPI) * PPI;
    const int bx = (int)((ppg + ppb - 1) / ppb);
    hipStream_t st = (hipStream_t)stream;
    {
        const int esz = E == 8 ? 2 : 4;
        const double t = (double)d->N * d->y.H * d->y.W * d->C * esz;         // one tensor
        EW_PROF(st, "bnbwd_reduce_kernel", d->dtype, t * (2 + (d->mask_mode == 1 ? 1 : 0) + (d->dy.ptr ? (d->accumulate ? 2 : 1) : 0)) + (d->mask_mode == 3 ? t / 16 : 0));
    }
    const size_t lds = (size_t)2 * E * BS * sizeof(float);
#define RED_LAUNCH(BS_) do { \
        static bool attr_set = false; \
        if (!attr_set) { \
            (void)hipFuncSetAttribute((const void*)bnbwd_reduce_kernel<float, BS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
            (void)hipFuncSetAttribute((const void*)bnbwd_reduce_kernel<bf16_t, BS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
            (void)hipFuncSetAttribute((const void*)bnbwd_reduce_kernel<f16_t, BS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
            attr_set = true; \
        } \
        MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL((bnbwd_reduce_kernel<T_, BS_>), dim3(bx, G), dim3(Cg * PPI), lds, st, *d, Cg, PPI, ppb, ppg, g_ew_ablate)); \
    } while (0)
    if (BS == 1024) RED_LAUNCH(1024); else if (BS == 512) RED_LAUNCH(512); else RED_LAUNCH(256);
#undef RED_LAUNCH
    MFC_PROF_END(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

__global__ __launch_bounds__(256) void bnbwd_finalize_kernel(mfc_bnbwdfin_desc d) {
    __shared__ float sh[2][8][64];           // per (stat, group <= 8, channel) sums for the gamma / beta gradients
    const size_t rstride = (size_t)d.G * 2 * d.Cp;
    const int cl = threadIdx.x & 63, gs = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (c < d.Cp) {
        for (int g = gs; g < d.G; g += 4) {
            double s1 = 0.0, s2 = 0.0;
            if (c < d.C) {
                s1 = replica_sum(d.bstats + ((size_t)g * 2 + 0) * d.Cp + c, rstride);
                s2 = replica_sum(d.bstats + ((size_t)g * 2 + 1) * d.Cp + c, rstride);
            }
            d.bcoef[((size_t)g * 2 + 0) * d.Cp + c] = d.training ? (float)(s1 / (double)d.count) : 0.f;
            d.bcoef[((size_t)g * 2 + 1) * d.Cp + c] = d.training ? (float)(s2 / (double)d.count) : 0.f;
            sh[0][g][cl] = (float)s1; sh[1][g][cl] = (float)s2;
        }
    }
    __syncthreads();
    if (gs == 0 && c < d.C) {
        double db = 0.0, dg = 0.0;
        for (int g = 0; g < d.G; ++g) { db += (double)sh[0][g][cl]; dg += (double)sh[1][g][cl]; }
        d.dgamma[c] = (float)dg; d.dbeta[c] = (float)db;
    }
}

extern "C" int mfc_bnbwd_finalize(const mfc_bnbwdfin_desc* d, void* stream) {
    if (!d || !d->bstats || !d->bcoef || !d->dgamma || !d->dbeta || d->C <= 0 || d->C > d->Cp || d->G <= 0) return MFC_ERR_INVALID_ARG;
    if (d->G > 8) return MFC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bnbwd_finalize_kernel, dim3((d->Cp + 63) / 64), dim3(256), 0, (hipStream_t)stream, *d);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void bnbwd_apply_kernel(mfc_bnbwd_desc d, long total, int Cg, int wt) {
    constexpr int E = Gran<T>::E;
    constexpr int U = 4;                                   // granules in flight per thread
    const unsigned base = (blockIdx.x * 256u) * U + threadIdx.x;      // (host guarantees total < 2^31)
    const unsigned ppg = (unsigned)d.images_per_group * d.y.H * d.y.W;
    uint4 yr[U], gr[U], mr[U]; unsigned pix[U]; int gq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        unsigned idx = base + u * 256u;
        if (idx >= (unsigned)total) idx = (unsigned)total - 1;   // clamp: branch-free loads, masked at the store
        pix[u] = idx / (unsigned)Cg; gq[u] = (int)(idx - pix[u] * (unsigned)Cg);
        yr[u] = ld_lin<T>(d.y, pix[u], d.y.c_off + gq[u] * E);
        gr[u] = ld_lin<T>(d.g, pix[u], d.g.c_off + gq[u] * E);
        if (d.mask_mode == 1) mr[u] = ld_lin<T>(d.mask, pix[u], d.mask.c_off + gq[u] * E);
        else if (d.mask_mode == 3) mr[u].x = ld_bits(d.mask, pix[u], d.mask.c_off + gq[u] * E);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (base + u * 256u >= (unsigned)total) break;
        const int grp = (int)(pix[u] / ppg);
        const int c = d.y.c_off + gq[u] * E;
        const float* cfp = (const float*)d.y.coef + (size_t)grp * 4 * d.y.Cp + c;
        const float* bcp = d.bcoef + (size_t)grp * 2 * d.y.Cp + c;
        float cf[2 * E], mu[E], rs[E], b1[E], b2[E];
        ld_coef<E>(cfp, cf); ld_coef<E>(cfp + d.y.Cp, cf + E); ld_coef<E>(cfp + 2 * d.y.Cp, mu); ld_coef<E>(cfp + 3 * d.y.Cp, rs);
        ld_coef<E>(bcp, b1); ld_coef<E>(bcp + d.y.Cp, b2);
        float yv[E], gm[E], o[E];
        Gran<T>::unpack(yr[u], yv); Gran<T>::unpack(gr[u], gm);
        apply_mask<T>(d, mr[u], yv, cf, gm);
        // GroupNorm (gn_mode): the two means span the channels of a group, so they are not scaled by the channel's gamma:
        // dy = gamma rstd g - rstd (A + yhat B), bcoef = A, B of the channel's group (mfc_gnbwd_finalize).  Written as ONE formula
        // dy = scale * g - k * (b1 + yhat * b2), k = scale (BatchNorm) or rstd (GroupNorm), with every coefficient loaded
        // unconditionally: loads inside the arms of a select make hipcc branch and wait per element (7x slower, measured)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float yh = (yv[e] - mu[e]) * rs[e];
            o[e] = d.gn_mode ? (cf[e] * gm[e] - rs[e] * (b1[e] + yh * b2[e])) : cf[e] * (gm[e] - b1[e] - yh * b2[e]);
        }
        mfc_st16_if(((char*)d.dy.ptr + ((size_t)pix[u] * d.dy.Cp + d.dy.c_off + gq[u] * E) * sizeof(T)), Gran<T>::pack(o), wt);
    }
}

// apply with the finalize fused in: prologue = replica sums -> c1, c2 in LDS (and dgamma / dbeta from workgroup 0); then the same
// element-wise pass.  Every workgroup owns ONE contiguous range of `per_block` granules (a multiple of 8 = whole 128-byte lines; equal
// ranges, so the 4 workgroups of every CU finish together -- with chunks dealt round-robin some workgroups had 2 chunks and some 1),
// and the loads of its first 1024 granules are issued BEFORE the prologue, whose dependent fp64 replica reads (2-3 us of latency at
// the head of a 10-25 us launch) then run under them.
template <typename T>
__global__ __launch_bounds__(256) void bnbwd_apply_fin_kernel(mfc_bnbwd_desc d, unsigned total, int Cg, int G, unsigned per_block, int wt) {
    constexpr int E = Gran<T>::E;
    constexpr int U = 4;
    __shared__ float lbc[8 * 2 * 128];               // [g][stat][c] (G <= 8, C <= 128): c1 = mean(g*m), c2 = mean(g*m*yhat)
    const int Cs = Cg * E, Cp = d.y.Cp;
    const unsigned lo = blockIdx.x * per_block;
    const unsigned hi = (lo + per_block < total) ? lo + per_block : total;          // (host: grid * per_block >= total, lo < total)
    uint4 yr[U], gr[U], mr[U]; unsigned pix[U]; int gq[U];
    auto load = [&](unsigned base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned idx = base + u * 256u;
            if (idx >= hi) idx = hi - 1;             // clamp: branch-free loads, masked at the store
            pix[u] = idx / (unsigned)Cg; gq[u] = (int)(idx - pix[u] * (unsigned)Cg);
            yr[u] = ld_lin<T>(d.y, pix[u], d.y.c_off + gq[u] * E);
            gr[u] = ld_lin<T>(d.g, pix[u], d.g.c_off + gq[u] * E);
            if (d.mask_mode == 1) mr[u] = ld_lin<T>(d.mask, pix[u], d.mask.c_off + gq[u] * E);
            else if (d.mask_mode == 3) mr[u].x = ld_bits(d.mask, pix[u], d.mask.c_off + gq[u] * E);
        }
    };
    load(lo + threadIdx.x);
    const size_t rstride = (size_t)G * 2 * Cp;
    for (int i = threadIdx.x; i < G * 2 * Cs; i += 256) {
        const int c = i % Cs, gs = i / Cs;            // gs = g*2 + stat
        double s = 0.0;
        if (c < d.fin_C) s = replica_sum(d.bstats + (size_t)gs * Cp + d.y.c_off + c, rstride);
        lbc[i] = (float)s;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < d.fin_C; c += 256) {
            double db = 0.0, dg = 0.0;
            for (int g = 0; g < G; ++g) { db += (double)lbc[(g * 2 + 0) * Cs + c]; dg += (double)lbc[(g * 2 + 1) * Cs + c]; }
            d.fin_dgamma[c] = (float)dg; d.fin_dbeta[c] = (float)db;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < G * 2 * Cs; i += 256)
        lbc[i] = d.fin_training ? (float)((double)lbc[i] / (double)d.fin_count) : 0.f;
    __syncthreads();
    const unsigned ppg = (unsigned)d.images_per_group * d.y.H * d.y.W;
    for (unsigned base = lo + threadIdx.x; ; ) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * 256u >= hi) break;
            const int grp = (int)(pix[u] / ppg);
            const int c = d.y.c_off + gq[u] * E;
            const float* cfp = (const float*)d.y.coef + (size_t)grp * 4 * Cp + c;
            const float* b1 = lbc + (grp * 2 + 0) * Cs + gq[u] * E;
            const float* b2 = lbc + (grp * 2 + 1) * Cs + gq[u] * E;
            float cf[2 * E], mu[E], rs[E];
            ld_coef<E>(cfp, cf); ld_coef<E>(cfp + Cp, cf + E); ld_coef<E>(cfp + 2 * Cp, mu); ld_coef<E>(cfp + 3 * Cp, rs);
            float yv[E], gm[E], o[E];
            Gran<T>::unpack(yr[u], yv); Gran<T>::unpack(gr[u], gm);
            apply_mask<T>(d, mr[u], yv, cf, gm);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float yh = (yv[e] - mu[e]) * rs[e];
                o[e] = cf[e] * (gm[e] - b1[e] - yh * b2[e]);
            }
            mfc_st16_if(((char*)d.dy.ptr + ((size_t)pix[u] * d.dy.Cp + d.dy.c_off + gq[u] * E) * sizeof(T)), Gran<T>::pack(o), wt);
        }
        base += 256u * U;
        if (base - threadIdx.x >= hi) break;          // (block-uniform)
        load(base);
    }
}

extern "C" int mfc_bnbwd_apply(const mfc_bnbwd_desc* d, void* stream) {
    int E; int rc = bnbwd_check(d, E); if (rc < 0) return rc;
    if (!view_ok(d->dy, E) || d->dy.H != d->y.H || d->dy.W != d->y.W) return MFC_ERR_INVALID_ARG;
    if (!mfc_ptrs_ok(d->g.ptr, d->y.ptr, d->y.coef, d->mask.ptr, d->dy.ptr, d->bstats, d->bcoef, d->fin_dgamma, d->fin_dbeta)) return MFC_ERR_INVALID_ARG;
    const int wt = mfc_wt_for((double)d->N * d->y.H * d->y.W * d->C * (E == 8 ? 2.0 : 4.0));      // write-through stores for a large output (common.h)
    if (d->fin_dgamma) {       // finalize fused into this launch
        if (d->gn_mode) return MFC_ERR_UNSUPPORTED;
        const int G = d->N / d->images_per_group;
        if (!d->bstats || !d->fin_dbeta || d->fin_C <= 0 || d->fin_C > d->C || d->C > 128 || G > 8 || d->fin_count <= 0.f) return MFC_ERR_INVALID_ARG;
        const int Cgf = d->C / E;
        const long tot = (long)d->N * d->y.H * d->y.W * Cgf;
        if (tot >= (1L << 31) - 2048) return MFC_ERR_UNSUPPORTED;
        const unsigned nchunks = (unsigned)((tot + 1023) / 1024);
        int grid = nchunks < (unsigned)g_applyfin_blocks ? (int)nchunks : g_applyfin_blocks;
        unsigned per_block = (unsigned)(((tot + grid - 1) / grid + 7) / 8 * 8);
        grid = (int)((tot + per_block - 1) / per_block);
        hipStream_t s2 = (hipStream_t)stream;
        EW_PROF(s2, "bnbwd_apply_fin_kernel", d->dtype, (double)tot * 16.0 * (3 + (d->mask_mode == 1 ? 1 : 0)) + (d->mask_mode == 3 ? (double)tot : 0.0));
        MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(bnbwd_apply_fin_kernel<T_>, dim3(grid), dim3(256), 0, s2, *d, (unsigned)tot, Cgf, G, per_block, wt));
        MFC_PROF_END(s2);
        MFC_CHECK_LAUNCH();
        return MFC_OK;
    }
    if (!d->bcoef) return MFC_ERR_INVALID_ARG;
    const int Cg = d->C / E;
    const long total = (long)d->N * d->y.H * d->y.W * Cg;
    if (total >= (1L << 31) - 2048) return MFC_ERR_UNSUPPORTED;      // kernels index granules with 32 bits
    const int blocks = (int)((total + 1023) / 1024);
    hipStream_t st = (hipStream_t)stream;
    EW_PROF(st, "bnbwd_apply_kernel", d->dtype, (double)total * 16.0 * (3 + (d->mask_mode == 1 ? 1 : 0)) + (d->mask_mode == 3 ? (double)total : 0.0));
    MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(bnbwd_apply_kernel<T_>, dim3(blocks), dim3(256), 0, st, *d, total, Cg, wt));
    MFC_PROF_END(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// ------------------------------------------------------------------ mask_add (+ adjoint bilinear)
// candidate high-res index range whose bilinear footprint can touch low-res index `s`
__device__ inline void adj_range(int s, int in_size, int out_size, int& lo, int& hi) {
    const float f = (float)out_size / (float)in_size;
    lo = (int)floorf(f * ((float)s - 0.5f) - 0.5f) - 1;
    hi = (int)ceilf(f * ((float)s + 1.5f) - 0.5f) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}

// adjoint of the bilinear up-sampling (gather form): S = 2^LS adjacent lanes share one low-resolution granule and split
// the candidate high-resolution rows between them (a x8 fuse-path adjoint has ~18x18 candidates per output granule);
// the separable column weights are computed once per 8-column chunk; partial sums meet through lane shuffles.
template <typename T>
__global__ __launch_bounds__(256) void mask_add_kernel(mfc_maskadd_desc d, long npixout, int Cg, int LS, int PB) {
    constexpr int E = Gran<T>::E;
    __shared__ float red[256 * E];
    // workgroup = PB output pixels x S column lanes x Cg granules (granule fastest: a row read touches S*Cg contiguous granules)
    const unsigned S = 1u << LS;
    const unsigned tid = threadIdx.x;
    const int g = (int)(tid % (unsigned)Cg);
    const unsigned t2 = tid / (unsigned)Cg, j = t2 & (S - 1);
    unsigned pix = blockIdx.x * (unsigned)PB + (t2 >> LS);
    const bool live = (t2 >> LS) < (unsigned)PB && pix < (unsigned)npixout;
    if (!live) pix = (unsigned)npixout - 1;
    const int w = (int)(pix % (unsigned)d.dst.W); pix /= (unsigned)d.dst.W;
    const int h = (int)(pix % (unsigned)d.dst.H); const int n = (int)(pix / (unsigned)d.dst.H);
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    int hlo, hhi, wlo, whi;
    adj_range(h, d.dst.H, d.g.H, hlo, hhi);
    adj_range(w, d.dst.W, d.g.W, wlo, whi);
    // the S lanes of an output granule take adjacent candidate COLUMNS (with the granule index fastest, a wave touches
    // S*Cg contiguous granules per row); the row weights are computed once per 8-row chunk
    for (int hh0 = hlo; hh0 <= hhi; hh0 += 8) {
        float wrow[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int hh = hh0 + q;
            int h0, h1; float lh;
            bilin_src(hh <= hhi ? hh : hhi, d.dst.H, d.g.H, h0, h1, lh);
            wrow[q] = hh <= hhi ? ((h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f)) : 0.f;
        }
        for (int ww = wlo + (int)j; ww <= whi; ww += (int)S) {
            int w0, w1; float lw;
            bilin_src(ww, d.dst.W, d.g.W, w0, w1, lw);
            const float wc = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
            if (wc == 0.f) continue;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float wt = wc * wrow[q];
                if (wt == 0.f) continue;
                float gv[E];
                load_gran_f<T>(d.g, n, hh0 + q, ww, d.g.c_off + g * E, gv);
                if (d.mask_mode == 1) {
                    float m[E];
                    load_gran_f<T>(d.mask, n, hh0 + q, ww, d.mask.c_off + g * E, m);
#pragma unroll
                    for (int e = 0; e < E; ++e) gv[e] = m[e] > 0.f ? gv[e] : 0.f;
                }
                if (d.mask_mode == 3) {
                    const unsigned mb = ld_bits(d.mask, ((long)n * d.mask.H + hh0 + q) * d.mask.W + ww, d.mask.c_off + g * E);
#pragma unroll
                    for (int e = 0; e < E; ++e) gv[e] = ((mb >> e) & 1u) ? gv[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += wt * gv[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) red[e * 256 + tid] = live ? acc[e] : 0.f;
    __syncthreads();
    if (!live || j != 0) return;
    for (unsigned k = 1; k < S; ++k) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += red[e * 256 + tid + k * Cg];
    }
    char* o = (char*)d.dst.ptr + ((((size_t)n * d.dst.H + h) * d.dst.W + w) * d.dst.Cp + d.dst.c_off + g * E) * sizeof(T);
    if (d.accumulate) {
        float old[E];
        Gran<T>::unpack(*(const uint4*)o, old);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += old[e];
    }
    *(uint4*)(o) = Gran<T>::pack(acc);
}

// Separable form of the same adjoint (the bilinear weights are a product of a row and a column factor):
//   pass W: tmp[n, h, wl, c]  = sum_ww  Uw[ww -> wl] * (g*m)[n, h, ww, c]      (streams the high-resolution gradient once,
//                                                                              neighbouring lanes read neighbouring granules)
//   pass H: dst[n, hl, wl, c] (+)= sum_hh Uh[hh -> hl] * tmp[n, hh, wl, c]
// tmp is fp32, so the result is rounded once like the one-pass gather.
template <typename T>
__global__ __launch_bounds__(256) void mask_add_wpass_kernel(mfc_maskadd_desc d, unsigned total, int Cg) {
    constexpr int E = Gran<T>::E;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;           // (n, h, wl, granule), granule fastest
    if (idx >= total) return;
    const int g = (int)(idx % (unsigned)Cg); unsigned r = idx / (unsigned)Cg;
    const int wl = (int)(r % (unsigned)d.dst.W); r /= (unsigned)d.dst.W;          // r = n * g.H + h
    int wlo, whi;
    adj_range(wl, d.dst.W, d.g.W, wlo, whi);
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    const char* grow = (const char*)d.g.ptr + ((size_t)r * d.g.W * d.g.Cp + d.g.c_off + g * E) * sizeof(T);
    const char* mrow = d.mask_mode == 1 ? (const char*)d.mask.ptr + ((size_t)r * d.mask.W * d.mask.Cp + d.mask.c_off + g * E) * sizeof(T) : nullptr;
    for (int ww0 = wlo; ww0 <= whi; ww0 += 4) {
        uint4 gr[4], mr[4]; float wt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                               // unconditional (clamped) loads, weight 0 outside the footprint
            const int ww = min(ww0 + q, whi);
            int w0, w1; float lw;
            bilin_src(ww, d.dst.W, d.g.W, w0, w1, lw);
            wt[q] = (ww0 + q <= whi) ? ((w0 == wl ? 1.f - lw : 0.f) + (w1 == wl ? lw : 0.f)) : 0.f;
            gr[q] = *(const uint4*)(grow + (size_t)ww * d.g.Cp * sizeof(T));
            if (d.mask_mode == 1) mr[q] = *(const uint4*)(mrow + (size_t)ww * d.mask.Cp * sizeof(T));
            else if (d.mask_mode == 3) mr[q].x = ld_bits(d.mask, (long)r * d.mask.W + ww, d.mask.c_off + g * E);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float gv[E];
            Gran<T>::unpack(gr[q], gv);
            if (d.mask_mode == 1) {
                float m[E];
                Gran<T>::unpack(mr[q], m);
#pragma unroll
                for (int e = 0; e < E; ++e) gv[e] = m[e] > 0.f ? gv[e] : 0.f;
            }
            if (d.mask_mode == 3) {
#pragma unroll
                for (int e = 0; e < E; ++e) gv[e] = ((mr[q].x >> e) & 1u) ? gv[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += wt[q] * gv[e];
        }
    }
    float* o = d.scratch + (size_t)idx * E;
#pragma unroll
    for (int e = 0; e < E; e += 4) *(uint4*)(o + e) = make_uint4(__float_as_uint(acc[e]), __float_as_uint(acc[e + 1]), __float_as_uint(acc[e + 2]), __float_as_uint(acc[e + 3]));
}

template <typename T>
__global__ __launch_bounds__(256) void mask_add_hpass_kernel(mfc_maskadd_desc d, unsigned total, int Cg) {
    constexpr int E = Gran<T>::E;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;           // (n, hl, wl, granule)
    if (idx >= total) return;
    const unsigned rowg = (unsigned)d.dst.W * (unsigned)Cg;          // granules per tmp row
    const unsigned col = idx % rowg; unsigned r = idx / rowg;
    const int hl = (int)(r % (unsigned)d.dst.H); const int n = (int)(r / (unsigned)d.dst.H);
    int hlo, hhi;
    adj_range(hl, d.dst.H, d.g.H, hlo, hhi);
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    const float* base = d.scratch + ((size_t)n * d.g.H * rowg + col) * E;
    for (int hh = hlo; hh <= hhi; ++hh) {
        int h0, h1; float lh;
        bilin_src(hh, d.dst.H, d.g.H, h0, h1, lh);
        const float wh = (h0 == hl ? 1.f - lh : 0.f) + (h1 == hl ? lh : 0.f);
        if (wh == 0.f) continue;
        const float* p = base + (size_t)hh * rowg * E;
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            const float4 v = *(const float4*)(p + e);
            acc[e] += wh * v.x; acc[e + 1] += wh * v.y; acc[e + 2] += wh * v.z; acc[e + 3] += wh * v.w;
        }
    }
    const int g = (int)(col % (unsigned)Cg), wl = (int)(col / (unsigned)Cg);
    char* o = (char*)d.dst.ptr + ((((size_t)n * d.dst.H + hl) * d.dst.W + wl) * d.dst.Cp + d.dst.c_off + g * E) * sizeof(T);
    if (d.accumulate) {
        float old[E];
        Gran<T>::unpack(*(const uint4*)o, old);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += old[e];
    }
    *(uint4*)(o) = Gran<T>::pack(acc);
}

// same-resolution fast path: linear pixel index, 4 granules in flight per thread
template <typename T>
__global__ __launch_bounds__(256) void mask_add_same_kernel(mfc_maskadd_desc d, long total, int Cg) {
    constexpr int E = Gran<T>::E;
    constexpr int U = 4;
    const unsigned base = (blockIdx.x * 256u) * U + threadIdx.x;      // (host guarantees total < 2^31)
    uint4 gr[U], mr[U], dr[U]; unsigned pix[U]; int gq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        unsigned idx = base + u * 256u;
        if (idx >= (unsigned)total) idx = (unsigned)total - 1;
        pix[u] = idx / (unsigned)Cg; gq[u] = (int)(idx - pix[u] * (unsigned)Cg);
        gr[u] = ld_lin<T>(d.g, pix[u], d.g.c_off + gq[u] * E);
        if (d.mask_mode == 1) mr[u] = ld_lin<T>(d.mask, pix[u], d.mask.c_off + gq[u] * E);
        else if (d.mask_mode == 3) mr[u].x = ld_bits(d.mask, pix[u], d.mask.c_off + gq[u] * E);
        if (d.accumulate) dr[u] = ld_lin<T>(d.dst, pix[u], d.dst.c_off + gq[u] * E);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (base + u * 256u >= (unsigned)total) break;
        float gv[E];
        Gran<T>::unpack(gr[u], gv);
        if (d.mask_mode == 1) {
            float m[E];
            Gran<T>::unpack(mr[u], m);
#pragma unroll
            for (int e = 0; e < E; ++e) gv[e] = m[e] > 0.f ? gv[e] : 0.f;
        }
        if (d.mask_mode == 3) {
#pragma unroll
            for (int e = 0; e < E; ++e) gv[e] = ((mr[u].x >> e) & 1u) ? gv[e] : 0.f;
        }
        if (d.accumulate) {
            float o[E];
            Gran<T>::unpack(dr[u], o);
#pragma unroll
            for (int e = 0; e < E; ++e) gv[e] += o[e];
        }
        *(uint4*)(((char*)d.dst.ptr + ((size_t)pix[u] * d.dst.Cp + d.dst.c_off + gq[u] * E) * sizeof(T))) = Gran<T>::pack(gv);
    }
}

extern "C" int mfc_mask_add(const mfc_maskadd_desc* d, void* stream) {
    if (!d || (!mfc_dtype_ok(d->dtype))) return MFC_ERR_INVALID_ARG;
    const int E = mfc_is16(d->dtype) ? 8 : 4;
    if (!view_ok(d->g, E) || !view_ok(d->dst, E) || d->C <= 0 || d->C % E || d->N <= 0) return MFC_ERR_INVALID_ARG;
    if (d->mask_mode != 0 && d->mask_mode != 1 && d->mask_mode != 3) return MFC_ERR_INVALID_ARG;
    if (!mfc_ptrs_ok(d->g.ptr, d->mask.ptr, d->dst.ptr, d->scratch)) return MFC_ERR_INVALID_ARG;
    if (d->mask_mode == 1 && (!view_ok(d->mask, E) || d->mask.H != d->g.H || d->mask.W != d->g.W)) return MFC_ERR_INVALID_ARG;
    if (d->mask_mode == 3 && (!mfc_is16(d->dtype) || !d->mask.ptr || d->mask.Cp % 8 || d->mask.c_off % 8 || d->mask.H != d->g.H || d->mask.W != d->g.W)) return MFC_ERR_INVALID_ARG;
    const int Cg = d->C / E;
    const long total = (long)d->N * d->dst.H * d->dst.W * Cg;
    if (total >= (1L << 31) - 2048 || (long)d->N * d->g.H * d->g.W * Cg >= (1L << 31) - 2048) return MFC_ERR_UNSUPPORTED;
    const int blocks = (int)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (d->dst.H == d->g.H && d->dst.W == d->g.W) {
        const int b4 = (int)((total + 1023) / 1024);
        EW_PROF(st, "mask_add_same_kernel", d->dtype, (double)total * 16.0 * (2 + (d->mask_mode == 1 ? 1 : 0) + (d->accumulate ? 1 : 0)) + (d->mask_mode == 3 ? (double)total : 0.0));
        MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(mask_add_same_kernel<T_>, dim3(b4), dim3(256), 0, st, *d, total, Cg));
        MFC_PROF_END(st);
        MFC_CHECK_LAUNCH();
        return MFC_OK;
    }
    if (d->g.H < d->dst.H || d->g.W < d->dst.W) return MFC_ERR_UNSUPPORTED;      // the adjoint of an UP-sampling only
    if (d->scratch) {          // separable two-pass form
        const long t1 = (long)d->N * d->g.H * d->dst.W * Cg;
        if (t1 >= (1L << 31) - 2048) return MFC_ERR_UNSUPPORTED;
        const int b1 = (int)((t1 + 255) / 256), b2 = (int)((total + 255) / 256);
        // (both passes in one row: the pair is one up-sampling adjoint; bytes = the high-resolution gradient (+ mask) read once + the result)
        const double hi = (double)d->N * d->g.H * d->g.W * Cg * 16.0;
        EW_PROF(st, "mask_add_wpass+hpass_kernel", d->dtype, hi * (1 + (d->mask_mode == 1 ? 1 : 0)) + (d->mask_mode == 3 ? hi / 16 : 0) + (double)total * 16.0 * (d->accumulate ? 2 : 1));
        MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(mask_add_wpass_kernel<T_>, dim3(b1), dim3(256), 0, st, *d, (unsigned)t1, Cg);
                  hipLaunchKernelGGL(mask_add_hpass_kernel<T_>, dim3(b2), dim3(256), 0, st, *d, (unsigned)total, Cg));
        MFC_PROF_END(st);
        MFC_CHECK_LAUNCH();
        return MFC_OK;
    }
    const int ratio = (d->g.H + d->dst.H - 1) / d->dst.H;
    int LS = ratio >= 4 ? 3 : ratio >= 2 ? 2 : 1;                               // column lanes per output granule = 2^LS
    while (LS > 0 && (Cg << LS) > 256) --LS;
    if (Cg > 256) return MFC_ERR_UNSUPPORTED;
    const int PB = 256 / (Cg << LS);
    const long npo = (long)d->N * d->dst.H * d->dst.W;
    const int blocksS = (int)((npo + PB - 1) / PB);
    (void)blocks;
    MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(mask_add_kernel<T_>, dim3(blocksS), dim3(256), 0, st, *d, npo, Cg, LS, PB));
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// ------------------------------------------------------------------ bias gradient
// db[c] += sum over pixels of dy[pix][c]   (db must be zero at step start)
// SLICES: no atomics -- workgroup x writes its partial sums to db[x * Cp + c] (every channel, zeros included); the caller adds the slices up
// in a fixed order (mfc_unpack_wgrad), so the result does not depend on the order the workgroups finish in
template <typename T, bool SLICES>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* dy, float* db, long npix, int Cp, int C, int Cg, int PPI, int pix_per_block) {
    constexpr int E = Gran<T>::E;
    constexpr int U = 4;                       // granules in flight per thread
    __shared__ float red[256 * 8];
    const int gi = threadIdx.x % Cg, prow = threadIdx.x / Cg;
    const int ch = (blockIdx.y * Cg + gi) * E;
    const bool live = ch < Cp;
    const long p0 = (long)blockIdx.x * pix_per_block;
    long p1 = p0 + pix_per_block; if (p1 > npix) p1 = npix;
    float s[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = 0.f;
    const int chc = live ? ch : 0;
    for (long pp = p0 + prow; pp < p1; pp += (long)U * PPI) {
        uint4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long q = pp + (long)u * PPI;
            r[u] = *(const uint4*)(dy + (q < p1 ? q : p0 + prow) * Cp + chc);       // clamped: branch-free loads
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pp + (long)u * PPI < p1) {
                float v[E];
                Gran<T>::unpack(r[u], v);
#pragma unroll
                for (int e = 0; e < E; ++e) s[e] += v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) red[threadIdx.x * E + e] = s[e];
    __syncthreads();
    if (prow == 0 && live) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float a = 0.f;
            for (int q = 0; q < PPI; ++q) a += red[(q * Cg + gi) * E + e];
            if (SLICES) db[(size_t)blockIdx.x * Cp + ch + e] = (ch + e < C) ? a : 0.f;
            else if (ch + e < C) atomicAdd(db + ch + e, a);
        }
    }
}

extern "C" int mfc_bias_grad(const void* dy, float* db, int32_t dtype, int64_t npix, int32_t Cp, int32_t C, void* stream) {
    if (!dy || !db || npix <= 0 || Cp <= 0 || C > Cp || Cp % 8) return MFC_ERR_INVALID_ARG;
    if (!mfc_dtype_ok(dtype)) return MFC_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int E = mfc_is16(dtype) ? 8 : 4;
    const int Cgt = Cp / E;                                // granules per pixel
    // channel slabs of <= 16 granules: many pixel rows per workgroup, so few workgroups (= few same-address atomics) suffice
    const int nslab = (Cgt + 15) / 16;
    const int Cg = (Cgt + nslab - 1) / nslab;
    const int PPI = 256 / Cg;
    long want = npix / (16 * PPI) + 1; if (want > 1024 / nslab + 1) want = 1024 / nslab + 1;
    int ppb = (int)((npix + want - 1) / want);
    ppb = ((ppb + PPI - 1) / PPI) * PPI;
    const int blocks = (int)((npix + ppb - 1) / ppb);
    EW_PROF(st, "bias_grad_kernel", dtype, (double)npix * Cp * (E == 8 ? 2 : 4));
    MFC_TYPED(dtype, T_, hipLaunchKernelGGL((bias_grad_kernel<T_, false>), dim3(blocks, nslab), dim3(Cg * PPI), 0, st, (const T_*)dy, db, (long)npix, Cp, C, Cg, PPI, ppb));
    MFC_PROF_END(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

extern "C" int mfc_bias_grad_slices(const void* dy, float* slices, int32_t dtype, int64_t npix, int32_t Cp, int32_t C, int32_t nparts, void* stream) {
    if (!dy || !slices || npix <= 0 || Cp <= 0 || C > Cp || Cp % 8 || nparts <= 0 || nparts > 4096) return MFC_ERR_INVALID_ARG;
    if (!mfc_dtype_ok(dtype)) return MFC_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int E = mfc_is16(dtype) ? 8 : 4;
    const int Cgt = Cp / E;
    const int nslab = (Cgt + 15) / 16;
    const int Cg = (Cgt + nslab - 1) / nslab;
    const int PPI = 256 / Cg;
    int ppb = (int)((npix + nparts - 1) / nparts);
    ppb = ((ppb + PPI - 1) / PPI) * PPI;            // (workgroups past the last pixel write zero slices)
    EW_PROF(st, "bias_grad_kernel", dtype, (double)npix * Cp * (E == 8 ? 2 : 4));
    MFC_TYPED(dtype, T_, hipLaunchKernelGGL((bias_grad_kernel<T_, true>), dim3(nparts, nslab), dim3(Cg * PPI), 0, st, (const T_*)dy, slices, (long)npix, Cp, C, Cg, PPI, ppb));
    MFC_PROF_END(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// ------------------------------------------------------------------ layout bridges
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* src, T* dst, int N, int C, int H, int W, int Cp, int c_off, long total) {
    constexpr int E = Gran<T>::E;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;      // one pixel per thread
    if (idx >= total) return;
    const long HW = (long)H * W;
    const int n = (int)(idx / HW); const long r = idx - (long)n * HW;
    const int ng = (C + E - 1) / E;
    for (int g = 0; g < ng; ++g) {
        float f[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int c = g * E + e;
            f[e] = c < C ? src[((size_t)n * C + c) * HW + r] : 0.f;
        }
        mfc_st16(((char*)dst + ((size_t)idx * Cp + c_off + g * E) * sizeof(T)), Gran<T>::pack(f));
    }
}

extern "C" int mfc_nchw_to_nhwc(const float* src, void* dst, int32_t dtype, int32_t N, int32_t C, int32_t H, int32_t W,
                                int32_t Cp, int32_t c_off, int32_t zero_pad, void* stream) {
    (void)zero_pad;
    if (!src || !dst || N <= 0 || C <= 0 || Cp % 8 || !mfc_dtype_ok(dtype)) return MFC_ERR_INVALID_ARG;
    const int E = mfc_is16(dtype) ? 8 : 4;
    if (c_off % E || c_off + ((C + E - 1) / E) * E > Cp) return MFC_ERR_INVALID_ARG;
    const long total = (long)N * H * W;
    const int blocks = (int)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    MFC_TYPED(dtype, T_, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T_>, dim3(blocks), dim3(256), 0, st, src, (T_*)dst, N, C, H, W, Cp, c_off, total));
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* src, float* dst, int C, long HW, int Cp, long total) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;      // one output element per thread
    if (idx >= total) return;
    const long r = idx % HW; long t = idx / HW;
    const int c = (int)(t % C); const long n = t / C;
    dst[idx] = ld_elem<T>(src + (n * HW + r) * Cp + c);
}

extern "C" int mfc_nhwc_to_nchw(const void* src, float* dst, int32_t dtype, int32_t N, int32_t C, int32_t H, int32_t W,
                                int32_t Cp, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || C > Cp || !mfc_dtype_ok(dtype)) return MFC_ERR_INVALID_ARG;
    const long HW = (long)H * W, total = (long)N * C * HW;
    const int blocks = (int)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    MFC_TYPED(dtype, T_, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T_>, dim3(blocks), dim3(256), 0, st, (const T_*)src, dst, C, HW, Cp, total));
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// ------------------------------------------------------------------ head gather
// up4(logits)[h, w, k] for one frame: bilinear from the low-res logit map (hrnet.py:473-474)
template <typename T>
__device__ inline float up_logit(const T* L, int Hs, int Ws, int Lp, int H, int W, int h, int w, int k) {
    int h0, h1, w0, w1; float lh, lw;
    bilin_src(h, Hs, H, h0, h1, lh);
    bilin_src(w, Ws, W, w0, w1, lw);
    const float a = ld_elem<T>(L + ((size_t)h0 * Ws + w0) * Lp + k), b = ld_elem<T>(L + ((size_t)h0 * Ws + w1) * Lp + k);
    const float c = ld_elem<T>(L + ((size_t)h1 * Ws + w0) * Lp + k), e = ld_elem<T>(L + ((size_t)h1 * Ws + w1) * Lp + k);
    return (1.f - lh) * ((1.f - lw) * a + lw * b) + lh * ((1.f - lw) * c + lw * e);
}

// grid_sample(bilinear, zeros, align_corners=True) sample position of output pixel (h, w) under the
// MultiFrameNetBasic warp: grid normalised for 576x720 and cropped (multiframe_model.py:156-167,179-182)
__device__ inline void warp_pos(int h, int w, int H, int W, float fx, float fy, float& x, float& y) {
    const float gx = 2.0f * (float)w / 719.0f - 1.0f + fx / ((float)(W - 1) / 2.0f);
    const float gy = 2.0f * (float)h / 575.0f - 1.0f + fy / ((float)(H - 1) / 2.0f);
    x = (gx + 1.f) * 0.5f * (float)(W - 1);
    y = (gy + 1.f) * 0.5f * (float)(H - 1);
}

template <typename T>
__global__ __launch_bounds__(256) void head_gather_fwd_kernel(mfc_headgather_desc d, long total) {
    constexpr int E = Gran<T>::E;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;      // one output pixel per thread
    if (idx >= total) return;
    const long HW = (long)d.H * d.W;
    const int b = (int)(idx / HW); const long r = idx - (long)b * HW;
    const int h = (int)(r / d.W), w = (int)(r - (long)h * d.W);
    float v[40];
    const int Cp = d.Cp;
    for (int c = 0; c < Cp; ++c) v[c] = 0.f;
    int c = 0;
    for (int t = 0; t < d.T; ++t) {
        const T* L = (const T*)d.logits + (size_t)(t * d.B + b) * d.Hs * d.Ws * d.Lp;
        if (d.warp && t > 0) {
            const float fx = d.flow[t - 1][((size_t)b * 2 + 0) * HW + r], fy = d.flow[t - 1][((size_t)b * 2 + 1) * HW + r];
            float x, y; warp_pos(h, w, d.H, d.W, fx, fy, x, y);
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = (int)xf, y0 = (int)yf;
            const float ax = x - xf, ay = y - yf;
            for (int k = 0; k < d.nc; ++k) {
                float s = 0.f;
                for (int dy_ = 0; dy_ < 2; ++dy_)
                    for (int dx_ = 0; dx_ < 2; ++dx_) {
                        const int yy = y0 + dy_, xx = x0 + dx_;
                        if (yy < 0 || yy >= d.H || xx < 0 || xx >= d.W) continue;
                        const float wt = (dy_ ? ay : 1.f - ay) * (dx_ ? ax : 1.f - ax);
                        s += wt * up_logit<T>(L, d.Hs, d.Ws, d.Lp, d.H, d.W, yy, xx, k);
                    }
                v[c++] = s;
            }
        } else {
            for (int k = 0; k < d.nc; ++k) v[c++] = up_logit<T>(L, d.Hs, d.Ws, d.Lp, d.H, d.W, h, w, k);
        }
    }
    if (!d.warp && d.flow[0]) {
        for (int t = 0; t < d.T - 1; ++t) {
            v[c++] = d.flow[t][((size_t)b * 2 + 0) * HW + r];
            v[c++] = d.flow[t][((size_t)b * 2 + 1) * HW + r];
        }
    }
    if (d.depth[0]) {
        for (int t = 0; t < d.T; ++t) {
            if (d.warp && t > 0) {
                const float fx = d.flow[t - 1][((size_t)b * 2 + 0) * HW + r], fy = d.flow[t - 1][((size_t)b * 2 + 1) * HW + r];
                float x, y; warp_pos(h, w, d.H, d.W, fx, fy, x, y);
                const float xf = floorf(x), yf = floorf(y);
                const int x0 = (int)xf, y0 = (int)yf;
                const float ax = x - xf, ay = y - yf;
                float s = 0.f;
                for (int dy_ = 0; dy_ < 2; ++dy_)
                    for (int dx_ = 0; dx_ < 2; ++dx_) {
                        const int yy = y0 + dy_, xx = x0 + dx_;
                        if (yy < 0 || yy >= d.H || xx < 0 || xx >= d.W) continue;
                        s += (dy_ ? ay : 1.f - ay) * (dx_ ? ax : 1.f - ax) * d.depth[t][(size_t)b * HW + (size_t)yy * d.W + xx];
                    }
                v[c++] = s;
            } else {
                v[c++] = d.depth[t][(size_t)b * HW + r];
            }
        }
    }
    for (int g = 0; g < Cp / E; ++g)
        mfc_st16(((char*)d.xh + ((size_t)idx * Cp + g * E) * sizeof(T)), Gran<T>::pack(v + g * E));
}

static int head_check(const mfc_headgather_desc* d) {
    if (!d || !d->logits || !d->xh || (!mfc_dtype_ok(d->dtype))) return MFC_ERR_INVALID_ARG;
    if (d->T < 1 || d->T > 8 || d->nc < 1 || d->nc > d->Lp || d->Cp % 8 || d->Cp > 40 || d->Lp % 8) return MFC_ERR_INVALID_ARG;
    int c = d->T * d->nc + ((!d->warp && d->flow[0]) ? 2 * (d->T - 1) : 0) + (d->depth[0] ? d->T : 0);
    if (c > d->Cp) return MFC_ERR_INVALID_ARG;
    if (d->warp && (!d->flow[0] || d->T > 7)) return MFC_ERR_INVALID_ARG;      // depth[7] carries the warp-backward scratch
    if (d->warp && (d->H > 576 || d->W > 720)) return MFC_ERR_UNSUPPORTED;   // reference raises here too (SURVEY A7 ii)
    return MFC_OK;
}

extern "C" int mfc_head_gather_fwd(const mfc_headgather_desc* d, void* stream) {
    int rc = head_check(d); if (rc < 0) return rc;
    const long total = (long)d->B * d->H * d->W;
    const int blocks = (int)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    const int hesz = mfc_is16(d->dtype) ? 2 : 4;
    EW_PROF(st, "head_gather_fwd_kernel", d->dtype, (double)total * d->Cp * hesz + (double)d->T * d->B * d->Hs * d->Ws * d->Lp * hesz
            + (double)total * 4.0 * ((d->flow[0] ? 2 * (d->T - 1) : 0) + (d->depth[0] ? d->T : 0)));
    MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(head_gather_fwd_kernel<T_>, dim3(blocks), dim3(256), 0, st, *d, total));
    MFC_PROF_END(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// adjoint wrt the low-res logits.  Without warp: one gather per low-res pixel.  With the MultiFrameNetBasic warp the
// sample positions are data dependent, so the adjoint of grid_sample is a scatter: stage 1 scatters d(xh) of the warped
// frames (t >= 1) into an fp32 full-resolution scratch dU[B,H,W,(T-1)*nc] with atomics (4 taps per pixel and channel),
// stage 2 is the same x4 bilinear-adjoint gather, reading dU for t >= 1.
template <typename T>
__global__ __launch_bounds__(256) void head_warp_scatter_kernel(mfc_headgather_desc d, float* dU, long total) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;      // one full-resolution pixel per thread
    if (idx >= total) return;
    const long HW = (long)d.H * d.W;
    const int b = (int)(idx / HW); const long r = idx - (long)b * HW;
    const int h = (int)(r / d.W), w = (int)(r - (long)h * d.W);
    const int CU = (d.T - 1) * d.nc;
    const T* g = (const T*)d.xh + (size_t)idx * d.Cp;
    for (int t = 1; t < d.T; ++t) {
        const float fx = d.flow[t - 1][((size_t)b * 2 + 0) * HW + r], fy = d.flow[t - 1][((size_t)b * 2 + 1) * HW + r];
        float x, y; warp_pos(h, w, d.H, d.W, fx, fy, x, y);
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf;
        const float ax = x - xf, ay = y - yf;
        for (int dy_ = 0; dy_ < 2; ++dy_)
            for (int dx_ = 0; dx_ < 2; ++dx_) {
                const int yy = y0 + dy_, xx = x0 + dx_;
                if (yy < 0 || yy >= d.H || xx < 0 || xx >= d.W) continue;
                const float wt = (dy_ ? ay : 1.f - ay) * (dx_ ? ax : 1.f - ax);
                float* o = dU + (((size_t)b * d.H + yy) * d.W + xx) * CU + (t - 1) * d.nc;
                for (int k = 0; k < d.nc; ++k) atomicAdd(o + k, wt * ld_elem<T>(g + t * d.nc + k));
            }
    }
}

// one thread per low-resolution pixel of one clip b: accumulates the adjoint for ALL T frames at once (the T*nc gradient
// channels of a full-resolution pixel are contiguous, so a tap is Cp/E granule loads instead of T*nc scalar ones); the
// separable tap weights are computed once per row / column.
template <typename T>
__global__ __launch_bounds__(256) void head_gather_bwd_kernel(mfc_headgather_desc d, T* dl, const float* dU, long total) {
    constexpr int E = Gran<T>::E;
    constexpr int MAXC = 40;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;      // (b, hs, ws)
    if (idx >= total) return;
    const long HWs = (long)d.Hs * d.Ws;
    const int b = (int)(idx / HWs); const long r = idx - (long)b * HWs;
    const int hs = (int)(r / d.Ws), ws = (int)(r - (long)hs * d.Ws);
    const int TC = d.T * d.nc;                            // logit-gradient channels of xh (<= Cp <= 40)
    float acc[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) acc[k] = 0.f;
    int hlo, hhi, wlo, whi;
    adj_range(hs, d.Hs, d.H, hlo, hhi);
    adj_range(ws, d.Ws, d.W, wlo, whi);
    const int CU = (d.T - 1) * d.nc;
    const char* G = (const char*)d.xh + (size_t)b * d.H * d.W * d.Cp * sizeof(T);
    const float* GU = dU ? dU + (size_t)b * d.H * d.W * CU : nullptr;
    // granules that hold logit gradients (with the warp only frame 0 comes from xh; acc[MAXC-1-k] then holds dU channel k)
    const int ng = GU ? (d.nc + E - 1) / E : (TC + E - 1) / E;
    for (int ww0 = wlo; ww0 <= whi; ww0 += 8) {
        float wcol[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ww = ww0 + j;
            int w0, w1; float lw;
            bilin_src(ww <= whi ? ww : whi, d.Ws, d.W, w0, w1, lw);
            wcol[j] = ww <= whi ? ((w0 == ws ? 1.f - lw : 0.f) + (w1 == ws ? lw : 0.f)) : 0.f;
        }
        for (int hh = hlo; hh <= hhi; ++hh) {
            int h0, h1; float lh;
            bilin_src(hh, d.Hs, d.H, h0, h1, lh);
            const float wh = (h0 == hs ? 1.f - lh : 0.f) + (h1 == hs ? lh : 0.f);
            if (wh == 0.f) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wt = wh * wcol[j];
                if (wt == 0.f) continue;
                const size_t pix = (size_t)hh * d.W + (ww0 + j);
                const char* gp = G + pix * d.Cp * sizeof(T);
#pragma unroll
                for (int g = 0; g < MAXC / E; ++g) {
                    if (g < ng) {
                        float v[E];
                        Gran<T>::unpack(*(const uint4*)(gp + g * 16), v);
#pragma unroll
                        for (int e = 0; e < E; ++e) acc[g * E + e] += wt * v[e];
                    }
                }
                if (GU) {       // warped frames (t >= 1): their adjoint comes from the scattered fp32 scratch instead
                    const float* up = GU + pix * CU;
#pragma unroll
                    for (int k = 0; k < 30; ++k) if (k < CU) acc[MAXC - 1 - k] += wt * up[k];
                }
            }
        }
    }
    for (int t = 0; t < d.T; ++t) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = 0.f;
            if (k < d.nc) {
                // (dynamic register-array indexing is avoided: select with a compile-time unrolled scan)
                const int src = (GU && t > 0) ? (MAXC - 1 - ((t - 1) * d.nc + k)) : (t * d.nc + k);
#pragma unroll
                for (int q = 0; q < MAXC; ++q) v = (q == src) ? acc[q] : v;
            }
            o[k] = v;
        }
        T* op = dl + ((size_t)(t * d.B + b) * HWs + r) * d.Lp;
        for (int g = 0; g < d.Lp / E; ++g) {
            float z[E];
#pragma unroll
            for (int e = 0; e < E; ++e) z[e] = (g * E + e) < 8 ? o[(g * E + e) & 7] : 0.f;
            *(uint4*)(((char*)op + g * 16)) = Gran<T>::pack(z);
        }
    }
}

extern "C" int mfc_head_gather_bwd(const mfc_headgather_desc* d, void* dlogits, void* stream) {
    int rc = head_check(d); if (rc < 0) return rc;
    if (!dlogits || d->nc > 8) return MFC_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* dU = nullptr;
    if (d->warp && d->T > 1) {
        dU = (float*)d->depth[7];                  // fp32 scratch [B, H, W, (T-1)*nc] supplied by the caller (see header)
        if (!dU) return MFC_ERR_INVALID_ARG;
        const size_t bytes = (size_t)d->B * d->H * d->W * (d->T - 1) * d->nc * sizeof(float);
        if (hipMemsetAsync(dU, 0, bytes, st) != hipSuccess) return MFC_ERR_LAUNCH;
        const long tot = (long)d->B * d->H * d->W;
        const int blk = (int)((tot + 255) / 256);
        MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(head_warp_scatter_kernel<T_>, dim3(blk), dim3(256), 0, st, *d, dU, tot));
    }
    const long total = (long)d->B * d->Hs * d->Ws;
    const int blocks = (int)((total + 255) / 256);
    const int hesz = mfc_is16(d->dtype) ? 2 : 4;
    EW_PROF(st, "head_gather_bwd_kernel", d->dtype, (double)d->B * d->H * d->W * d->Cp * hesz + (double)d->T * total * d->Lp * hesz);
    MFC_TYPED(d->dtype, T_, hipLaunchKernelGGL(head_gather_bwd_kernel<T_>, dim3(blocks), dim3(256), 0, st, *d, (T_*)dlogits, (const float*)dU, total));
    MFC_PROF_END(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}


// ------------------------------------------------------------------ ResUnet_VB pieces (models/resunet.py), inference
// one workgroup per output channel: biased mean / variance of its cin*kh*kw weights in fp64, then the standardised weights
__global__ __launch_bounds__(256) void ws_normalize_kernel(const float* w, float* wo, int per_out, float eps) {
    __shared__ double rs[256], rq[256];
    const float* src = w + (size_t)blockIdx.x * per_out;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < per_out; i += 256) { const double v = (double)src[i]; s += v; q += v * v; }
    rs[threadIdx.x] = s; rq[threadIdx.x] = q;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) { rs[threadIdx.x] += rs[threadIdx.x + h]; rq[threadIdx.x] += rq[threadIdx.x + h]; }
        __syncthreads();
    }
    const double mean = rs[0] / per_out;
    double var = rq[0] / per_out - mean * mean; if (var < 0.0) var = 0.0;
    const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
    for (int i = threadIdx.x; i < per_out; i += 256) wo[(size_t)blockIdx.x * per_out + i] = (src[i] - m) * r;
}
extern "C" int mfc_ws_normalize(const float* w, float* w_out, int32_t Cout, int32_t per_out, float eps, void* stream) {
    if (!w || !w_out || Cout <= 0 || per_out <= 0) return MFC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(ws_normalize_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, w_out, per_out, eps);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// GroupNorm coefficients: grid (channel groups, N); a workgroup sums the replica rows of its group's channels, then writes scale / shift
__global__ __launch_bounds__(256) void gn_finalize_kernel(mfc_gnfin_desc d) {
    __shared__ double rs[256], rq[256];
    const int g = blockIdx.x, n = blockIdx.y, cpg = d.C / d.groups;
    const size_t rstride = (size_t)d.N * 2 * d.Cp;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < cpg * MFC_R; i += 256) {
        const int c = g * cpg + i % cpg, r = i / cpg;
        s += (double)d.stats[(size_t)r * rstride + ((size_t)n * 2 + 0) * d.Cp + c];
        q += (double)d.stats[(size_t)r * rstride + ((size_t)n * 2 + 1) * d.Cp + c];
    }
    rs[threadIdx.x] = s; rq[threadIdx.x] = q;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) { rs[threadIdx.x] += rs[threadIdx.x + h]; rq[threadIdx.x] += rq[threadIdx.x + h]; }
        __syncthreads();
    }
    const double cnt = (double)d.count * cpg;
    const double mean = rs[0] / cnt;
    double var = rq[0] / cnt - mean * mean; if (var < 0.0) var = 0.0;
    const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)d.eps));
    for (int i = threadIdx.x; i < cpg; i += 256) {
        const int c = g * cpg + i;
        float* cf = d.coef + (size_t)n * 4 * d.Cp + c;
        const float scale = d.gamma[c] * r;
        cf[0] = scale; cf[d.Cp] = d.beta[c] - m * scale; cf[2 * d.Cp] = m; cf[3 * d.Cp] = r;
    }
}
extern "C" int mfc_gn_finalize(const mfc_gnfin_desc* d, void* stream) {
    if (!d || !d->stats || !d->coef || !d->gamma || !d->beta || d->C <= 0 || d->C > d->Cp || d->N <= 0 || d->groups <= 0 || d->C % d->groups || d->count <= 0.f)
        return MFC_ERR_INVALID_ARG;
    if (d->N > 65535) return MFC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(d->groups, d->N), dim3(256), 0, (hipStream_t)stream, *d);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// nearest-neighbour x2: dst[n, h, w, :] = src[n, h / 2, w / 2, :]; one thread per destination granule
template <typename T>
__global__ __launch_bounds__(256) void upsample_nearest2x_kernel(const char* src, char* dst, int H, int W, int Cg, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % Cg); long pix = idx / Cg;
    const int w = (int)(pix % (2 * W)); pix /= (2 * W);
    const int h = (int)(pix % (2 * H)); const long n = pix / (2 * H);
    *(uint4*)((dst + idx * 16)) = *(const uint4*)(src + ((((size_t)n * H + (h >> 1)) * W + (w >> 1)) * Cg + g) * 16);
}
extern "C" int mfc_upsample_nearest2x(const void* src, void* dst, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t Cp, void* stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || Cp <= 0 || Cp % 8 || (!mfc_dtype_ok(dtype))) return MFC_ERR_INVALID_ARG;
    const int Cg = Cp / (mfc_is16(dtype) ? 8 : 4);
    const long total = (long)N * 2 * H * 2 * W * Cg;
    hipLaunchKernelGGL(upsample_nearest2x_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, (char*)dst, H, W, Cg, total);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}


// ------------------------------------------------------------------ ResUnet_VB backward pieces (models/resunet.py:46-76 under loss.backward())
// GroupNorm backward, finalize: from the per-(image, channel) sums S1 = sum g, S2 = sum g*yhat (mfc_bnbwd_reduce with images_per_group = 1,
// mask_mode 4 folds the SiLU derivative into g) to  A[n, group] = mean over the group's channels and pixels of gamma*g,  B = the same of
// gamma*g*yhat  (written per channel into bcoef [N][2][Cp] for mfc_bnbwd_apply with gn_mode), and  dgamma[c] = sum_n S2, dbeta[c] = sum_n S1.
// One workgroup walks the images one after the other (fixed order: deterministic).
__global__ __launch_bounds__(256) void gnbwd_finalize_kernel(mfc_gnbwdfin_desc d) {
    extern __shared__ float gsh[];                 // [2][C] gamma-weighted sums of the current image, then [2][groups]
    float* w1 = gsh; float* w2 = gsh + d.C; float* ga = gsh + 2 * d.C; float* gb = ga + d.groups;
    const int cpg = d.C / d.groups;
    const size_t rstride = (size_t)d.N * 2 * d.Cp;
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};          // channels tid, tid + 256, ... (C <= 1024)
    for (int n = 0; n < d.N; ++n) {
        for (int k = 0, c = threadIdx.x; c < d.C; c += 256, ++k) {
            const double s1 = replica_sum(d.bstats + ((size_t)n * 2 + 0) * d.Cp + c, rstride);
            const double s2 = replica_sum(d.bstats + ((size_t)n * 2 + 1) * d.Cp + c, rstride);
            db[k] += (float)s1; dg[k] += (float)s2;
            w1[c] = d.gamma[c] * (float)s1; w2[c] = d.gamma[c] * (float)s2;
        }
        __syncthreads();
        for (int g = threadIdx.x; g < d.groups; g += 256) {
            double a = 0.0, b = 0.0;
            for (int i = 0; i < cpg; ++i) { a += (double)w1[g * cpg + i]; b += (double)w2[g * cpg + i]; }
            ga[g] = (float)(a / ((double)d.count * cpg)); gb[g] = (float)(b / ((double)d.count * cpg));
        }
        __syncthreads();
        for (int c = threadIdx.x; c < d.Cp; c += 256) {
            d.bcoef[((size_t)n * 2 + 0) * d.Cp + c] = c < d.C ? ga[c / cpg] : 0.f;
            d.bcoef[((size_t)n * 2 + 1) * d.Cp + c] = c < d.C ? gb[c / cpg] : 0.f;
        }
        __syncthreads();
    }
    for (int k = 0, c = threadIdx.x; c < d.C; c += 256, ++k) { d.dgamma[c] = dg[k]; d.dbeta[c] = db[k]; }
}
extern "C" int mfc_gnbwd_finalize(const mfc_gnbwdfin_desc* d, void* stream) {
    if (!d || !d->bstats || !d->bcoef || !d->gamma || !d->dgamma || !d->dbeta || d->C <= 0 || d->C > d->Cp || d->N <= 0 || d->groups <= 0 ||
        d->C % d->groups || d->count <= 0.f) return MFC_ERR_INVALID_ARG;
    if (d->C > 1024) return MFC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gnbwd_finalize_kernel, dim3(1), dim3(256), (size_t)(2 * d->C + 2 * d->groups) * 4, (hipStream_t)stream, *d);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// weight standardisation backward (resunet.py:55-60): w' = (w - mean) * r, r = (var + eps)^-1/2 over the cin*kh*kw weights of an output
// channel  =>  dw = r * (dw' - mean(dw') - w' * mean(dw' * w')).  One workgroup per output channel; statistics in fp64 as in the forward.
__global__ __launch_bounds__(256) void ws_backward_kernel(const float* w, const float* dws, float* dw, int per_out, float eps) {
    __shared__ double ra[256], rb[256];
    const size_t base = (size_t)blockIdx.x * per_out;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < per_out; i += 256) { const double v = (double)w[base + i]; s += v; q += v * v; }
    ra[threadIdx.x] = s; rb[threadIdx.x] = q;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) { ra[threadIdx.x] += ra[threadIdx.x + h]; rb[threadIdx.x] += rb[threadIdx.x + h]; }
        __syncthreads();
    }
    const double mean = ra[0] / per_out;
    double var = rb[0] / per_out - mean * mean; if (var < 0.0) var = 0.0;
    const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
    __syncthreads();
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < per_out; i += 256) {
        const double g = (double)dws[base + i];
        a += g; b += g * (double)((w[base + i] - m) * r);
    }
    ra[threadIdx.x] = a; rb[threadIdx.x] = b;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) { ra[threadIdx.x] += ra[threadIdx.x + h]; rb[threadIdx.x] += rb[threadIdx.x + h]; }
        __syncthreads();
    }
    const float m1 = (float)(ra[0] / per_out), m2 = (float)(rb[0] / per_out);
    for (int i = threadIdx.x; i < per_out; i += 256) {
        const float wh = (w[base + i] - m) * r;
        dw[base + i] = r * (dws[base + i] - m1 - wh * m2);
    }
}
extern "C" int mfc_ws_backward(const float* w, const float* dws, float* dw, int32_t Cout, int32_t per_out, float eps, void* stream) {
    if (!w || !dws || !dw || Cout <= 0 || per_out <= 0) return MFC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(ws_backward_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, dws, dw, per_out, eps);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// adjoint of the nearest-neighbour x2 up-sampling: dst[n, h, w, :] (+)= sum of the four src pixels it was copied to (fp32 sum, one rounding)
template <typename T>
__global__ __launch_bounds__(256) void upsample_nearest2x_bwd_kernel(const char* src, char* dst, int H, int W, int Cg, long total, int accumulate) {
    constexpr int E = Gran<T>::E;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % Cg); long pix = idx / Cg;
    const int w = (int)(pix % W); pix /= W;
    const int h = (int)(pix % H); const long n = pix / H;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    if (accumulate) Gran<T>::unpack(*(const uint4*)(dst + idx * 16), acc);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float f[E];
            Gran<T>::unpack(*(const uint4*)(src + ((((size_t)n * 2 * H + 2 * h + a) * 2 * W + 2 * w + b) * Cg + g) * 16), f);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += f[e];
        }
    *(uint4*)((dst + idx * 16)) = Gran<T>::pack(acc);
}
extern "C" int mfc_upsample_nearest2x_bwd(const void* dsrc, void* ddst, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t Cp, int32_t accumulate, void* stream) {
    if (!dsrc || !ddst || N <= 0 || H <= 0 || W <= 0 || Cp <= 0 || Cp % 8 || (!mfc_dtype_ok(dtype))) return MFC_ERR_INVALID_ARG;
    const int Cg = Cp / (mfc_is16(dtype) ? 8 : 4);
    const long total = (long)N * H * W * Cg;
    MFC_TYPED(dtype, T_, hipLaunchKernelGGL(upsample_nearest2x_bwd_kernel<T_>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                            (const char*)dsrc, (char*)ddst, H, W, Cg, total, accumulate));
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}
