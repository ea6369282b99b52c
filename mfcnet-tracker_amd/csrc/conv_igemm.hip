// Implicit-GEMM convolution on MFMA for gfx950 (forward, and data-gradient by re-use).
//
// One workgroup (4 waves) computes a tile of <=128 output pixels (a TH x TW patch of one image)
// x NT*16 output channels.  The input halo patch of the tile is staged ONCE per channel chunk into
// LDS (so the k*k taps re-read LDS, not L2/HBM), with the producer's BatchNorm-apply + ReLU fused
// into the staging pass; packed weights are staged per tap row.  MFMA roles: A = weights
// (rows = cout), B = pixels (cols = pixel), so each lane ends up with 4 consecutive output
// channels of one pixel -> vector stores along the NHWC channel axis.
//   bf16: v_mfma_f32_16x16x32_bf16, one k-step = 4 granules (32 channels), fp32 accumulate
//   fp32: v_mfma_f32_16x16x4_f32,   one k-step = 1 granule  (4 channels), exact fp32
// Replaces: nn.Conv2d forward / cuDNN dgrad of models/hrnet.py:39-42,82-88,200-230,361-386,334-351
// and models/multiframe_model.py:191-201 (see include/mfcnet_hip.h).
#include "common.h"

struct ConvK {
    const char* in; const char* wp; char* out;
    const float* bias; const float* in_coef; float* out_stats;
    int N, Hin, Win, Cin_p, Cin_g;      // Cin_g: granules to reduce over
    int Hout, Wout, Cout_p, Cout;
    int Hl, Wl, TA, TB, dh0, dw0, s;
    int osh, osw, ooh, oow;
    int in_relu, ipg, G, accumulate;
    int TH, TW, tilesY, tilesX;
    int KG, nchunks, Kg_total, Np;      // chunk granules, #chunks, packed K granules, packed N
    int PH, PW, pitch;                  // patch dims (pixels) and pixel pitch (bytes)
    int Yblocks, nwg;
    int off_w, off_ktab;                // LDS offsets (bytes)
};

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvK p) {
    constexpr int E = Gran<T>::E;
    constexpr bool BF = (E == 8);
    constexpr int NT16 = NT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    char* wl = smem + p.off_w;
    int* ktab = (int*)(smem + p.off_ktab);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = xcd_remap(blockIdx.x, p.nwg);
    const int yb = L % p.Yblocks;
    int xt = L / p.Yblocks;
    const int tx_i = xt % p.tilesX; xt /= p.tilesX;
    const int ty_i = xt % p.tilesY;
    const int n = xt / p.tilesY;
    const int n0 = yb * NT16;
    const int i0 = ty_i * p.TH, j0 = tx_i * p.TW;
    const int ih0 = i0 * p.s + p.dh0, iw0 = j0 * p.s + p.dw0;
    const int grp = n / p.ipg;

    // this lane's two pixels (one per m-tile)
    int pbase[2]; bool pvalid[2]; int pty[2], ptx[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int pp = (wave * 2 + mt) * 16 + (lane & 15);
        int ty = pp / p.TW, tx = pp - ty * p.TW;
        bool v = (pp < p.TH * p.TW) && (i0 + ty < p.Hl) && (j0 + tx < p.Wl);
        if (!v) { ty = 0; tx = 0; }
        pty[mt] = ty; ptx[mt] = tx; pvalid[mt] = v;
        pbase[mt] = ((ty * p.s) * p.PW + tx * p.s) * p.pitch + (BF ? 0 : (lane >> 4) * 4);
    }

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int npix = p.PH * p.PW;
    const size_t in_img = (size_t)n * p.Hin * p.Win;

    for (int c = 0; c < p.nchunks; ++c) {
        const int g0 = c * p.KG;
        const int kg = min(p.KG, p.Cin_g - g0);
        int KGP = 1; while (KGP < kg) KGP <<= 1;
        const int nslots = BF ? ((p.TB * kg + 3) & ~3) : p.TB * kg;
        __syncthreads();                      // previous chunk fully consumed
        if (tid < nslots) {                   // slot -> patch offset (tail slots repeat the last valid one)
            int q = min(tid, p.TB * kg - 1);
            int b = q / kg, gi = q - b * kg;
            ktab[tid] = b * p.pitch + gi * 16;
        }
        {   // ---- stage the input patch, fused BN-apply + ReLU, zero padding ----
            const int gi = tid & (KGP - 1);
            const int pstep = 256 / KGP;
            float sc[E], sh[E];
            const bool xf = (p.in_coef != nullptr);
            if (xf && gi < kg) {
                const float* cf = p.in_coef + (size_t)grp * 4 * p.Cin_p + (g0 + gi) * E;
#pragma unroll
                for (int e = 0; e < E; ++e) { sc[e] = cf[e]; sh[e] = cf[p.Cin_p + e]; }
            }
            if (gi < kg) {
                for (int pix = tid / KGP; pix < npix; pix += pstep) {
                    int py = pix / p.PW, px = pix - py * p.PW;
                    int ih = ih0 + py, iw = iw0 + px;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Win) {
                        v = *(const uint4*)(p.in + ((in_img + (size_t)ih * p.Win + iw) * p.Cin_p) * sizeof(T) + (size_t)(g0 + gi) * 16);
                        if (xf) {
                            float f[E];
                            Gran<T>::unpack(v, f);
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                float t = f[e] * sc[e] + sh[e];
                                f[e] = p.in_relu ? fmaxf(t, 0.f) : t;
                            }
                            v = Gran<T>::pack(f);
                        }
                    }
                    *(uint4*)(patch + pix * p.pitch + gi * 16) = v;
                }
            }
        }
        for (int a = 0; a < p.TA; ++a) {
            if (a > 0) __syncthreads();       // previous tap row's weights consumed
            // ---- stage packed weights of tap row a for this chunk: [slot][NT16][16B] ----
            for (int b = 0; b < p.TB; ++b) {
                const char* src = p.wp + ((size_t)((a * p.TB + b) * p.Kg_total + g0) * p.Np + n0) * 16;
                for (int idx = tid; idx < kg * NT16; idx += 256) {
                    int gi = idx / NT16, nn = idx - gi * NT16;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (n0 + nn < p.Np) v = *(const uint4*)(src + ((size_t)gi * p.Np + nn) * 16);
                    *(uint4*)(wl + ((b * kg + gi) * NT16 + nn) * 16) = v;
                }
            }
            for (int idx = p.TB * kg * NT16 + tid; idx < nslots * NT16; idx += 256)
                *(uint4*)(wl + idx * 16) = make_uint4(0, 0, 0, 0);
            __syncthreads();
            const int arow = a * p.PW * p.pitch;
            if constexpr (BF) {
                const int nk = nslots >> 2;
                for (int ks = 0; ks < nk; ++ks) {
                    const int q = ks * 4 + (lane >> 4);
                    const int ko = ktab[q];
                    bf16x8 xf0 = *(const bf16x8*)(patch + pbase[0] + arow + ko);
                    bf16x8 xf1 = *(const bf16x8*)(patch + pbase[1] + arow + ko);
                    const char* wq = wl + (q * NT16 + (lane & 15)) * 16;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        bf16x8 wf = *(const bf16x8*)(wq + nt * 256);
                        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf0, acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf1, acc[1][nt], 0, 0, 0);
                    }
                }
            } else {
                for (int q = 0; q < nslots; ++q) {
                    const int ko = ktab[q];
                    float x0 = *(const float*)(patch + pbase[0] + arow + ko);
                    float x1 = *(const float*)(patch + pbase[1] + arow + ko);
                    const char* wq = wl + (q * NT16 + (lane & 15)) * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float wf = *(const float*)(wq + nt * 256);
                        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, x0, acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, x1, acc[1][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---------------- epilogue: bias, accumulate, store, statistics ----------------
    const int cq = (lane >> 4) * 4;
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int oi = (i0 + pty[mt]) * p.osh + p.ooh, oj = (j0 + ptx[mt]) * p.osw + p.oow;
        char* orow = p.out + (((size_t)n * p.Hout + oi) * p.Wout + oj) * p.Cout_p * sizeof(T);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = n0 + nt * 16 + cq;
            if (co >= p.Cout_p) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[mt][nt][r];
                if (p.bias && co + r < p.Cout) v[r] += p.bias[co + r];
            }
            if (pvalid[mt]) {
                T* o = (T*)orow + co;
                if (p.accumulate) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += ld_elem<T>(o + r);
                }
                if constexpr (BF) {
                    uint2 pk = make_uint2(bf16_bits(v[0]) | (bf16_bits(v[1]) << 16), bf16_bits(v[2]) | (bf16_bits(v[3]) << 16));
                    *(uint2*)o = pk;
                } else {
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[nt][r] += v[r]; ssq[nt][r] += v[r] * v[r]; }
            }
        }
    }
    if (p.out_stats) {
        __syncthreads();                      // LDS free for reuse
        float* red = (float*)smem;            // [4 waves][2][NT16]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = wave16_sum(ssum[nt][r]);
                float b = wave16_sum(ssq[nt][r]);
                if ((lane & 15) == 0) {
                    red[(wave * 2 + 0) * NT16 + nt * 16 + cq + r] = a;
                    red[(wave * 2 + 1) * NT16 + nt * 16 + cq + r] = b;
                }
            }
        __syncthreads();
        if (tid < 2 * NT16) {
            const int which = tid / NT16, cl = tid - which * NT16;
            if (n0 + cl < p.Cout_p) {
                float s = red[(0 * 2 + which) * NT16 + cl] + red[(1 * 2 + which) * NT16 + cl] +
                          red[(2 * 2 + which) * NT16 + cl] + red[(3 * 2 + which) * NT16 + cl];
                const int rep = (blockIdx.x / p.Yblocks) % MFC_R;
                atomicAdd(p.out_stats + (((size_t)rep * p.G + grp) * 2 + which) * p.Cout_p + n0 + cl, s);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
static void choose_tile(int Hl, int Wl, int& TH, int& TW) {
    // maximise useful pixels per 128-pixel tile; prefer wide tiles (contiguous NHWC rows)
    double best = -1; int bh = 8, bw = 16;
    for (int tw = 1; tw <= 128 && tw <= Wl + 15; ++tw) {
        int th = 128 / tw; if (th > Hl) th = Hl; if (th < 1) continue;
        if (tw > Wl) continue;
        double tiles = (double)ceil_div(Hl, th) * ceil_div(Wl, tw);
        double eff = (double)Hl * Wl / (tiles * 128.0);
        double halo = (double)(th * tw) / ((th + 2.0) * (tw + 2.0));   // mild preference for square-ish
        double score = eff + 0.05 * halo + ((tw % 16 == 0) ? 0.01 : 0.0);
        if (score > best) { best = score; bh = th; bw = tw; }
    }
    TH = bh; TW = bw;
}

static int conv_setup(const mfc_conv_desc* d, ConvK& k, int& NT, size_t& lds) {
    if (!d || !d->in || !d->wp || !d->out) return MFC_ERR_INVALID_ARG;
    if (d->dtype != MFC_F32 && d->dtype != MFC_BF16) return MFC_ERR_INVALID_ARG;
    const int E = d->dtype == MFC_BF16 ? 8 : 4;
    if (d->Cin_p % 8 || d->Cout_p % 8 || d->Cin > d->Cin_p || d->Cout > d->Cout_p) return MFC_ERR_INVALID_ARG;
    if (d->N <= 0 || d->Hl <= 0 || d->Wl <= 0 || d->TA <= 0 || d->TB <= 0 || d->in_stride < 1) return MFC_ERR_INVALID_ARG;
    if ((d->Hl - 1) * d->out_sh + d->out_oh >= d->Hout || (d->Wl - 1) * d->out_sw + d->out_ow >= d->Wout) return MFC_ERR_INVALID_ARG;
    if (d->images_per_group <= 0 || d->N % d->images_per_group) return MFC_ERR_INVALID_ARG;
    k.in = (const char*)d->in; k.wp = (const char*)d->wp; k.out = (char*)d->out;
    k.bias = d->bias; k.in_coef = d->in_coef; k.out_stats = d->out_stats;
    k.N = d->N; k.Hin = d->Hin; k.Win = d->Win; k.Cin_p = d->Cin_p; k.Cin_g = ceil_div(d->Cin, E);
    k.Hout = d->Hout; k.Wout = d->Wout; k.Cout_p = d->Cout_p; k.Cout = d->Cout;
    k.Hl = d->Hl; k.Wl = d->Wl; k.TA = d->TA; k.TB = d->TB; k.dh0 = d->dh0; k.dw0 = d->dw0; k.s = d->in_stride;
    k.osh = d->out_sh; k.osw = d->out_sw; k.ooh = d->out_oh; k.oow = d->out_ow;
    k.in_relu = d->in_relu; k.ipg = d->images_per_group; k.G = d->N / d->images_per_group; k.accumulate = d->accumulate;
    k.TH = d->TH; k.TW = d->TW;
    if (k.TH <= 0 || k.TW <= 0) choose_tile(d->Hl, d->Wl, k.TH, k.TW);
    if (k.TH * k.TW > 128) return MFC_ERR_INVALID_ARG;
    k.tilesY = ceil_div(d->Hl, k.TH); k.tilesX = ceil_div(d->Wl, k.TW);
    k.Kg_total = ceil_div(d->Cin, E);              // (packer and kernel agree on ceil(Cin/E))
    k.Np = ceil_div(d->Cout, 16) * 16;
    const int n16 = k.Np / 16;
    // N tile: fewest computed n-tiles, mild preference for wide tiles (more reuse of the staged patch)
    {
        const int cand[5] = {6, 4, 3, 2, 1};
        double bestc = 1e30; NT = 1;
        for (int i = 0; i < 5; ++i) {
            double c = (double)ceil_div(n16, cand[i]) * cand[i] * (1.0 + 0.5 / cand[i]);
            if (c < bestc - 1e-9) { bestc = c; NT = cand[i]; }
        }
    }
    k.Yblocks = ceil_div(n16, NT);
    k.PH = (k.TH - 1) * k.s + k.TA; k.PW = (k.TW - 1) * k.s + k.TB;
    // channel chunk: balanced chunks of <= 8 granules, shrunk until the LDS budget holds
    int kgmax = 8;
    for (;;) {
        int nch = ceil_div(k.Cin_g, kgmax);
        k.KG = ceil_div(k.Cin_g, nch); k.nchunks = ceil_div(k.Cin_g, k.KG);
        k.pitch = k.KG * 16 + 16;
        int nslots = (E == 8) ? ((k.TB * k.KG + 3) & ~3) : k.TB * k.KG;
        size_t patch = (size_t)k.PH * k.PW * k.pitch;
        size_t wbytes = (size_t)nslots * NT * 16 * 16;
        k.off_w = (int)((patch + 15) & ~(size_t)15);
        k.off_ktab = k.off_w + (int)wbytes;
        lds = k.off_ktab + (size_t)nslots * 4 + 64;
        size_t red = (size_t)4 * 2 * NT * 16 * 4;
        if (lds < red) lds = red;
        if (lds <= 64 * 1024 || kgmax == 1) break;
        kgmax = (kgmax > 2) ? kgmax / 2 : 1;
    }
    if (lds > 160 * 1024) return MFC_ERR_UNSUPPORTED;
    if (k.TB * k.KG > 256) return MFC_ERR_UNSUPPORTED;       // ktab is filled by one pass of 256 threads
    k.nwg = k.N * k.tilesY * k.tilesX * k.Yblocks;
    return MFC_OK;
}

template <typename T, int NT>
static int conv_launch(const ConvK& k, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on) {
        const int slot = NT == 1 ? 0 : NT == 2 ? 1 : NT == 3 ? 2 : NT == 4 ? 3 : 4;
        const double E = sizeof(T) == 2 ? 8.0 : 4.0;
        const double flops = 2.0 * k.N * k.Hl * k.Wl * (double)k.Cout * k.TA * k.TB * (k.Cin_g * E);
        const double bytes = ((double)k.N * k.Hin * k.Win * k.Cin_g * 16.0) / (k.osh * k.osw) + (double)k.N * k.Hl * k.Wl * k.Cout_p * sizeof(T);
        mfc_prof_before(st, 0 * 16 + (sizeof(T) == 2 ? 8 : 0) + slot, flops, bytes);
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, NT>), dim3(k.nwg), dim3(256), lds, st, k);
    if (g_mfc_prof_on) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

extern "C" int mfc_conv2d_lds_bytes(const mfc_conv_desc* d) {
    ConvK k; int NT; size_t lds;
    int rc = conv_setup(d, k, NT, lds);
    return rc < 0 ? rc : (int)lds;
}

extern "C" int mfc_conv2d_fwd(const mfc_conv_desc* d, void* stream) {
    ConvK k; int NT; size_t lds;
    int rc = conv_setup(d, k, NT, lds);
    if (rc < 0) return rc;
    hipStream_t st = (hipStream_t)stream;
#define MFC_CONV_CASE(nt) \
    case nt: return d->dtype == MFC_BF16 ? conv_launch<bf16_t, nt>(k, lds, st) : conv_launch<float, nt>(k, lds, st);
    switch (NT) {
        MFC_CONV_CASE(1) MFC_CONV_CASE(2) MFC_CONV_CASE(3) MFC_CONV_CASE(4) MFC_CONV_CASE(6)
    }
#undef MFC_CONV_CASE
    return MFC_ERR_UNSUPPORTED;
}
