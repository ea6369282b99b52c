// Weight-gradient convolution on MFMA for gfx950.
//   dWp[a,b][co][ci] += sum_{n,i,j} dy[n,i,j,co] * f(x[n, i*s+dh0+a, j*s+dw0+b, ci])
// GEMM view: M = cout, N = cin, K = pixels.  A workgroup owns one tap row `a`, a block of
// NCO*16 couts and NCI*16 cins (TB*NCO*NCI accumulator tiles spread over its 4 waves) and walks
// a strided set of 128-pixel tiles, staging the dy tile and the matching input rows in LDS (with
// the producer's BN-apply + ReLU fused into the staging pass, exactly as the forward kernel sees
// the input).  Both MFMA operands need the PIXEL axis packed in-lane while memory has channels
// contiguous: bf16 uses ds_read_b64_tr_b16 (hardware transpose read), fp32 reads single dwords.
// Partials are flushed once per workgroup with fp32 atomics into the packed image
// [tap][Co16][Ci16] (64-byte contiguous segments), later unpacked by mfc_unpack_wgrad.
// Replaces: the cuDNN wgrad triggered by loss.backward() (src/engine.py:70) for every nn.Conv2d
// of models/hrnet.py and models/multiframe_model.py:191-201.
#include "common.h"
#include <cstdio>
#include <cstdlib>

static int g_wgrad_use_tr = 1;
static int g_wgrad_ksplit = 0;      // tuning: mfc_set_flag(3, 1) forces gc = gi = 1 (every wave holds all tiles, 4-way K split)
int mfc_conv_set_force_mt(int v);
int mfc_conv_set_grid(int v);
int mfc_conv_set_ablate(int v);
static int g_wgrad_ablate = 0;
static int g_wgrad_deep = 0;          // prefetch-distance-2 wave kernel when a launch has at most one workgroup per CU.  Alone it is 15 % faster
                                     // (35.7 -> 30.2 us), but its 308 VGPRs cannot share a SIMD with a conv wave (246), so next to the
                                     // critical chain the step gets SLOWER (546 -> 530 frames/s): off by default; mfc_set_flag(17, v)
static int g_wgrad_maxpx = 6000;     // output pixels one workgroup may walk before the pixel axis is split beyond g_wgrad_blocks (0 = never); mfc_set_flag(21, n)
int g_wgrad_blocks = 256;            // target workgroups per wave-kernel / DMA-kernel launch (S = blocks / Y).  The launches run on the detached stream
                                     // next to the critical chain.  Round 1 (register-staged kernel): 256 -> 575, 192 -> 582, 128 -> 586, 64 -> 556 frames/s.
                                     // Round 2 (LDS-DMA ring kernel, fused BatchNorm-backward epilogues): 128 and 256 give the same step time
                                     // (39.64 / 39.65 ms, tools/sweep_grid.sh) while the kernel alone is 1.5x faster at 256 (28.2 -> 18.7 us for
                                     // 32 -> 32 at 120x160: two workgroups per CU hide each other's fix-up pass), so 256; tuning: mfc_set_flag(11, n)
int mfc_conv_set_lds_kb(int v);
int mfc_conv_set_ybfast(int v);
int mfc_set_lanes(int on);
int mfc_set_probe_streams(int v);
int mfc_set_defer_join(int v);
int mfc_set_async_streams(int n);
int mfc_set_lane_streams(int n);
int mfc_set_async_on_lane(int k);
int mfc_set_own_main(int v);
int mfc_set_skip_kinds(int m);
int mfc_set_async_prio(int v);
int mfc_conv_set_fill_pct(int v);
int mfc_conv_set_nw8(int v);
int mfc_set_lane4_fwd(int v);
extern int g_conv_wres, g_conv_gemm, g_conv_gemm_minc, g_wgrad_gemm, g_wgrad_gemm_minc, g_bnred_blocks, g_wgrad_dma, g_conv_ring, g_ring_ablate, g_ring_wgs, g_ring_stagger, g_wgrad_dma_xf8, g_applyfin_blocks, g_ew_ablate, g_bnred_threads, g_bnred_minpx, g_wgrad_dma_s2, g_wgrad_dma48, g_wgrad_dma48_x2, g_ring_grid, g_conv_nw8_fused;
extern long g_ring_c64_unfused_px;
int mfc_ring_set_mt(int v);
extern "C" int mfc_set_flag(int id, int value) {
    if (id == 1) { g_wgrad_use_tr = value; return 0; }
    if (id == 2) return mfc_conv_set_force_mt(value);
    if (id == 3) { g_wgrad_ksplit = value; return 0; }
    if (id == 4) return mfc_conv_set_grid(value);
    if (id == 5) return mfc_conv_set_ablate(value);
    if (id == 6) return mfc_conv_set_lds_kb(value);
    if (id == 7) { g_wgrad_ablate = value; return 0; }
    if (id == 8) return mfc_conv_set_ybfast(value);
    if (id == 9) return mfc_set_lanes(value);
    if (id == 10) return mfc_set_async_streams(value);
    if (id == 12) return mfc_set_lane_streams(value);
    if (id == 13) return mfc_set_async_on_lane(value);
    if (id == 14) return mfc_set_own_main(value);
    if (id == 15) return mfc_set_skip_kinds(value);
    if (id == 16) return mfc_set_async_prio(value);
    if (id == 17) { g_wgrad_deep = value; return 0; }
    if (id == 18) return mfc_conv_set_fill_pct(value);
    if (id == 19) return mfc_conv_set_nw8(value);
    if (id == 20) { g_conv_wres = value; return 0; }
    if (id == 21) { g_wgrad_maxpx = value; return 0; }
    if (id == 22) return mfc_set_probe_streams(value);
    if (id == 23) { g_conv_gemm = value; return 0; }
    if (id == 24) { g_conv_gemm_minc = value; return 0; }
    if (id == 25) { g_wgrad_gemm = value; return 0; }
    if (id == 26) { g_wgrad_gemm_minc = value; return 0; }
    if (id == 27) { g_bnred_blocks = value > 0 ? value : 1024; return 0; }
    if (id == 28) return mfc_set_defer_join(value);
    if (id == 29) { g_wgrad_dma = value; return 0; }
    if (id == 30) { g_conv_ring = value; return 0; }
    if (id == 31) return mfc_ring_set_mt(value);
    if (id == 32) { g_ring_ablate = value; return 0; }
    if (id == 33) { g_ring_wgs = value > 0 ? value : 2; return 0; }
    if (id == 37) { g_ring_stagger = value; return 0; }
    if (id == 38) { g_wgrad_dma_xf8 = value; return 0; }
    if (id == 39) { g_applyfin_blocks = value > 0 ? value : 1024; return 0; }
    if (id == 40) { g_ew_ablate = value; return 0; }
    if (id == 41) { g_bnred_threads = value; return 0; }
    if (id == 42) { g_bnred_minpx = value > 0 ? value : 8; return 0; }
    if (id == 46) { g_wgrad_dma_s2 = value; return 0; }
    if (id == 47) { g_wgrad_dma48 = value; return 0; }
    if (id == 48) { g_wgrad_dma48_x2 = value; return 0; }
    if (id == 52) { g_ring_grid = value > 0 ? value : 0; return 0; }
    if (id == 53) { g_mfc_validate_ptrs = value ? 1 : 0; return 0; }
    if (id == 56) return mfc_set_lane4_fwd(value);
    if (id == 57) { g_ring_c64_unfused_px = (long)value * 1000; return 0; }
    if (id == 55) { g_conv_nw8_fused = value; return 0; }
    if (id == 54) { g_mfc_wt_min_mb = value > 0 ? value : 0; return 0; }
    if (id == 11) { g_wgrad_blocks = value > 0 ? value : 256; return 0; }
    return MFC_ERR_INVALID_ARG;
}

struct WgradK {
    const char* x; const char* dy; float* dwp; const float* in_coef;
    int N, Hin, Win, Cin_p, Hout, Wout, Cout_p;
    int TA, TB, dh0, dw0, s, in_relu, ipg;
    int TH, TW, tilesY, tilesX, ntiles, splits;
    int Co16, Ci16, NCO, NCI, co_blocks, ci_blocks;
    int PW, pitch_d, pitch_x, off_x, off_tab;
    int ntile_mm, cnt;      // accumulator tiles per block, per wave
    int slice;              // floats per partial-sum slice of dwp
};

template <typename T, int TPW, bool TR>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradK p) {
    constexpr int E = Gran<T>::E;
    constexpr bool BF = (E == 8);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ds = smem;
    char* Xs = smem + p.off_x;
    int* ptab = (int*)(smem + p.off_tab);        // [128] ty<<16|tx (or -1)
    int* xoff = ptab + 128;                      // [128] patch byte offset of pixel (tap b = 0)
    int* ttab = xoff + 128;                      // [4][TPW][4]: offA, offB, out index, valid

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int y = blockIdx.y;
    const int ib = y % p.ci_blocks; y /= p.ci_blocks;
    const int cb = y % p.co_blocks; const int a = y / p.co_blocks;
    const int co0 = cb * p.NCO * 16, ci0 = ib * p.NCI * 16;

    if (tid < 128) {
        int ty = tid / p.TW, tx = tid - ty * p.TW;
        bool v = tid < p.TH * p.TW;
        ptab[tid] = v ? ((ty << 16) | tx) : -1;
        xoff[tid] = v ? (ty * p.PW + tx * p.s) * p.pitch_x : 0;
    }
    for (int t = tid; t < 4 * TPW; t += 256) {
        int w = t / TPW, i = t - w * TPW;
        int tile = w * p.cnt + i;
        bool v = (i < p.cnt) && (tile < p.ntile_mm);
        int b = 0, cot = 0, cit = 0;
        if (v) { cit = tile % p.NCI; int r = tile / p.NCI; cot = r % p.NCO; b = r / p.NCO; }
        ttab[t * 4 + 0] = cot * 16 * (int)sizeof(T);
        ttab[t * 4 + 1] = b * p.pitch_x + cit * 16 * (int)sizeof(T);
        ttab[t * 4 + 2] = ((a * p.TB + b) * p.Co16 + co0 + cot * 16) * p.Ci16 + ci0 + cit * 16;
        ttab[t * 4 + 3] = (v && co0 + cot * 16 < p.Co16 && ci0 + cit * 16 < p.Ci16) ? 1 : 0;
    }

    f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int gd = p.NCO * 16 / E, gx = p.NCI * 16 / E;
    int GDP = 1; while (GDP < gd) GDP <<= 1;
    int GXP = 1; while (GXP < gx) GXP <<= 1;
    const int npx = p.TH * p.PW;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += p.splits) {
        int r = tile;
        const int tx_i = r % p.tilesX; r /= p.tilesX;
        const int ty_i = r % p.tilesY; const int n = r / p.tilesY;
        const int i0 = ty_i * p.TH, j0 = tx_i * p.TW;
        const int grp = n / p.ipg;
        __syncthreads();                      // tables ready / previous tile consumed
        {   // ---- dy tile -> Ds[pixel][co] ----
            const int gi = tid & (GDP - 1);
            const bool gok = gi < gd && (co0 + gi * E) < p.Cout_p;
            if (gi < gd) {
                for (int pp = tid / GDP; pp < 128; pp += 256 / GDP) {
                    int tt = ptab[pp];
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (tt >= 0 && gok) {
                        int oi = i0 + (tt >> 16), oj = j0 + (tt & 0xffff);
                        if (oi < p.Hout && oj < p.Wout)
                            v = *(const uint4*)(p.dy + ((((size_t)n * p.Hout + oi) * p.Wout + oj) * p.Cout_p + co0) * sizeof(T) + (size_t)gi * 16);
                    }
                    *(uint4*)(Ds + pp * p.pitch_d + gi * 16) = v;
                }
            }
        }
        {   // ---- input rows of tap row a -> Xs[ty][px][ci], fused BN-apply + ReLU ----
            const int gi = tid & (GXP - 1);
            const bool gok = gi < gx && (ci0 + gi * E) < p.Cin_p;
            float sc[E], sh[E];
            const bool xf = (p.in_coef != nullptr) && gok;
            if (xf) {
                const float* cf = p.in_coef + (size_t)grp * 4 * p.Cin_p + ci0 + gi * E;
#pragma unroll
                for (int e = 0; e < E; ++e) { sc[e] = cf[e]; sh[e] = cf[p.Cin_p + e]; }
            }
            if (gi < gx) {
                for (int pix = tid / GXP; pix < npx; pix += 256 / GXP) {
                    int ty = pix / p.PW, px = pix - ty * p.PW;
                    int ih = (i0 + ty) * p.s + p.dh0 + a, iw = j0 * p.s + p.dw0 + px;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (gok && ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Win) {
                        v = *(const uint4*)(p.x + ((((size_t)n * p.Hin + ih) * p.Win + iw) * p.Cin_p + ci0) * sizeof(T) + (size_t)gi * 16);
                        if (xf) {
                            float f[E];
                            Gran<T>::unpack(v, f);
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                float t = f[e] * sc[e] + sh[e];
                                f[e] = p.in_relu ? relu_nan(t) : t;
                            }
                            v = Gran<T>::pack(f);
                        }
                    }
                    *(uint4*)(Xs + pix * p.pitch_x + gi * 16) = v;
                }
            }
        }
        __syncthreads();
        if constexpr (BF) {
#pragma unroll 1
            for (int ks = 0; ks < 4; ++ks) {
                if constexpr (TR) {
                    const int pr0 = ks * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2);
                    const int csub = (lane & 3) * 8;            // 4 bf16 = 8 bytes
                    const int dA0 = pr0 * p.pitch_d + csub, dA1 = (pr0 + 4) * p.pitch_d + csub;
                    const int dB0 = xoff[pr0] + csub, dB1 = xoff[pr0 + 4] + csub;
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        const int* tt = ttab + (wave * TPW + i) * 4;
                        if (i < p.cnt) {
                            const int oa = tt[0], ob = tt[1];
                            typedef __attribute__((address_space(3))) s16x4 lds_s4;
                            s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(Ds + dA0 + oa));
                            s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(Ds + dA1 + oa));
                            s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(Xs + dB0 + ob));
                            s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(Xs + dB1 + ob));
                            typedef __attribute__((ext_vector_type(8))) short s16x8;
                            s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                            s16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                            acc[i] = mfma16<T>(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[i]);
                        }
                    }
                } else {
                    const int pk = ks * 32 + 8 * (lane >> 4);
                    const int cl = (lane & 15) * 2;
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        const int* tt = ttab + (wave * TPW + i) * 4;
                        if (i < p.cnt) {
                            const int oa = tt[0], ob = tt[1];
                            typedef __attribute__((ext_vector_type(8))) short s16x8;
                            s16x8 av, bv;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                av[j] = *(const short*)(Ds + (pk + j) * p.pitch_d + oa + cl);
                                bv[j] = *(const short*)(Xs + xoff[pk + j] + ob + cl);
                            }
                            acc[i] = mfma16<T>(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[i]);
                        }
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int ks = 0; ks < 32; ++ks) {
                const int pr = ks * 4 + (lane >> 4);
                const int dA = pr * p.pitch_d + (lane & 15) * 4;
                const int dB = xoff[pr] + (lane & 15) * 4;
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int* tt = ttab + (wave * TPW + i) * 4;
                    if (i < p.cnt) {
                        float av = *(const float*)(Ds + dA + tt[0]);
                        float bv = *(const float*)(Xs + dB + tt[1]);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- flush: D[row = co][col = ci]; lane holds col = lane&15, rows 4*(lane>>4)+r ----
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int* tt = ttab + (wave * TPW + i) * 4;
        if (i < p.cnt && tt[3]) {
            float* o = p.dwp + (size_t)blockIdx.x * p.slice + tt[2] + (size_t)((lane >> 4) * 4) * p.Ci16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(size_t)r * p.Ci16] = acc[i][r];
        }
    }
}

// ================================================================================================
// Fast bf16 path: register-tiled, double-buffered, prefetched.
//   * a wave owns a TB x WCO x WCI block of 16x16 accumulator tiles: per 32-pixel k-step it reads WCO
//     dy^T fragments once (shared by all taps) and TB*WCI input fragments -> ~0.9 transpose reads / MFMA;
//   * the 4 waves of a workgroup are arranged gc x gi x gk: gc*WCO cout tiles, gi*WCI cin tiles, and gk-way
//     split of the pixel (K) axis (small channel counts: every wave holds ALL tiles and takes every gk-th
//     k-step; partial sums meet in the final atomics);
//   * the dy tile and the input rows of the NEXT pixel tile are prefetched into registers while the MFMAs
//     of the current one run from the other LDS buffer: one barrier per pixel tile.
// ================================================================================================
// prefetch piece budgets (16-B pieces per thread per pixel tile): BIG = wide channel blocks (1 workgroup / CU, up to 512 VGPRs),
// small = narrow blocks (2 workgroups / CU, <= 256 VGPRs).  Spilling here serialises the prefetch loads: never spill.

struct WgradF {
    const char* x; const char* dy; float* dwp; const float* in_coef;
    int N, Hin, Win, Cin_p, Hout, Wout, Cout_p;
    int TA, TB, dh0, dw0, s, in_relu, ipg;
    int TH, TW, tilesY, tilesX, ntiles, splits;
    int Co16, Ci16, co_blocks, ci_blocks;
    int gc, gi, gk;                   // wave arrangement
    int gd, gx;                       // granules per pixel staged for dy / x
    int PW, pitch_d, pitch_x, buf_bytes, off_x, off_tab, off_coef, G;
    int slice;                        // floats per partial-sum slice of dwp
    int f16;                          // element type: 0 bf16, 1 fp16 (host-side dispatch only)
};

template <typename TE, int TB, int WCO, int WCI, bool BIG>
__global__ __launch_bounds__(256, BIG ? 1 : 2) void conv_wgrad_fast_kernel(WgradF p) {
    typedef TE T;
    constexpr int WG_DP = BIG ? 6 : 3, WG_XP = BIG ? 7 : 4;
    constexpr int E = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* xoff = (int*)(smem + p.off_tab);        // [128] patch byte offset of pixel (tap b = 0), 0 for padded pixels
    float* coefs = (float*)(smem + p.off_coef);  // [G][2][gx*8] scale / shift of this block's input channels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, weight block fastest, XCD-contiguous: the Y workgroups that walk the SAME pixel tiles run on one XCD at the
    // same time, so their re-reads of the dy / x tiles hit that XCD's L2 (PMC: 8-10x over-fetch from HBM otherwise)
    const int Ytot = p.TA * p.co_blocks * p.ci_blocks;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    int y = Lb % Ytot;
    const int bsplit = Lb / Ytot;
    const int ib = y % p.ci_blocks; y /= p.ci_blocks;
    const int cb = y % p.co_blocks; const int a = y / p.co_blocks;
    const int wc = wave % p.gc, wi = (wave / p.gc) % p.gi, wk = wave / (p.gc * p.gi);
    const int co0 = cb * p.gc * WCO * 16, ci0 = ib * p.gi * WCI * 16;      // block origin (channels)
    const int cow = wc * WCO * 16, ciw = wi * WCI * 16;                    // wave origin inside the block

    if (tid < 128) {
        int ty = tid / p.TW, tx = tid - ty * p.TW;
        xoff[tid] = (tid < p.TH * p.TW) ? (ty * p.PW + tx * p.s) * p.pitch_x : 0;
    }
    if (p.in_coef) {
        const int nch = p.gx * 8;
        for (int i = tid; i < p.G * 2 * nch; i += 256) {
            const int ch = i % nch, w = (i / nch) & 1, g = i / (2 * nch);
            coefs[i] = (ci0 + ch < p.Cin_p) ? p.in_coef[((size_t)g * 4 + w) * p.Cin_p + ci0 + ch] : 0.f;
        }
    }
    // staging tables (dense): piece idx = tid + i*256 -> (pixel, granule); packed as gi | px<<4 | ty<<12, or -1 (unused),
    // bit 30 set = pixel row beyond the tile (stored as zeros)
    const int npx = p.TH * p.PW;
    int dpk[WG_DP], xpk[WG_XP];
#pragma unroll
    for (int i = 0; i < WG_DP; ++i) {
        const int idx = tid + i * 256;
        const int pp = idx / p.gd, gi = idx - pp * p.gd;
        const int ty = pp / p.TW, tx = pp - ty * p.TW;
        dpk[i] = (pp < 128) ? (gi | (tx << 4) | (ty << 12) | ((pp >= p.TH * p.TW || (co0 + gi * E) >= p.Cout_p) ? (1 << 30) : 0)) : -1;
    }
#pragma unroll
    for (int i = 0; i < WG_XP; ++i) {
        const int idx = tid + i * 256;
        const int pix = idx / p.gx, gi = idx - pix * p.gx;
        const int ty = pix / p.PW, px = pix - ty * p.PW;
        xpk[i] = (pix < npx) ? (gi | (px << 4) | (ty << 12) | (((ci0 + gi * E) >= p.Cin_p) ? (1 << 30) : 0)) : -1;
    }

    f32x4 acc[TB][WCO][WCI];
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
        for (int i = 0; i < WCO; ++i)
#pragma unroll
            for (int j = 0; j < WCI; ++j) acc[b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 dreg[WG_DP], xreg[WG_XP]; unsigned dmask = 0, xmask = 0;
    int x_grp = 0;
    const bool xf = (p.in_coef != nullptr);

    auto tile_coords = [&](int tile, int& n, int& i0, int& j0) {
        const int txi = tile % p.tilesX; int r = tile / p.tilesX;
        const int tyi = r % p.tilesY; n = r / p.tilesY;
        i0 = tyi * p.TH; j0 = txi * p.TW;
    };
    auto load_tile = [&](int tile) {
        int n, i0, j0; tile_coords(tile, n, i0, j0);
        dmask = 0; xmask = 0;
        x_grp = n / p.ipg;
        // Branch-free prefetch (see conv_igemm.hip load_patch): clamped addresses, unconditional loads, validity in a bit mask.
        const char* dbase = p.dy + ((size_t)n * p.Hout * p.Wout * p.Cout_p + co0) * sizeof(T);
#pragma unroll
        for (int i = 0; i < WG_DP; ++i) {
            const int oi = i0 + ((dpk[i] >> 12) & 0xff), oj = j0 + ((dpk[i] >> 4) & 0xff);
            const bool inr = dpk[i] >= 0 && !(dpk[i] & (1 << 30)) && oi < p.Hout && oj < p.Wout;
            const int oic = min(oi, p.Hout - 1), ojc = min(oj, p.Wout - 1);
            const int gic = ((co0 + (dpk[i] & 15) * E) < p.Cout_p) ? (dpk[i] & 15) : 0;
            dreg[i] = *(const uint4*)(dbase + (size_t)(oic * p.Wout + ojc) * (p.Cout_p * (int)sizeof(T)) + gic * 16);
            dmask |= (inr ? 1u : 0u) << i;
        }
        const char* xbase = p.x + ((size_t)n * p.Hin * p.Win * p.Cin_p + ci0) * sizeof(T);
        const int ihb = i0 * p.s + p.dh0 + a, iwb = j0 * p.s + p.dw0;
#pragma unroll
        for (int i = 0; i < WG_XP; ++i) {
            const int ih = ihb + ((xpk[i] >> 12) & 0xff) * p.s, iw = iwb + ((xpk[i] >> 4) & 0xff);
            const bool inr = xpk[i] >= 0 && !(xpk[i] & (1 << 30)) && ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Win;
            const int ihc = min(max(ih, 0), p.Hin - 1), iwc = min(max(iw, 0), p.Win - 1);
            const int gic = ((ci0 + (xpk[i] & 15) * E) < p.Cin_p) ? (xpk[i] & 15) : 0;
            xreg[i] = *(const uint4*)(xbase + (size_t)(ihc * p.Win + iwc) * (p.Cin_p * (int)sizeof(T)) + gic * 16);
            xmask |= (inr ? 1u : 0u) << i;
        }
    };
    auto store_tile = [&](char* buf) {
        char* Ds = buf; char* Xs = buf + p.off_x;
#pragma unroll
        for (int i = 0; i < WG_DP; ++i) {
            if (dpk[i] >= 0) {
                const int row = ((dpk[i] >> 12) & 0xff) * p.TW + ((dpk[i] >> 4) & 0xff);     // = pixel index inside the tile
                *(uint4*)(Ds + row * p.pitch_d + (dpk[i] & 15) * 16) = (dmask & (1u << i)) ? dreg[i] : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < WG_XP; ++i) {
            if (xpk[i] >= 0) {
                uint4 v = make_uint4(0, 0, 0, 0);
                const int gi = xpk[i] & 15;
                if (xmask & (1u << i)) {
                    v = xreg[i];
                    if (xf) {
                        const float* cf = coefs + (size_t)x_grp * 2 * (p.gx * 8) + gi * E;
                        float f[E];
                        Gran<T>::unpack(v, f);
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            float t = f[e] * cf[e] + cf[p.gx * 8 + e];
                            f[e] = p.in_relu ? relu_nan(t) : t;
                        }
                        v = Gran<T>::pack(f);
                    }
                }
                *(uint4*)(Xs + (((xpk[i] >> 12) & 0xff) * p.PW + ((xpk[i] >> 4) & 0xff)) * p.pitch_x + gi * 16) = v;
            }
        }
    };

    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    int it = 0;
    int tile = bsplit;
    if (tile < p.ntiles) { load_tile(tile); store_tile(smem); }
    __syncthreads();
    for (; tile < p.ntiles; tile += p.splits, ++it) {
        const int nxt = tile + p.splits;
        if (nxt < p.ntiles) load_tile(nxt);
        const char* Ds = smem + (it & 1) * p.buf_bytes;
        const char* Xs = Ds + p.off_x;
        const int csub = (lane & 3) * 8;
#pragma unroll 1
        for (int ks = wk; ks < 4; ks += p.gk) {
            const int pr0 = ks * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2);
            const char* dA0 = Ds + pr0 * p.pitch_d + cow * 2 + csub;
            const char* dA1 = dA0 + 4 * p.pitch_d;
            const char* dB0 = Xs + xoff[pr0] + ciw * 2 + csub;
            const char* dB1 = Xs + xoff[pr0 + 4] + ciw * 2 + csub;
            bf16x8 af[WCO];
#pragma unroll
            for (int i = 0; i < WCO; ++i) {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dA0 + i * 32));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dA1 + i * 32));
                af[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                bf16x8 bfr[WCI];
#pragma unroll
                for (int j = 0; j < WCI; ++j) {
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dB0 + b * p.pitch_x + j * 32));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dB1 + b * p.pitch_x + j * 32));
                    bfr[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int i = 0; i < WCO; ++i)
#pragma unroll
                    for (int j = 0; j < WCI; ++j)
                        acc[b][i][j] = mfma16<T>(af[i], bfr[j], acc[b][i][j]);
            }
        }
        if (nxt < p.ntiles) store_tile(smem + ((it + 1) & 1) * p.buf_bytes);
        __syncthreads();
    }
    // ---- K-split waves: tree-reduce the accumulators through LDS so that only one wave per tile set issues atomics ----
    bool flusher = true;
    if (p.gk > 1) {
        constexpr int NTW = TB * WCO * WCI;
        const int stride = p.gc * p.gi;                   // waves with equal (wc, wi) are `stride` apart
        for (int half = p.gk >> 1; half >= 1; half >>= 1) {
            __syncthreads();                              // staging buffers / previous round no longer read
            const bool dump = flusher && wk >= half && wk < 2 * half;
            const bool take = flusher && wk < half;
            char* region = smem + (size_t)((wk % half) * stride + wc + p.gc * wi) * (NTW * 1024);
            if (dump) {
                int t = 0;
#pragma unroll
                for (int b = 0; b < TB; ++b)
#pragma unroll
                    for (int i = 0; i < WCO; ++i)
#pragma unroll
                        for (int j = 0; j < WCI; ++j, ++t) *(f32x4*)(region + (t * 64 + lane) * 16) = acc[b][i][j];
                flusher = false;
            }
            __syncthreads();
            if (take) {
                int t = 0;
#pragma unroll
                for (int b = 0; b < TB; ++b)
#pragma unroll
                    for (int i = 0; i < WCO; ++i)
#pragma unroll
                        for (int j = 0; j < WCI; ++j, ++t) acc[b][i][j] += *(const f32x4*)(region + (t * 64 + lane) * 16);
            }
        }
    }
    // ---- flush: D[row = co][col = ci]; lane holds col = lane&15, rows 4*(lane>>4)+r ----
    if (flusher) {
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int i = 0; i < WCO; ++i)
#pragma unroll
                for (int j = 0; j < WCI; ++j) {
                    const int co = co0 + cow + i * 16 + (lane >> 4) * 4, ci = ci0 + ciw + j * 16 + (lane & 15);
                    if (co < p.Co16 && ci < p.Ci16) {
                        float* o = p.dwp + (size_t)bsplit * p.slice + ((size_t)(a * TB + b) * p.Co16 + co) * p.Ci16 + ci;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[(size_t)r * p.Ci16] = acc[b][i][j][r];
                    }
                }
    }
}

// ================================================================================================
// Wave-private K-split path (3x3 / 11x11 taps): every wave of the workgroup walks its OWN sequence of
// 32-pixel sub-tiles (one MFMA k-step each), stages them in its own double-buffered LDS slice and holds all
// TB*WCO*WCI accumulator tiles of the workgroup's 48x48-channel block.  No workgroup barrier in the main loop:
// a wave's LDS writes and reads are ordered by the LDS pipeline itself, so the 8 waves of a CU drift apart and
// hide each other's global/LDS latency.  The four partial sums meet in an LDS tree reduction at the end.
// ================================================================================================
struct WgradW {
    const char* x; const char* dy; float* dwp; const float* in_coef;
    int N, Hin, Win, Cin_p, Hout, Wout, Cout_p;
    int TA, TB, dh0, dw0, s, in_relu, ipg, G;
    int TH, TW, tilesY, tilesX, ntiles, splits;     // sub-tile (TH*TW <= 32)
    int Co16, Ci16, co_blocks, ci_blocks;
    int gd, gx, PW, PHX, pitch_d, pitch_x, off_x, buf_bytes, wave_bytes, off_tab, off_coef;
    int slice;       // floats per partial-sum slice of dwp
    int ablate;      // tuning only: 1 skip global loads, 2 skip LDS staging stores, 4 skip LDS reads + MFMAs, 8 skip reduction + atomics
    int f16;         // element type: 0 bf16, 1 fp16 (host-side dispatch only)
};

// Up to 8 weight gradients of IDENTICAL geometry share one launch (mfc_conv2d_wgrad_batch): the workgroups are dealt to the
// problems round-robin, so each problem's pixel axis is split over 1/n of the grid and the per-workgroup fixed costs
// (prologue, LDS tree reduction, partial-sum store) are paid once per n times more pixels.
#define MFC_WGRAD_MAXBATCH 8
struct WgradBatch {
    const char* x[MFC_WGRAD_MAXBATCH]; const char* dy[MFC_WGRAD_MAXBATCH]; float* dwp[MFC_WGRAD_MAXBATCH];
    const float* coef[MFC_WGRAD_MAXBATCH]; int relu[MFC_WGRAD_MAXBATCH]; int n;
};

template <typename TE, int TAA, int TB, int WCO, int WCI, int XP, int PF>
__global__ __launch_bounds__(256, PF == 2 ? 1 : 2) void conv_wgrad_wave_kernel(WgradW p, WgradBatch tb) {
    typedef TE T;
    constexpr int E = 8;
    constexpr int DP = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* xoff = (int*)(smem + p.off_tab);        // [32] patch byte offset of sub-tile pixel (tap b = 0)
    float* coefs = (float*)(smem + p.off_coef);  // [G][2][gx*8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* wbase = smem + wave * p.wave_bytes;
    // 1-D grid, weight block fastest, XCD-contiguous (see conv_wgrad_fast_kernel): co-scheduled re-reads hit the XCD's L2
    const int Ytot = (p.TA / TAA) * p.co_blocks * p.ci_blocks;
    int Lb = xcd_remap(blockIdx.x, gridDim.x);
    if (tb.n > 1) {             // this workgroup's problem (wave-uniform: scalar loads from the kernel arguments)
        const int prob = Lb % tb.n;
        Lb /= tb.n;
        p.x = tb.x[prob]; p.dy = tb.dy[prob]; p.dwp = tb.dwp[prob]; p.in_coef = tb.coef[prob]; p.in_relu = tb.relu[prob];
    }
    int y = Lb % Ytot;
    const int bsplit = Lb / Ytot;
    const int ib = y % p.ci_blocks; y /= p.ci_blocks;
    const int cb = y % p.co_blocks; const int a = (y / p.co_blocks) * TAA;     // first tap row of this workgroup
    const int co0 = cb * WCO * 16, ci0 = ib * WCI * 16;

    if (tid < 32) {
        int ty = tid / p.TW, tx = tid - ty * p.TW;
        xoff[tid] = (tid < p.TH * p.TW) ? (ty * p.s * p.PW + tx * p.s) * p.pitch_x : 0;
    }
    if (p.in_coef) {
        const int nch = p.gx * 8;
        for (int i = tid; i < p.G * 2 * nch; i += 256) {
            const int ch = i % nch, w = (i / nch) & 1, g = i / (2 * nch);
            coefs[i] = (ci0 + ch < p.Cin_p) ? p.in_coef[((size_t)g * 4 + w) * p.Cin_p + ci0 + ch] : 0.f;
        }
    }
    // per-lane staging pieces: idx = lane + i*64 -> (pixel, granule); packed gi | px<<4 | ty<<12.  Every piece has
    //   *_lds : its LDS byte offset inside a staging buffer -- pieces that never carry data (beyond the sub-tile / the
    //           channel range) point at a dummy slot; their real slots are zeroed once below and never written again
    //   *_vof : its byte offset from the first pixel of an INTERIOR tile (wave-uniform base + constant lane offset: the
    //           main loop then spends no vector instruction on addresses)
    const int npx = p.PHX * p.PW;               // staged input rows: ((TH-1)*s + TAA) x PW
    const int dummy = p.buf_bytes - 16;         // (host pads every buffer by 16 B)
    int dpk[DP], xpk[XP], d_lds[DP], x_lds[XP], d_vof[DP], x_vof[XP];
    unsigned dstat = 0, xstat = 0;              // bit i: piece i can carry data
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        const int idx = lane + i * 64;
        const int pp = idx / p.gd, gi = idx - pp * p.gd;
        const int ty = pp / p.TW, tx = pp - ty * p.TW;
        const bool ok = pp < p.TH * p.TW && (co0 + gi * E) < p.Cout_p;
        dpk[i] = gi | (tx << 4) | (ty << 12);
        dstat |= (ok ? 1u : 0u) << i;
        d_lds[i] = ok ? pp * p.pitch_d + gi * 16 : dummy;
        d_vof[i] = ok ? ((ty * p.Wout + tx) * p.Cout_p * (int)sizeof(T) + gi * 16) : 0;
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int idx = lane + i * 64;
        const int pix = idx / p.gx, gi = idx - pix * p.gx;
        const int ty = pix / p.PW, px = pix - ty * p.PW;
        const bool ok = pix < npx && (ci0 + gi * E) < p.Cin_p;
        xpk[i] = gi | (px << 4) | (ty << 12);
        xstat |= (ok ? 1u : 0u) << i;
        x_lds[i] = ok ? p.off_x + pix * p.pitch_x + gi * 16 : dummy;
        x_vof[i] = ok ? ((ty * p.Win + px) * p.Cin_p * (int)sizeof(T) + gi * 16) : 0;
    }
    for (int o = lane * 16; o < p.wave_bytes; o += 64 * 16) *(uint4*)(wbase + o) = make_uint4(0, 0, 0, 0);

    f32x4 acc[TAA * TB][WCO][WCI];
#pragma unroll
    for (int b = 0; b < TAA * TB; ++b)
#pragma unroll
        for (int i = 0; i < WCO; ++i)
#pragma unroll
            for (int j = 0; j < WCI; ++j) acc[b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // one staged sub-tile in registers (global loads in flight or landed) with what the LDS store needs to know about it
    struct Stage { uint4 d[DP]; uint4 x[XP]; unsigned dmask, xmask; int grp; bool interior; };
    const bool xf = (p.in_coef != nullptr);
    const float relu_floor = p.in_relu ? 0.f : -3.0e38f;

    // tile coordinates are tracked incrementally (stride decomposed once: no division per tile)
    const int stride = p.splits * 4;
    const int st_x = stride % p.tilesX, st_y = (stride / p.tilesX) % p.tilesY, st_n = stride / (p.tilesX * p.tilesY);
    struct TC { int n, tyi, txi; };
    auto tc_next = [&](TC c) {
        c.txi += st_x; if (c.txi >= p.tilesX) { c.txi -= p.tilesX; ++c.tyi; }
        c.tyi += st_y; if (c.tyi >= p.tilesY) { c.tyi -= p.tilesY; ++c.n; }
        c.n += st_n;
        return c;
    };

    auto load_tile = [&](const TC& c, Stage& S) {
        const int n = c.n, i0 = c.tyi * p.TH, j0 = c.txi * p.TW;
        S.grp = n / p.ipg;
        const int ihb = i0 * p.s + p.dh0 + a, iwb = j0 * p.s + p.dw0;
        S.interior = i0 + p.TH <= p.Hout && j0 + p.TW <= p.Wout && ihb >= 0 && ihb + p.PHX <= p.Hin && iwb >= 0 && iwb + p.PW <= p.Win;
        const char* dimg = p.dy + ((size_t)n * p.Hout * p.Wout * p.Cout_p + co0) * sizeof(T);
        const char* ximg = p.x + ((size_t)n * p.Hin * p.Win * p.Cin_p + ci0) * sizeof(T);
        if (S.interior) {
            const char* dt = dimg + (size_t)(i0 * p.Wout + j0) * (p.Cout_p * (int)sizeof(T));
            const char* xt = ximg + (size_t)(ihb * p.Win + iwb) * (p.Cin_p * (int)sizeof(T));
#pragma unroll
            for (int i = 0; i < DP; ++i) S.d[i] = *(const uint4*)(dt + (unsigned)d_vof[i]);
#pragma unroll
            for (int i = 0; i < XP; ++i) S.x[i] = *(const uint4*)(xt + (unsigned)x_vof[i]);
        } else {
            S.dmask = 0; S.xmask = 0;
            const int drow = p.Cout_p * (int)sizeof(T), xrow = p.Cin_p * (int)sizeof(T);
#pragma unroll
            for (int i = 0; i < DP; ++i) {
                const int oi = i0 + ((dpk[i] >> 12) & 0xff), oj = j0 + ((dpk[i] >> 4) & 0xff);
                const bool inr = ((dstat >> i) & 1u) && oi < p.Hout && oj < p.Wout;
                const int oic = min(oi, p.Hout - 1), ojc = min(oj, p.Wout - 1);
                const int gic = ((dstat >> i) & 1u) ? (dpk[i] & 15) : 0;
                S.d[i] = *(const uint4*)(dimg + (unsigned)((oic * p.Wout + ojc) * drow + gic * 16));
                S.dmask |= (inr ? 1u : 0u) << i;
            }
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int ih = ihb + ((xpk[i] >> 12) & 0xff), iw = iwb + ((xpk[i] >> 4) & 0xff);
                const bool inr = ((xstat >> i) & 1u) && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
                const int ihc = min(max(ih, 0), p.Hin - 1), iwc = min(max(iw, 0), p.Win - 1);
                const int gic = ((xstat >> i) & 1u) ? (xpk[i] & 15) : 0;
                S.x[i] = *(const uint4*)(ximg + (unsigned)((ihc * p.Win + iwc) * xrow + gic * 16));
                S.xmask |= (inr ? 1u : 0u) << i;
            }
        }
    };
    // every lane stores every piece (no divergence): dead pieces go to the dummy slot
    auto store_tile = [&](char* buf, const Stage& S) {
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            uint4 v = S.d[i];
            if (!S.interior && !(S.dmask & (1u << i))) v = make_uint4(0, 0, 0, 0);
            *(uint4*)(buf + d_lds[i]) = v;
        }
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            uint4 v = S.x[i];
            if (xf) {
                const float* cf = coefs + (size_t)S.grp * 2 * (p.gx * 8) + (xpk[i] & 15) * E;
                float f[E];
                Gran<T>::unpack(v, f);
#pragma unroll
                // (v_max swallows a NaN of x here, unlike relu_nan: harmless in a weight gradient -- a non-finite activation has already made
                //  the loss, hence dy and every product below, non-finite; one VALU op per element instead of compare + select)
                for (int e = 0; e < E; ++e) f[e] = fmaxf(f[e] * cf[e] + cf[p.gx * 8 + e], relu_floor);
                v = Gran<T>::pack(f);
            }
            if (!S.interior && !(S.xmask & (1u << i))) v = make_uint4(0, 0, 0, 0);
            *(uint4*)(buf + x_lds[i]) = v;
        }
    };

    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    __syncthreads();                               // tables + coefficients visible
    int tile = bsplit * 4 + wave;
    TC tc;
    { tc.txi = tile % p.tilesX; const int r = tile / p.tilesX; tc.tyi = r % p.tilesY; tc.n = r / p.tilesY; }
    const int pr0 = 8 * (lane >> 4) + ((lane & 15) >> 2);
    const int csub = (lane & 3) * 8;
    const int xo0 = xoff[pr0] + csub, xo1 = xoff[pr0 + 4] + csub;
    auto compute = [&](int par) {                  // all MFMAs of the sub-tile staged in LDS buffer `par`
        const char* Ds = wbase + par * p.buf_bytes;
        const char* Xs = Ds + p.off_x;
        const char* dA0 = Ds + pr0 * p.pitch_d + csub;
        const char* dA1 = dA0 + 4 * p.pitch_d;
        if (p.ablate & 4) return;
        bf16x8 af[WCO];
#pragma unroll
        for (int i = 0; i < WCO; ++i) {
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dA0 + i * 32));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(dA1 + i * 32));
            af[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int b = 0; b < TAA * TB; ++b) {
            const int toff = ((b / TB) * p.PW + (b % TB)) * p.pitch_x;      // tap (row b/TB, column b%TB)
            bf16x8 bfr[WCI];
#pragma unroll
            for (int j = 0; j < WCI; ++j) {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(Xs + xo0 + toff + j * 32));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(Xs + xo1 + toff + j * 32));
                bfr[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < WCO; ++i)
#pragma unroll
                for (int j = 0; j < WCI; ++j)
                    acc[b][i][j] = mfma16<T>(af[i], bfr[j], acc[b][i][j]);
        }
    };
    if constexpr (PF == 1) {
        Stage S;
        if (tile < p.ntiles) { load_tile(tc, S); store_tile(wbase, S); }
        for (int it = 0; tile < p.ntiles; tile += stride, ++it) {
            const int nxt = tile + stride;
            tc = tc_next(tc);
            if (nxt < p.ntiles && !(p.ablate & 1)) load_tile(tc, S);
            compute(it & 1);
            if (nxt < p.ntiles && !(p.ablate & 2)) store_tile(wbase + ((it + 1) & 1) * p.buf_bytes, S);
        }
    } else {
        // Prefetch distance 2 (one workgroup per CU: a wave has a SIMD to itself, so nobody else hides its load latency; the
        // 512-register budget of that occupancy pays for a second set of staging registers).  While sub-tile t is multiplied
        // out of LDS buffer t&1, the loads of t+1 (issued an iteration ago) land and are stored into the other buffer, and the
        // loads of t+2 are already in flight.
        Stage SA, SB;
        if (tile < p.ntiles) {
            load_tile(tc, SA); store_tile(wbase, SA);
            tc = tc_next(tc);
            if (tile + stride < p.ntiles) load_tile(tc, SB);
        }
        for (;;) {
            if (tile >= p.ntiles) break;
            {   // even step: compute buffer 0, t+2 -> SA, land / store t+1 from SB into buffer 1
                const bool has1 = tile + stride < p.ntiles, has2 = tile + 2 * stride < p.ntiles;
                tc = tc_next(tc);
                if (has2 && !(p.ablate & 1)) load_tile(tc, SA);
                compute(0);
                if (has1 && !(p.ablate & 2)) store_tile(wbase + p.buf_bytes, SB);
                tile += stride;
            }
            if (tile >= p.ntiles) break;
            {   // odd step: compute buffer 1, t+2 -> SB, store t+1 from SA into buffer 0
                const bool has1 = tile + stride < p.ntiles, has2 = tile + 2 * stride < p.ntiles;
                tc = tc_next(tc);
                if (has2 && !(p.ablate & 1)) load_tile(tc, SB);
                compute(1);
                if (has1 && !(p.ablate & 2)) store_tile(wbase, SA);
                tile += stride;
            }
        }
    }
    if (p.ablate & 8) return;
    // ---- tree-reduce the four waves' accumulators through LDS, then one wave issues the atomics ----
    constexpr int NTW = TAA * TB * WCO * WCI;
    bool flusher = true;
    for (int half = 2; half >= 1; half >>= 1) {
        __syncthreads();
        const bool dump = flusher && wave >= half && wave < 2 * half;
        const bool take = flusher && wave < half;
        char* region = smem + (size_t)(wave % half) * (NTW * 1024);
        if (dump) {
            int t = 0;
#pragma unroll
            for (int b = 0; b < TAA * TB; ++b)
#pragma unroll
                for (int i = 0; i < WCO; ++i)
#pragma unroll
                    for (int j = 0; j < WCI; ++j, ++t) *(f32x4*)(region + (t * 64 + lane) * 16) = acc[b][i][j];
            flusher = false;
        }
        __syncthreads();
        if (take) {
            int t = 0;
#pragma unroll
            for (int b = 0; b < TAA * TB; ++b)
#pragma unroll
                for (int i = 0; i < WCO; ++i)
#pragma unroll
                    for (int j = 0; j < WCI; ++j, ++t) acc[b][i][j] += *(const f32x4*)(region + (t * 64 + lane) * 16);
        }
    }
    if (flusher) {
#pragma unroll
        for (int b = 0; b < TAA * TB; ++b)
#pragma unroll
            for (int i = 0; i < WCO; ++i)
#pragma unroll
                for (int j = 0; j < WCI; ++j) {
                    const int co = co0 + i * 16 + (lane >> 4) * 4, ci = ci0 + j * 16 + (lane & 15);
                    if (co < p.Co16 && ci < p.Ci16) {
                        float* o = p.dwp + (size_t)bsplit * p.slice + ((size_t)(a * TB + b) * p.Co16 + co) * p.Ci16 + ci;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[(size_t)r * p.Ci16] = acc[b][i][j][r];
                    }
                }
    }
}

void mfc_choose_tile_wg(int Hl, int Wl, int& TH, int& TW) {
    double best = -1; int bh = 8, bw = 16;
    for (int tw = 1; tw <= 128 && tw <= Wl; ++tw) {
        int th = 128 / tw; if (th > Hl) th = Hl; if (th < 1) continue;
        double tiles = (double)ceil_div(Hl, th) * ceil_div(Wl, tw);
        double eff = (double)Hl * Wl / (tiles * 128.0);
        double score = eff + ((tw % 16 == 0) ? 0.01 : 0.0) + 0.02 * (double)(th * tw) / ((th + 2.0) * (tw + 2.0));
        if (score > best) { best = score; bh = th; bw = tw; }
    }
    TH = bh; TW = bw;
}

template <typename T, int TPW, bool TR>
static int wgrad_launch(const WgradK& k, size_t lds, int Y, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<T, TPW, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        MFC_PROF_NAME(pname, "conv_wgrad_kernel<%s, %d, %s>", mfc_tname<T>(), TPW, TR ? "true" : "false");
        const double flops = 2.0 * k.N * k.Hout * k.Wout * (double)k.Co16 * k.Ci16 * k.TA * k.TB;
        const double bytes = ((double)k.N * k.Hin * k.Win * k.Cin_p + (double)k.N * k.Hout * k.Wout * k.Cout_p) * sizeof(T);
        mfc_prof_before(st, pname, flops, bytes);
    }
    hipLaunchKernelGGL((conv_wgrad_kernel<T, TPW, TR>), dim3(k.splits, Y), dim3(256), lds, st, k);
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

template <int TB, int WCO, int WCI, bool BIG>
static int wgrad_fast_launch(const WgradF& f, size_t lds, int Y, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_fast_kernel<bf16_t, TB, WCO, WCI, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_fast_kernel<f16_t, TB, WCO, WCI, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        const double flops = 2.0 * f.N * f.Hout * f.Wout * (double)f.Co16 * f.Ci16 * f.TA * f.TB;
        const double bytes = ((double)f.N * f.Hin * f.Win * f.Cin_p + (double)f.N * f.Hout * f.Wout * f.Cout_p) * 2.0;
        MFC_PROF_NAME(pname, "conv_wgrad_fast_kernel<__bf16, %d, %d, %d, %s>", TB, WCO, WCI, BIG ? "true" : "false");
        MFC_PROF_NAME(hname, "conv_wgrad_fast_kernel<_Float16, %d, %d, %d, %s>", TB, WCO, WCI, BIG ? "true" : "false");
        mfc_prof_before(st, f.f16 ? hname : pname, flops, bytes);
    }
    if (f.f16) hipLaunchKernelGGL((conv_wgrad_fast_kernel<f16_t, TB, WCO, WCI, BIG>), dim3(f.splits * Y), dim3(256), lds, st, f);
    else hipLaunchKernelGGL((conv_wgrad_fast_kernel<bf16_t, TB, WCO, WCI, BIG>), dim3(f.splits * Y), dim3(256), lds, st, f);
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// returns MFC_ERR_UNSUPPORTED when the geometry does not fit the fast kernel (caller falls back)
static int wgrad_fast(const mfc_wgrad_desc* d, hipStream_t st, int* parts_only) {
    if (!mfc_is16(d->dtype) || !g_wgrad_use_tr) return MFC_ERR_UNSUPPORTED;
    if (d->TB != 1 && d->TB != 3 && d->TB != 11) return MFC_ERR_UNSUPPORTED;
    WgradF f;
    f.f16 = d->dtype == MFC_F16;
    f.x = (const char*)d->x; f.dy = (const char*)d->dy; f.dwp = d->dwp; f.in_coef = d->in_coef;
    f.N = d->N; f.Hin = d->Hin; f.Win = d->Win; f.Cin_p = d->Cin_p; f.Hout = d->Hout; f.Wout = d->Wout; f.Cout_p = d->Cout_p;
    f.TA = d->TA; f.TB = d->TB; f.dh0 = d->dh0; f.dw0 = d->dw0; f.s = d->in_stride; f.in_relu = d->in_relu; f.ipg = d->images_per_group;
    f.TH = d->TH; f.TW = d->TW;
    if (f.TH <= 0 || f.TW <= 0) mfc_choose_tile_wg(d->Hout, d->Wout, f.TH, f.TW);
    if (f.TH * f.TW > 128) return MFC_ERR_INVALID_ARG;
    f.tilesY = ceil_div(d->Hout, f.TH); f.tilesX = ceil_div(d->Wout, f.TW);
    f.ntiles = f.N * f.tilesY * f.tilesX;
    f.Co16 = ceil_div(d->Cout, 16) * 16; f.Ci16 = ceil_div(d->Cin, 16) * 16;
    const int co_t = f.Co16 / 16, ci_t = f.Ci16 / 16;
    int WCO, WCI;
    if (d->TB == 11) { WCO = 1; WCI = ci_t >= 2 ? 2 : 1; }
    else {      // balanced blocks of <= 3 tiles (4 tiles -> 2+2, not 3+1)
        WCO = ceil_div(co_t, ceil_div(co_t, 3)); WCI = ceil_div(ci_t, ceil_div(ci_t, 3));
    }
    // wave arrangement.  3x3 / 11x11 taps: every wave holds ALL TB*WCO*WCI tiles of a 48x48-channel block and the 4 waves
    // split the pixel axis (small staging footprint -> 2 workgroups per CU; measured 1.1-2.4x faster than tile-splitting).
    // 1x1: few tiles per block otherwise -> spread 2x2 waves over cout/cin tiles.
    const bool tile_split = (d->TB == 1) && !g_wgrad_ksplit;
    f.gc = (tile_split && co_t > WCO) ? 2 : 1;
    f.gi = (tile_split && ci_t > WCI) ? 2 : 1;
    f.gk = 4 / (f.gc * f.gi);
    f.co_blocks = ceil_div(co_t, f.gc * WCO); f.ci_blocks = ceil_div(ci_t, f.gi * WCI);
    f.gd = f.gc * WCO * 2; f.gx = f.gi * WCI * 2;
    f.PW = (f.TW - 1) * f.s + f.TB;
    // pixel pitches: odd multiples of 16 B (+16) -- tr reads walk pixel rows
    f.pitch_d = f.gd * 16 + 16; f.pitch_x = f.gx * 16 + 16;
    const int ndp = ceil_div(128 * f.gd, 256), nxp = ceil_div(f.TH * f.PW * f.gx, 256);
    if (ndp > 6 || nxp > 7) return MFC_ERR_UNSUPPORTED;
    const bool big = ndp > 3 || nxp > 4;
    if (f.gd > 15 || f.gx > 15 || f.PW > 255 || f.TH > 255) return MFC_ERR_UNSUPPORTED;
    const size_t ds = ((size_t)128 * f.pitch_d + 15) & ~(size_t)15, xs = ((size_t)f.TH * f.PW * f.pitch_x + 15) & ~(size_t)15;
    f.off_x = (int)ds; f.buf_bytes = (int)(ds + xs); f.off_tab = 2 * f.buf_bytes;
    f.G = d->N / d->images_per_group;
    f.off_coef = f.off_tab + 128 * 4;
    size_t lds = (size_t)f.off_coef + (d->in_coef ? (size_t)f.G * 2 * f.gx * 8 * 4 : 0);
    if (f.gk > 1) {      // room for the accumulator tree reduction: (gk/2) * gc * gi waves dump TB*WCO*WCI KiB each
        const size_t need = (size_t)(f.gk / 2) * f.gc * f.gi * d->TB * WCO * WCI * 1024;
        if (lds < need) lds = need;
    }
    if (lds > (big ? 160 : 80) * 1024) return MFC_ERR_UNSUPPORTED;
    const int Y = f.TA * f.co_blocks * f.ci_blocks;
    int S = d->splits;
    if (S <= 0) S = ceil_div(big ? 256 : 512, Y);      // ~1 workgroup per CU-slot: each flushes its tiles once
    if (S > f.ntiles) S = f.ntiles;
    if (S < 1) S = 1;
    f.splits = S;
    f.slice = d->TA * d->TB * f.Co16 * f.Ci16;
    if (parts_only) { *parts_only = S; return MFC_OK; }      // (every WCO/WCI/TB combination reaching this point has an instantiation below)
    static const bool dbg = getenv("MFC_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[wgrad fast] TB%d WCO%d WCI%d g=%dx%dx%d Y=%d S=%d lds=%zu tile %dx%d\n", d->TB, WCO, WCI, f.gc, f.gi, f.gk, Y, S, lds, f.TH, f.TW);
#define WGF(tb, a_, b_) if (d->TB == tb && WCO == a_ && WCI == b_) return big ? wgrad_fast_launch<tb, a_, b_, true>(f, lds, Y, st) : wgrad_fast_launch<tb, a_, b_, false>(f, lds, Y, st);
    WGF(3, 3, 3) WGF(3, 3, 2) WGF(3, 3, 1) WGF(3, 2, 3) WGF(3, 2, 2) WGF(3, 2, 1) WGF(3, 1, 3) WGF(3, 1, 2) WGF(3, 1, 1)
    WGF(1, 3, 3) WGF(1, 3, 2) WGF(1, 3, 1) WGF(1, 2, 3) WGF(1, 2, 2) WGF(1, 2, 1) WGF(1, 1, 3) WGF(1, 1, 2) WGF(1, 1, 1)
    WGF(11, 1, 2) WGF(11, 1, 1)
#undef WGF
    return MFC_ERR_UNSUPPORTED;
}

template <int TAA, int TB, int WCO, int WCI, int XP, int PF>
static int wgrad_wave_launch(const WgradW& f, const WgradBatch& tb, size_t lds, int Y, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_wave_kernel<bf16_t, TAA, TB, WCO, WCI, XP, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_wave_kernel<f16_t, TAA, TB, WCO, WCI, XP, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        const double flops = 2.0 * f.N * f.Hout * f.Wout * (double)f.Co16 * f.Ci16 * f.TA * f.TB * tb.n;
        const double bytes = ((double)f.N * f.Hin * f.Win * f.Cin_p + (double)f.N * f.Hout * f.Wout * f.Cout_p) * 2.0 * tb.n;
        MFC_PROF_NAME(pname, "conv_wgrad_wave_kernel<__bf16, %d, %d, %d, %d, %d, %d>", TAA, TB, WCO, WCI, XP, PF);
        MFC_PROF_NAME(hname, "conv_wgrad_wave_kernel<_Float16, %d, %d, %d, %d, %d, %d>", TAA, TB, WCO, WCI, XP, PF);
        mfc_prof_before(st, f.f16 ? hname : pname, flops, bytes);
    }
    if (f.f16) hipLaunchKernelGGL((conv_wgrad_wave_kernel<f16_t, TAA, TB, WCO, WCI, XP, PF>), dim3(f.splits * Y * tb.n), dim3(256), lds, st, f, tb);
    else hipLaunchKernelGGL((conv_wgrad_wave_kernel<bf16_t, TAA, TB, WCO, WCI, XP, PF>), dim3(f.splits * Y * tb.n), dim3(256), lds, st, f, tb);
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// max_patch > 0: the (TH + 2) x (TW + 2) halo patch must not exceed it (all-taps staging budget)
static void choose_subtile(int Hl, int Wl, int& TH, int& TW, int max_patch = 0) {
    double best = -1; int bh = 0, bw = 0;
    for (int tw = 1; tw <= 32 && tw <= Wl; ++tw) {
        int th = 32 / tw; if (th > Hl) th = Hl; if (th < 1) continue;
        if (max_patch > 0 && (th + 2) * (tw + 2) > max_patch) continue;
        double tiles = (double)ceil_div(Hl, th) * ceil_div(Wl, tw);
        double eff = (double)Hl * Wl / (tiles * 32.0);
        double score = eff + ((tw % 8 == 0) ? 0.01 : 0.0) + 0.02 * (double)tw / (tw + 2.0);   // wide rows: less halo
        if (score > best) { best = score; bh = th; bw = tw; }
    }
    TH = bh; TW = bw;
}

static int wgrad_wave(const mfc_wgrad_desc* d, hipStream_t st, int* parts_only, const mfc_wgrad_desc* batch = nullptr, int nbatch = 1) {
    if (!mfc_is16(d->dtype) || !g_wgrad_use_tr || g_wgrad_ksplit == 2) return MFC_ERR_UNSUPPORTED;
    if (d->TB != 3 && d->TB != 11) return MFC_ERR_UNSUPPORTED;
    WgradW f;
    f.f16 = d->dtype == MFC_F16;
    f.x = (const char*)d->x; f.dy = (const char*)d->dy; f.dwp = d->dwp; f.in_coef = d->in_coef;
    f.N = d->N; f.Hin = d->Hin; f.Win = d->Win; f.Cin_p = d->Cin_p; f.Hout = d->Hout; f.Wout = d->Wout; f.Cout_p = d->Cout_p;
    f.TA = d->TA; f.TB = d->TB; f.dh0 = d->dh0; f.dw0 = d->dw0; f.s = d->in_stride; f.in_relu = d->in_relu; f.ipg = d->images_per_group;
    f.G = d->N / d->images_per_group;
    f.Co16 = ceil_div(d->Cout, 16) * 16; f.Ci16 = ceil_div(d->Cin, 16) * 16;
    const int co_t = f.Co16 / 16, ci_t = f.Ci16 / 16;
    const bool want_all = d->TB == 3 && d->TA == 3 && d->in_stride == 1 && co_t % 2 == 0 && ci_t % 2 == 0 && g_wgrad_ksplit != 3;
    int WCO, WCI;
    if (d->TB == 11) { WCO = 1; WCI = ci_t >= 2 ? 2 : 1; }
    else { WCO = ceil_div(co_t, ceil_div(co_t, 3)); WCI = ceil_div(ci_t, ceil_div(ci_t, 3)); }
    f.TH = 0; f.TW = 0;
    if (want_all) choose_subtile(d->Hout, d->Wout, f.TH, f.TW, 64);     // 64 patch pixels x 4 granules = 4 pieces per lane
    // all-taps mode (3x3, 32x32-channel blocks): one workgroup accumulates all 9 taps, so the input rows are staged once
    // for the three tap rows instead of once per row (3x fewer staging instructions per MFMA)
    const bool alltaps = want_all && f.TH > 0 && ceil_div(((f.TH - 1) * f.s + 3) * ((f.TW - 1) * f.s + 3) * 4, 64) <= 4;     // staging must fit 4 pieces/lane
    if (alltaps) { WCO = 2; WCI = 2; }
    const int TAA = alltaps ? 3 : 1;
    if (!alltaps) {
        // one tap row per workgroup: pick the 32-pixel sub-tile whose staged input rows ((TH-1)*s+1) x ((TW-1)*s+TB) fit
        // the 7 pieces per lane, preferring full tiles, then the smallest patch (strided convs want flat, wide tiles)
        double best = -1; int bh = 0, bw = 0;
        for (int tw = 1; tw <= 32 && tw <= d->Wout; ++tw) {
            int th = 32 / tw; if (th > d->Hout) th = d->Hout; if (th < 1) continue;
            const int phx = (th - 1) * f.s + 1, pw = (tw - 1) * f.s + d->TB;
            if (pw > 255 || ceil_div(phx * pw * WCI * 2, 64) > 7) continue;
            const double tiles = (double)ceil_div(d->Hout, th) * ceil_div(d->Wout, tw);
            const double eff = (double)d->Hout * d->Wout / (tiles * 32.0);
            const double score = eff - 0.0015 * (double)(phx * pw) / 32.0 * 4.0 + ((tw % 8 == 0) ? 0.01 : 0.0);
            if (score > best) { best = score; bh = th; bw = tw; }
        }
        if (bh == 0) return MFC_ERR_UNSUPPORTED;
        f.TH = bh; f.TW = bw;
    }
    f.tilesY = ceil_div(d->Hout, f.TH); f.tilesX = ceil_div(d->Wout, f.TW);
    f.ntiles = f.N * f.tilesY * f.tilesX;
    f.co_blocks = ceil_div(co_t, WCO); f.ci_blocks = ceil_div(ci_t, WCI);
    f.gd = WCO * 2; f.gx = WCI * 2;
    f.PW = (f.TW - 1) * f.s + f.TB;
    f.PHX = (f.TH - 1) * f.s + TAA;
    f.pitch_d = f.gd * 16 + 16; f.pitch_x = f.gx * 16 + 16;
    const int nxp = ceil_div(f.PHX * f.PW * f.gx, 64);
    if (ceil_div(32 * f.gd, 64) > 3 || nxp > 7 || f.PW > 255) return MFC_ERR_UNSUPPORTED;
    const size_t ds = ((size_t)32 * f.pitch_d + 15) & ~(size_t)15, xs = ((size_t)f.PHX * f.PW * f.pitch_x + 15) & ~(size_t)15;
    f.off_x = (int)ds; f.buf_bytes = (int)(ds + xs) + 16;      // + a 16-byte dummy slot (dead staging pieces)
    f.wave_bytes = 2 * f.buf_bytes;
    size_t stage = (size_t)4 * f.wave_bytes;
    const size_t need = (size_t)2 * TAA * d->TB * WCO * WCI * 1024;      // tree reduction scratch (2 dumping waves)
    if (stage < need) stage = need;
    f.off_tab = (int)stage; f.off_coef = f.off_tab + 32 * 4;
    bool any_coef = d->in_coef != nullptr;
    for (int i = 0; batch && i < nbatch; ++i) any_coef = any_coef || batch[i].in_coef != nullptr;
    const size_t lds = (size_t)f.off_coef + (any_coef ? (size_t)f.G * 2 * f.gx * 8 * 4 : 0);
    if (lds > 80 * 1024) return MFC_ERR_UNSUPPORTED;
    const int Y = (f.TA / TAA) * f.co_blocks * f.ci_blocks;
    int S = d->splits;
    const int nb = d->batch > 1 ? d->batch : 1;      // problems sharing the launch: each gets 1/nb of the workgroups
    if (nb > MFC_WGRAD_MAXBATCH || (batch && nbatch != nb)) return MFC_ERR_INVALID_ARG;
    if (S <= 0) {
        S = ceil_div(g_wgrad_blocks, Y * nb);
        // few output tiles over a huge pixel count (temporal head at full resolution, stem): one workgroup per CU would walk tens of
        // thousands of pixels with 1-9 MFMAs per staged strip; split the pixel axis further (up to 8 workgroups per CU)
        if (g_wgrad_maxpx > 0) {
            long s2 = ((long)f.N * f.Hout * f.Wout + g_wgrad_maxpx - 1) / g_wgrad_maxpx;
            const long cap = ceil_div(2048, Y * nb);
            if (s2 > cap) s2 = cap;
            if (s2 > S) S = (int)s2;
        }
    }
    if (S * 4 > f.ntiles) S = ceil_div(f.ntiles, 4);
    if (S < 1) S = 1;
    f.splits = S;
    f.slice = d->TA * d->TB * f.Co16 * f.Ci16;
    if (parts_only) { *parts_only = S; return MFC_OK; }
    f.ablate = g_wgrad_ablate;
    if (nb > 1 && !batch) return MFC_ERR_INVALID_ARG;          // a batched descriptor must come through mfc_conv2d_wgrad_batch
    WgradBatch wb;
    wb.n = nb;
    for (int i = 0; i < MFC_WGRAD_MAXBATCH; ++i) {
        const mfc_wgrad_desc* q = (batch && i < nb) ? &batch[i] : d;
        wb.x[i] = (const char*)q->x; wb.dy[i] = (const char*)q->dy; wb.dwp[i] = q->dwp; wb.coef[i] = q->in_coef; wb.relu[i] = q->in_relu;
    }
    // one workgroup per CU (grid <= 256): the prefetch-distance-2 variant (1 wave per SIMD, 512 registers)
    const bool deep = g_wgrad_deep && (long)f.splits * Y * nb <= 256;
    if (alltaps) return deep ? wgrad_wave_launch<3, 3, 2, 2, 4, 2>(f, wb, lds, Y, st) : wgrad_wave_launch<3, 3, 2, 2, 4, 1>(f, wb, lds, Y, st);
#define WGW(tb, a_, b_) if (d->TB == tb && WCO == a_ && WCI == b_) { \
        if (nxp <= 4) return deep ? wgrad_wave_launch<1, tb, a_, b_, 4, 2>(f, wb, lds, Y, st) : wgrad_wave_launch<1, tb, a_, b_, 4, 1>(f, wb, lds, Y, st); \
        return deep ? wgrad_wave_launch<1, tb, a_, b_, 7, 2>(f, wb, lds, Y, st) : wgrad_wave_launch<1, tb, a_, b_, 7, 1>(f, wb, lds, Y, st); }
    WGW(3, 3, 3) WGW(3, 3, 2) WGW(3, 3, 1) WGW(3, 2, 3) WGW(3, 2, 2) WGW(3, 2, 1) WGW(3, 1, 3) WGW(3, 1, 2) WGW(3, 1, 1)
    WGW(11, 1, 2) WGW(11, 1, 1)
#undef WGW
    return MFC_ERR_UNSUPPORTED;
}

// conv_wgrad_dma.hip
bool wgrad_dma_eligible(const mfc_wgrad_desc* d);
int wgrad_dma_parts(const mfc_wgrad_desc* d);
int wgrad_dma_launch(const mfc_wgrad_desc* d, hipStream_t st);
// conv_wgrad_dma48.hip
bool wgrad_dma48_eligible(const mfc_wgrad_desc* d);
int wgrad_dma48_parts(const mfc_wgrad_desc* d);
int wgrad_dma48_launch(const mfc_wgrad_desc* d, hipStream_t st);
// conv_wgrad_dma_s2.hip
bool wgrad_dma_s2_eligible(const mfc_wgrad_desc* d);
int wgrad_dma_s2_parts(const mfc_wgrad_desc* d);
int wgrad_dma_s2_launch(const mfc_wgrad_desc* d, hipStream_t st);
// wgrad_gemm1x1.hip
bool wgrad_gemm1x1_eligible(const mfc_wgrad_desc* d);
int wgrad_gemm1x1_parts(const mfc_wgrad_desc* d);
int wgrad_gemm1x1_launch(const mfc_wgrad_desc* d, hipStream_t st);

static int wgrad_any(const mfc_wgrad_desc* d, void* stream, int* parts_only) {
    if (!d || !d->x || !d->dy || !d->dwp) return MFC_ERR_INVALID_ARG;
    if (!mfc_dtype_ok(d->dtype)) return MFC_ERR_INVALID_ARG;
    if (d->Cin_p % 8 || d->Cout_p % 8 || d->Cin > d->Cin_p || d->Cout > d->Cout_p) return MFC_ERR_INVALID_ARG;
    if (d->N <= 0 || d->TA <= 0 || d->TB <= 0 || d->in_stride < 1 || d->images_per_group <= 0 || d->N % d->images_per_group) return MFC_ERR_INVALID_ARG;
    if (wgrad_gemm1x1_eligible(d)) {
        if (parts_only) { *parts_only = wgrad_gemm1x1_parts(d); return MFC_OK; }
        return wgrad_gemm1x1_launch(d, (hipStream_t)stream);
    }
    if (wgrad_dma_eligible(d)) {
        if (parts_only) { *parts_only = wgrad_dma_parts(d); return MFC_OK; }
        return wgrad_dma_launch(d, (hipStream_t)stream);
    }
    if (wgrad_dma48_eligible(d)) {
        if (parts_only) { *parts_only = wgrad_dma48_parts(d); return MFC_OK; }
        return wgrad_dma48_launch(d, (hipStream_t)stream);
    }
    if (wgrad_dma_s2_eligible(d)) {
        if (parts_only) { *parts_only = wgrad_dma_s2_parts(d); return MFC_OK; }
        return wgrad_dma_s2_launch(d, (hipStream_t)stream);
    }
    {
        int rcf = wgrad_wave(d, (hipStream_t)stream, parts_only);
        if (rcf != MFC_ERR_UNSUPPORTED) return rcf;
        if (d->batch > 1) return MFC_ERR_UNSUPPORTED;       // only the wave-private kernel takes batches
        rcf = wgrad_fast(d, (hipStream_t)stream, parts_only);
        if (rcf != MFC_ERR_UNSUPPORTED) return rcf;
    }
    const int esz = mfc_is16(d->dtype) ? 2 : 4;
    WgradK k;
    k.x = (const char*)d->x; k.dy = (const char*)d->dy; k.dwp = d->dwp; k.in_coef = d->in_coef;
    k.N = d->N; k.Hin = d->Hin; k.Win = d->Win; k.Cin_p = d->Cin_p; k.Hout = d->Hout; k.Wout = d->Wout; k.Cout_p = d->Cout_p;
    k.TA = d->TA; k.TB = d->TB; k.dh0 = d->dh0; k.dw0 = d->dw0; k.s = d->in_stride; k.in_relu = d->in_relu; k.ipg = d->images_per_group;
    k.TH = d->TH; k.TW = d->TW;
    if (k.TH <= 0 || k.TW <= 0) mfc_choose_tile_wg(d->Hout, d->Wout, k.TH, k.TW);
    if (k.TH * k.TW > 128) return MFC_ERR_INVALID_ARG;
    k.tilesY = ceil_div(d->Hout, k.TH); k.tilesX = ceil_div(d->Wout, k.TW);
    k.ntiles = k.N * k.tilesY * k.tilesX;
    k.Co16 = ceil_div(d->Cout, 16) * 16; k.Ci16 = ceil_div(d->Cin, 16) * 16;
    const int co_t = k.Co16 / 16, ci_t = k.Ci16 / 16;
    // accumulator budget: TB * NCO * NCI <= 112 tiles per workgroup (28 per wave)
    int NCO = co_t < 6 ? co_t : 6, NCI = ci_t < 6 ? ci_t : 6;
    while (k.TB * NCO * NCI > 112) { if (NCI >= NCO && NCI > 1) --NCI; else if (NCO > 1) --NCO; else return MFC_ERR_UNSUPPORTED; }
    // even out the blocks
    k.co_blocks = ceil_div(co_t, NCO); NCO = ceil_div(co_t, k.co_blocks);
    k.ci_blocks = ceil_div(ci_t, NCI); NCI = ceil_div(ci_t, k.ci_blocks);
    k.NCO = NCO; k.NCI = NCI;
    k.ntile_mm = k.TB * NCO * NCI; k.cnt = ceil_div(k.ntile_mm, 4);
    k.PW = (k.TW - 1) * k.s + k.TB;
    k.pitch_d = NCO * 16 * esz + 16; k.pitch_x = NCI * 16 * esz + 16;
    size_t ds = (size_t)128 * k.pitch_d, xs = (size_t)k.TH * k.PW * k.pitch_x;
    k.off_x = (int)ds; k.off_tab = (int)(ds + xs);
    const int TPW = k.cnt <= 8 ? 8 : (k.cnt <= 16 ? 16 : 28);
    size_t lds = ds + xs + (size_t)(256 + 4 * TPW * 4) * 4;
    if (lds > 160 * 1024) return MFC_ERR_UNSUPPORTED;
    const int Y = k.TA * k.co_blocks * k.ci_blocks;
    int S = d->splits;
    if (S <= 0) { S = ceil_div(1024, Y); }
    if (S > k.ntiles) S = k.ntiles;
    if (S < 1) S = 1;
    k.splits = S;
    k.slice = d->TA * d->TB * k.Co16 * k.Ci16;
    if (parts_only) { *parts_only = S; return MFC_OK; }
    hipStream_t st = (hipStream_t)stream;
    const bool tr = g_wgrad_use_tr != 0;
    if (mfc_is16(d->dtype)) {
        int rcw;
        if (tr) {
            if (TPW == 8) MFC_TYPED16(d->dtype, T_, rcw = wgrad_launch<T_, 8, true>(k, lds, Y, st));
            else if (TPW == 16) MFC_TYPED16(d->dtype, T_, rcw = wgrad_launch<T_, 16, true>(k, lds, Y, st));
            else MFC_TYPED16(d->dtype, T_, rcw = wgrad_launch<T_, 28, true>(k, lds, Y, st));
            return rcw;
        }
        if (TPW == 8) MFC_TYPED16(d->dtype, T_, rcw = wgrad_launch<T_, 8, false>(k, lds, Y, st));
        else if (TPW == 16) MFC_TYPED16(d->dtype, T_, rcw = wgrad_launch<T_, 16, false>(k, lds, Y, st));
        else MFC_TYPED16(d->dtype, T_, rcw = wgrad_launch<T_, 28, false>(k, lds, Y, st));
        return rcw;
    }
    if (TPW == 8) return wgrad_launch<float, 8, false>(k, lds, Y, st);
    if (TPW == 16) return wgrad_launch<float, 16, false>(k, lds, Y, st);
    return wgrad_launch<float, 28, false>(k, lds, Y, st);
}

extern "C" int mfc_conv2d_wgrad(const mfc_wgrad_desc* d, void* stream) {
    if (d && !mfc_ptrs_ok(d->x, d->dy, d->dwp, d->in_coef)) return MFC_ERR_INVALID_ARG;
    return wgrad_any(d, stream, nullptr);
}

extern "C" int mfc_conv2d_wgrad_parts(const mfc_wgrad_desc* d) {
    if (!d) return MFC_ERR_INVALID_ARG;
    mfc_wgrad_desc t = *d;                 // geometry only: placeholder pointers are accepted
    if (!t.x) t.x = (const void*)16;
    if (!t.dy) t.dy = (const void*)16;
    if (!t.dwp) t.dwp = (float*)16;
    int parts = 0;
    const int rc = wgrad_any(&t, nullptr, &parts);
    return rc < 0 ? rc : parts;
}

// n weight gradients of identical geometry (everything but x / dy / dwp / in_coef / in_relu) in one launch; every descriptor
// carries batch = n and the same `splits`.  MFC_ERR_UNSUPPORTED when the geometry is not served by the wave-private kernel.
extern "C" int mfc_conv2d_wgrad_batch(const mfc_wgrad_desc* descs, int32_t n, void* stream) {
    if (!descs || n < 1 || n > MFC_WGRAD_MAXBATCH) return MFC_ERR_INVALID_ARG;
    if (n == 1 && descs[0].batch <= 1) return wgrad_any(&descs[0], stream, nullptr);
    const mfc_wgrad_desc& a = descs[0];
    for (int i = 0; i < n; ++i) {
        const mfc_wgrad_desc& b = descs[i];
        if (!b.x || !b.dy || !b.dwp || b.batch != n) return MFC_ERR_INVALID_ARG;
        if (b.dtype != a.dtype || b.N != a.N || b.Hin != a.Hin || b.Win != a.Win || b.Cin_p != a.Cin_p || b.Cin != a.Cin ||
            b.Hout != a.Hout || b.Wout != a.Wout || b.Cout_p != a.Cout_p || b.Cout != a.Cout || b.TA != a.TA || b.TB != a.TB ||
            b.dh0 != a.dh0 || b.dw0 != a.dw0 || b.in_stride != a.in_stride || b.images_per_group != a.images_per_group ||
            b.splits != a.splits)
            return MFC_ERR_INVALID_ARG;
    }
    if (!mfc_is16(a.dtype)) return MFC_ERR_UNSUPPORTED;
    if (a.Cin_p % 8 || a.Cout_p % 8 || a.Cin > a.Cin_p || a.Cout > a.Cout_p || a.N <= 0 || a.images_per_group <= 0 || a.N % a.images_per_group) return MFC_ERR_INVALID_ARG;
    return wgrad_wave(&a, (hipStream_t)stream, nullptr, descs, n);
}
