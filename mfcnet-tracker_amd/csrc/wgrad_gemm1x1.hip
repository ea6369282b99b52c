// libmfcnet_hip: weight gradient of the big 1x1 convolutions as a split-K GEMM (bf16, gfx950).
//
// dW[co][ci] = sum_p dY[p][co] * X[p][ci] for hrnet.py:334-351 `last_layer[0]` (480 x 480 over 460 800 pixels, 720 x 720 for W48).
// In the general 1x1 path (conv_wgrad_fast_kernel) a workgroup owns a 96 x 96 block of dW, so both operands are read five
// (W48: eight) times and the launch runs at 130 TFLOP/s, 1.6 ms -- the longest record of the detached stream.  Here:
//   * workgroup = 8 waves, block 256 co x 256 ci, wave = 128 co x 64 ci = 8 x 4 MFMA tiles (v_mfma_f32_16x16x32_bf16);
//   * K = pixels, 64 per stage; both operand tiles ([64 px][256 channels], 32 KiB each) go global -> LDS by DMA in their natural
//     pixel-major layout, double buffered.  The MFMA operands need the pixel axis in-lane: ds_read_b64_tr_b16 (hardware transpose
//     read) as in conv_wgrad.hip.  A pixel row is exactly 512 bytes, so without care the 16 rows a fragment read touches would all
//     sit on the same banks; the 32-byte channel blocks of row p are therefore stored at position (block ^ (p & 15)) -- the DMA
//     writes lane-contiguous LDS, so the permutation is applied to the SOURCE granule each lane fetches;
//   * split-K over the pixel axis: ~256 workgroups, each writes its partial block into its own slice of the partial-sum buffer
//     (mfc_conv2d_wgrad_parts), which the ordinary unpack launch adds up in a fixed order (deterministic, no atomics).
//   * the fused input transform (BatchNorm + ReLU of the producer layer) is applied to the X tile in place in LDS after it landed.
// Dispatched from wgrad_any (conv_wgrad.hip) when wgrad_gemm1x1_eligible().
#include "common.h"

struct WgG {
    const char* x; const char* dy; float* dwp; const float* in_coef;
    int in_relu, px_per_group, G;
    int M, Cin_p, Cin_g, Cout_p, Cout_g, Co16, Ci16;
    int co_blocks, ci_blocks, S, nstages, slice;
};

constexpr int WG_PX = 64, WG_C = 256;                            // pixels per stage, channels per block side
constexpr int WG_TILE = WG_PX * WG_C * 2;                        // 32 KiB
constexpr int WG_OFF_D0 = 0, WG_OFF_D1 = WG_TILE, WG_OFF_X0 = 2 * WG_TILE, WG_OFF_X1 = 3 * WG_TILE;
constexpr int WG_OFF_COEF = 4 * WG_TILE;                         // float [G <= 8][2][256]: scale / shift of the block's input channels
constexpr int WG_LDS = 4 * WG_TILE + 8 * 2 * WG_C * 4;

template <typename TE>
__global__ __launch_bounds__(512, 1) void wgrad_gemm1x1_kernel(WgG p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int nblk = p.co_blocks * p.ci_blocks;
    // the co x ci blocks of one pixel range read the same two operand tiles: keep them on one XCD (one L2)
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int split = Lb / nblk, rem = Lb - split * nblk;
    const int cb = rem / p.ci_blocks, ib = rem - cb * p.ci_blocks;
    const int s0 = (int)(((long)split * p.nstages) / p.S), s1 = (int)(((long)(split + 1) * p.nstages) / p.S);

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma1k = [&](const char* g, unsigned ldst) {
        unsigned keep;
        ldst = __builtin_amdgcn_readfirstlane(ldst);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(ldst) : "memory");
    };
    auto dma_stage = [&](int s, int par) {
        const unsigned ld = lds0 + (par ? WG_OFF_D1 : WG_OFF_D0), lx = lds0 + (par ? WG_OFF_X1 : WG_OFF_X0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave * 4 + i;                         // 1-KiB piece = pixel rows 2q, 2q+1
            const int pp = 2 * q + (lane >> 5);
            const int g = (lane & 31) ^ (2 * (pp & 15));        // granule (8 channels) stored at position lane & 31 of the row
            const size_t pix = (size_t)s * WG_PX + pp;
            dma1k(p.dy + (pix * p.Cout_p + (size_t)min(cb * 32 + g, p.Cout_g - 1) * 8) * 2, ld + q * 1024);
            dma1k(p.x + (pix * p.Cin_p + (size_t)min(ib * 32 + g, p.Cin_g - 1) * 8) * 2, lx + q * 1024);
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // wave-uniform counts of 16-channel tiles that hold real rows / columns of dW
    const int niv = min(8, (p.Co16 - (cb * WG_C + wm * 128) + 15) >> 4);
    const int njv = min(4, (p.Ci16 - (ib * WG_C + wn * 64) + 15) >> 4);
    const bool work = niv > 0 && njv > 0;

    // fragment addressing (see conv_wgrad.hip): the lane reads 8 bytes (4 channels) of pixel rows pr and pr + 4; after the transpose
    // lane (lane & 15) owns one channel of the 16-channel block and 8 pixels of the k-step
    const int sub = (lane & 15) >> 2, hi16 = lane >> 4, c8 = (lane & 3) * 8;
    int rowoff[2][2], key[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pr = ks * 32 + 8 * hi16 + sub + 4 * h;
            rowoff[ks][h] = pr * (WG_C * 2) + c8;
            key[ks][h] = pr & 15;
        }

    // fused input transform x' = relu?(x * scale[g, ci] + shift[g, ci]) (the producer's BatchNorm + ReLU, as the general kernels apply
    // while staging): the tile arrives raw by DMA, so it is transformed in place in LDS, 4 granules per thread and stage
    float* cfl = (float*)(smem + WG_OFF_COEF);
    if (p.in_coef) {
        for (int i = tid; i < p.G * 2 * WG_C; i += 512) {
            const int g = i / (2 * WG_C), r = i - g * 2 * WG_C, which = r / WG_C, c = r - which * WG_C;
            const int ch = ib * WG_C + c;
            cfl[i] = ch < p.Cin_p ? p.in_coef[((size_t)g * 4 + which) * p.Cin_p + ch] : 0.f;
        }
    }
    auto transform = [&](int s, int par) {
        char* X = smem + (par ? WG_OFF_X1 : WG_OFF_X0);
        const int grp = (int)(((long)s * WG_PX) / p.px_per_group);
        const float* sc = cfl + grp * 2 * WG_C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + k * 512;                      // granule slot of the tile: row idx >> 5, position idx & 31
            const int pp = idx >> 5, g = (idx & 31) ^ (2 * (pp & 15));
            uint4 v = *(const uint4*)(X + idx * 16);
            float f[8];
            Gran<TE>::unpack(v, f);
            const float4 s0 = *(const float4*)(sc + g * 8), s1 = *(const float4*)(sc + g * 8 + 4);
            const float4 h0 = *(const float4*)(sc + WG_C + g * 8), h1 = *(const float4*)(sc + WG_C + g * 8 + 4);
            const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = f[e] * scv[e] + shv[e];
                f[e] = p.in_relu ? relu_nan(t) : t;
            }
            *(uint4*)(X + idx * 16) = Gran<TE>::pack(f);
        }
    };

    if (s0 < s1) { dma_stage(s0, 0); }
    dma_wait();
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const int par = (s - s0) & 1;
        if (s + 1 < s1) dma_stage(s + 1, par ^ 1);
        if (p.in_coef) { transform(s, par); __syncthreads(); }
        if (work) {
            const char* D = smem + (par ? WG_OFF_D1 : WG_OFF_D0);
            const char* X = smem + (par ? WG_OFF_X1 : WG_OFF_X0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[8], bfr[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int blk = wm * 8 + i;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(D + rowoff[ks][0] + ((blk ^ key[ks][0]) << 5)));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(D + rowoff[ks][1] + ((blk ^ key[ks][1]) << 5)));
                    af[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int blk = wn * 4 + j;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(X + rowoff[ks][0] + ((blk ^ key[ks][0]) << 5)));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(X + rowoff[ks][1] + ((blk ^ key[ks][1]) << 5)));
                    bfr[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < njv) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (i < niv) acc[i][j] = mfma16<TE>(af[i], bfr[j], acc[i][j]);
                    }
            }
        }
        dma_wait();
        __syncthreads();
    }
    // ---- store the partial block: D[row = co][col = ci]; lane holds col = lane & 15, rows 4*(lane>>4) + r ----
    float* slice = p.dwp + (size_t)split * p.slice;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = cb * WG_C + wm * 128 + i * 16 + (lane >> 4) * 4, ci = ib * WG_C + wn * 64 + j * 16 + (lane & 15);
            if (co < p.Co16 && ci < p.Ci16) {
                float* o = slice + (size_t)co * p.Ci16 + ci;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(size_t)r * p.Ci16] = acc[i][j][r];
            }
        }
}

int g_wgrad_gemm_minc = 64;          // smallest Cin / Cout sent there; tuning: mfc_set_flag(26, n)
int g_wgrad_gemm = 1;                // big 1x1 weight gradients through wgrad_gemm1x1_kernel; tuning: mfc_set_flag(25, v)

bool wgrad_gemm1x1_eligible(const mfc_wgrad_desc* d) {
    if (!g_wgrad_gemm || !mfc_is16(d->dtype)) return false;
    if (d->TA != 1 || d->TB != 1 || d->in_stride != 1 || d->dh0 != 0 || d->dw0 != 0) return false;
    if (d->Hin != d->Hout || d->Win != d->Wout || d->batch > 1) return false;
    if (d->in_coef) {      // a 64-pixel stage never straddles a statistics group
        if (d->images_per_group <= 0 || d->N % d->images_per_group || d->N / d->images_per_group > 8) return false;
        if (((long)d->images_per_group * d->Hout * d->Wout) % WG_PX) return false;
    }
    if (d->Cin < g_wgrad_gemm_minc || d->Cout < g_wgrad_gemm_minc || d->Cin_p % 8 || d->Cout_p % 8) return false;
    const long M = (long)d->N * d->Hout * d->Wout;
    if (M % WG_PX || M / WG_PX < 16 || M / WG_PX > 0x3fffffff) return false;
    return true;
}

static void wgrad_gemm1x1_setup(const mfc_wgrad_desc* d, WgG& k) {
    k.x = (const char*)d->x; k.dy = (const char*)d->dy; k.dwp = d->dwp; k.in_coef = d->in_coef; k.in_relu = d->in_relu;
    k.G = d->in_coef ? d->N / d->images_per_group : 1;
    k.px_per_group = d->in_coef ? d->images_per_group * d->Hout * d->Wout : d->N * d->Hout * d->Wout;
    k.M = d->N * d->Hout * d->Wout; k.Cin_p = d->Cin_p; k.Cin_g = d->Cin_p / 8; k.Cout_p = d->Cout_p; k.Cout_g = d->Cout_p / 8;
    k.Co16 = ceil_div(d->Cout, 16) * 16; k.Ci16 = ceil_div(d->Cin, 16) * 16;
    k.co_blocks = ceil_div(k.Co16, WG_C); k.ci_blocks = ceil_div(k.Ci16, WG_C);
    k.nstages = k.M / WG_PX;
    int S = d->splits;
    if (S <= 0) S = 256 / (k.co_blocks * k.ci_blocks);            // one workgroup per CU, never more than 256 (a 257th would run alone in a second round)
    if (S > k.nstages / 8) S = k.nstages / 8;                     // (a split walks at least 8 stages)
    if (S < 1) S = 1;
    k.S = S;
    k.slice = k.Co16 * k.Ci16;
}

int wgrad_gemm1x1_parts(const mfc_wgrad_desc* d) { WgG k; wgrad_gemm1x1_setup(d, k); return k.S; }

int wgrad_gemm1x1_launch(const mfc_wgrad_desc* d, hipStream_t st) {
    WgG k; wgrad_gemm1x1_setup(d, k);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_gemm1x1_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_gemm1x1_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        const double flops = 2.0 * k.M * (double)k.Co16 * k.Ci16;
        const double bytes = (double)k.M * (k.Cin_p + k.Cout_p) * 2.0;
        mfc_prof_before(st, d->dtype == MFC_F16 ? "wgrad_gemm1x1_kernel<_Float16>" : "wgrad_gemm1x1_kernel<__bf16>", flops, bytes);
    }
    MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL(wgrad_gemm1x1_kernel<T_>, dim3(k.co_blocks * k.ci_blocks * k.S), dim3(512), WG_LDS, st, k));
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}
