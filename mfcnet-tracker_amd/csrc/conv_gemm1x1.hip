// libmfcnet_hip: the big 1x1 convolutions as a plain GEMM (bf16, gfx950).
//
// hrnet.py:334-351 `last_layer[0]` (480 -> 480 channels for W32, 720 -> 720 for W48, at 120x160 with all T*B images) and its
// data gradient are 212 / 478 GFLOP each and sit after the last join of the branch lanes, i.e. on the critical chain of the step.
// In the general kernel (conv_igemm.hip) a wave owns 32 pixels x 96 channels and the input tile goes through registers in
// 64-channel chunks: 12 MFMAs per 8 LDS fragment reads (LDS-bandwidth bound, ~220-280 TFLOP/s) and the input is re-read once per
// 96-channel block.  A 1x1 / stride-1 convolution has no halo, so here it runs as Y[M][Cout] = X[M][Cin] . W^T:
//   * workgroup = 8 waves, tile 256 pixels x 256 output channels, K chunks of 64 channels; wave (wm, wn) owns 128 pixels x 64
//     channels = 8 x 4 MFMA tiles (v_mfma_f32_16x16x32_bf16): 32 MFMAs per 12 fragment reads;
//   * both operands go global -> LDS by DMA (global_load_lds_dwordx4), no registers in the staging path; a pipeline stage is one
//     k-step (32 channels: 16 KiB + 16 KiB) and four stages are resident, so the DMA runs three stages ahead of the MFMAs; the
//     weights are the packed image of mfc_pack_weights ([chunk][cout block][granule][256 couts], one 32 KiB stage = one LDS
//     image, its halves one stage each), the activations land as [pixel][granule ^ ((pixel >> 1) & 3)] in 64-byte rows (XOR swizzle: the DMA writes lane-contiguous LDS, so the
//     swizzle is applied to the SOURCE granule each lane fetches -- free), which makes the fragment reads conflict free;
//   * persistent workgroups over contiguous pixel-tile ranges; the input is read once per 256-channel block;
//   * the fused input transform (BatchNorm + ReLU of the producer) is applied to the activation tile in place in LDS after it landed;
//   * epilogue as in conv_igemm: + bias, per-(group, channel) sum / sum of squares, bf16 rows transposed across the four 16-lane
//     groups (v_permlane32_swap / v_permlane16_swap) so that every lane stores 16 contiguous bytes.
// Dispatched from mfc_conv2d_fwd / mfc_conv2d_layout (conv_igemm.hip) when gemm1x1_eligible(); everything else, and the
// fp32 parity mode, keeps the general kernel.
#include "common.h"

struct GemmK {
    const char* in; const char* wp; char* out; const float* bias; mfc_stat_t* out_stats; const float* in_coef;
    int in_relu;
    int M, Cin_p, Cin_g, Cout_p, Cout;
    int nchunks, Yblocks, ntiles, tiles_per_block;
    int px_per_group, G, accumulate;
    int wt;                              // write-through output stores (common.h: large outputs only)
    // data-gradient epilogue fusions (mfc_conv_desc.acc_src / bn_y; FUSE instantiation only), as in conv_igemm.hip
    const char* acc_src; const char* bn_y; const float* bn_coef; const unsigned char* bn_bits; int bn_mode;
};

template <int CTRL> __device__ inline float g_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ inline float g_row16_sum(float v) {      // sum over the 16 lanes of a DPP row (result in every lane)
    v = g_dpp_add<0xB1>(v); v = g_dpp_add<0x4E>(v); v = g_dpp_add<0x141>(v); v = g_dpp_add<0x140>(v);
    return v;
}

constexpr int G_BM = 256, G_BN = 256, G_KG = 8;                 // pixels, couts per tile; granules (of 8 channels) per packed weight chunk
constexpr int G_BBYTES = G_KG * G_BN * 16;                       // one packed weight chunk image: 32 KiB
// a pipeline stage is HALF a chunk (4 granules = 32 channels = one MFMA k-step): 16 KiB of activations + 16 KiB of weights, and FOUR
// stages are resident, so the DMA of stage s+3 is issued while stage s computes (one DMA round trip is ~3 us, a k-step 0.4 us)
constexpr int G_HG = 4, G_NBUF = 4;
constexpr int G_AH = G_BM * G_HG * 16, G_BH = G_HG * G_BN * 16;  // 16 KiB each
constexpr int G_OFF_A = 0, G_OFF_B = G_NBUF * G_AH;
constexpr int G_MAXYB = 4;                                       // cout blocks whose statistics are kept in LDS side by side
constexpr int G_RED1 = 8 * 2 * 64;                               // floats of one block's sums: [8 waves][2][64]
constexpr int G_OFF_RED = G_NBUF * (G_AH + G_BH);                // float [G_MAXYB][8 waves][2][64]
constexpr int G_OFF_COEF = G_OFF_RED + G_MAXYB * G_RED1 * 4;     // float [G][2][Cin_p]: scale / shift of the fused input transform
constexpr int G_COEF_FLOATS = 3072;
constexpr int G_OFF_BIAS = G_OFF_COEF + G_COEF_FLOATS * 4;       // float [Cout <= 1024] (LDS: an epilogue load from global would drain the DMA pipeline)
constexpr int G_BIAS_FLOATS = 1024;
constexpr int G_LDS = G_OFF_BIAS + G_BIAS_FLOATS * 4;            // = 160 KiB

// FUSE: the data-gradient epilogue fusions are compiled in (a separate instantiation: the plain one keeps its registers)
template <typename TE, bool FUSE>
__global__ __launch_bounds__(512, 1) void conv_gemm1x1_kernel(GemmK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)(smem + G_OFF_RED);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int t0 = Lb * p.tiles_per_block;
    const int ntl = min(p.tiles_per_block, p.ntiles - t0);
    if (ntl <= 0) return;
    // cout block fastest: the Yblocks units of a pixel tile run back to back and the re-reads of the tile hit L2.  The statistics of
    // up to G_MAXYB blocks are kept side by side in LDS across tiles; with more blocks the order flips (block slowest)
    const bool ybfast = (p.out_stats == nullptr) || p.Yblocks <= G_MAXYB;
    const int nunits = ntl * p.Yblocks;
    const int T = nunits * p.nchunks * 2;                    // pipeline stages (half chunks)
    const int cq = (lane >> 4) * 4;

    auto unit_of = [&](int u, int& t, int& yb) {
        if (ybfast) { t = t0 + u / p.Yblocks; yb = u - (u / p.Yblocks) * p.Yblocks; }
        else { yb = u / ntl; t = t0 + (u - yb * ntl); }
    };
    // DMA of stage (tile t, chunk c, cout block yb) into buffer `par`
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma1k = [&](const char* g, unsigned ldst) {
        unsigned keep;
        ldst = __builtin_amdgcn_readfirstlane(ldst);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(ldst) : "memory");
    };
    auto dma_stage = [&](int t, int c, int h, int yb, int buf) {
        const unsigned la = lds0 + G_OFF_A + buf * G_AH, lb = lds0 + G_OFF_B + buf * G_BH;
        const char* wsrc = p.wp + (size_t)(c * p.Yblocks + yb) * G_BBYTES + h * G_BH + lane * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = wave * 2 + i;                     // 1-KiB piece: pixels q*16 .. q*16+15, the 4 granule slots of the half chunk
            const int pl = q * 16 + (lane >> 2);
            const int gi = min(c * G_KG + h * G_HG + ((lane & 3) ^ ((pl >> 1) & 3)), p.Cin_g - 1);      // (tail: a valid granule; its weights are zero)
            dma1k(p.in + ((size_t)t * G_BM + pl) * p.Cin_p * 2 + (size_t)gi * 16, la + q * 1024);
            dma1k(wsrc + q * 1024, lb + q * 1024);
        }
    };
    // wait until at most k later stages (4 DMA instructions per wave and stage) are still in flight
    auto dma_wait = [&](int k) {
        if (k >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (k == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment addressing: A (pixels) [pl][slot ^ ((pl >> 1) & 3)] with 64-byte rows, pl = wm*128 + mt*16 + (lane & 15); B (couts) [slot][256 couts]
    const int a_off = (wm * 128 + (lane & 15)) * 64 + (((lane >> 4) ^ (((lane & 15) >> 1) & 3)) * 16);
    const int b_off = ((lane >> 4) * G_BN + wn * 64 + (lane & 15)) * 16;

    for (int i = tid; i < G_MAXYB * G_RED1; i += 512) red[i] = 0.f;
    float* biasl = (float*)(smem + G_OFF_BIAS);
    for (int i = tid; i < G_BIAS_FLOATS; i += 512) biasl[i] = (p.bias && i < p.Cout) ? p.bias[i] : 0.f;
    int t, yb;
    unit_of(0, t, yb);
    // issue cursor: runs G_NBUF-1 stages ahead of the compute cursor
    int iu = 0, ic = 0, ih = 0, it = t, iyb = yb, issued = 0;
    auto issue_next = [&]() {
        dma_stage(it, ic, ih, iyb, issued & (G_NBUF - 1));
        ++issued;
        if (++ih == 2) { ih = 0; if (++ic == p.nchunks) { ic = 0; ++iu; if (iu < nunits) unit_of(iu, it, iyb); } }
    };
    for (int k = 0; k < G_NBUF - 1 && issued < T; ++k) issue_next();
    dma_wait(issued - 1);
    __syncthreads();

    int u = 0, c = 0, h = 0;
    int red_grp = -1, red_yb = -1; bool red_live = false;
    auto stats_flush = [&]() {          // every wave has added its row sums to red[yb][wave]; sum the two pixel halves, publish, clear
        __syncthreads();
        const int y0 = ybfast ? 0 : red_yb, y1 = ybfast ? p.Yblocks : red_yb + 1;
        for (int y = y0; y < y1; ++y) {
            const float* rd = red + (ybfast ? y : 0) * G_RED1;
            const int which = tid >> 8, cl = tid & 255;
            const int w0 = (cl >> 6) * 2, ch = cl & 63;
            const float s = rd[(w0 * 2 + which) * 64 + ch] + rd[((w0 + 1) * 2 + which) * 64 + ch];
            const int co = y * G_BN + cl;
            if (co < p.Cout_p) atomicAdd(p.out_stats + (((size_t)(Lb % MFC_R) * p.G + red_grp) * 2 + which) * p.Cout_p + co, (mfc_stat_t)s);
        }
        __syncthreads();
        for (int i = tid; i < G_MAXYB * G_RED1; i += 512) red[i] = 0.f;
        __syncthreads();
        red_live = false;
    };

    // fused input transform x' = relu?(x * scale[g, ci] + shift[g, ci]) (the producer's BatchNorm + ReLU): the tile arrives raw by DMA
    // and is transformed in place in LDS, 4 granules per thread and stage
    float* cfl = (float*)(smem + G_OFF_COEF);
    if (p.in_coef) {
        for (int i = tid; i < p.G * 2 * p.Cin_p; i += 512) {
            const int g = i / (2 * p.Cin_p), r = i - g * 2 * p.Cin_p;
            cfl[i] = p.in_coef[(size_t)g * 4 * p.Cin_p + r];          // rows 0 (scale) and 1 (shift) of [G][4][Cin_p]
        }
        __syncthreads();
    }
    if constexpr (FUSE) {
        if (p.bn_y) {      // (gemm1x1_launch: fits, and excludes in_coef)
            for (int i = tid; i < p.G * 4 * p.Cout_p; i += 512) cfl[i] = p.bn_coef[i];
            __syncthreads();
        }
    }
    auto transform = [&](int buf, int tt, int cc, int hh) {
        char* At = smem + G_OFF_A + buf * G_AH;
        const float* sc = cfl + (size_t)(((long)tt * G_BM) / p.px_per_group) * 2 * p.Cin_p;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + k * 512;                      // granule slot: pixel idx >> 2, position idx & 3
            const int pl = idx >> 2, cg = cc * G_KG + hh * G_HG + ((idx & 3) ^ ((pl >> 1) & 3));
            if (cg < p.Cin_g) {
                float f[8];
                Gran<TE>::unpack(*(const uint4*)(At + idx * 16), f);
                const float4 s0 = *(const float4*)(sc + cg * 8), s1 = *(const float4*)(sc + cg * 8 + 4);
                const float4 h0 = *(const float4*)(sc + p.Cin_p + cg * 8), h1 = *(const float4*)(sc + p.Cin_p + cg * 8 + 4);
                const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float tv = f[e] * scv[e] + shv[e];
                    f[e] = p.in_relu ? relu_nan(tv) : tv;
                }
                *(uint4*)(At + idx * 16) = Gran<TE>::pack(f);
            }
        }
    };

    for (int s = 0; s < T; ++s) {
        if (issued < T) issue_next();                       // stage s + 3 into the buffer stage s - 1 has just left
        const int buf = s & (G_NBUF - 1);
        if (p.in_coef) { transform(buf, t, c, h); __syncthreads(); }

        // ---------------- compute: one k-step of 32 channels ----------------
        const char* A = smem + G_OFF_A + buf * G_AH + a_off;
        const char* B = smem + G_OFF_B + buf * G_BH + b_off;
        const int n0 = yb * G_BN + wn * 64;
        const int ntv = min(4, (p.Cout - n0 + 15) >> 4);         // cout tiles of this wave that hold real channels (wave-uniform)
        if (ntv > 0) {
            bf16x8 x[8], w[4];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) x[mt] = *(const bf16x8*)(A + mt * 16 * 64);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) w[nt] = *(const bf16x8*)(B + nt * 256);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (nt < ntv) {
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) acc[mt][nt] = mfma16<TE>(w[nt], x[mt], acc[mt][nt]);
                }
        }
        // stage s + 1 has landed (this wave's pieces; the later ones may still be in flight), then everybody's
        dma_wait(issued - s - 2);
        __syncthreads();

        if (h == 1 && c == p.nchunks - 1) {
            // ---------------- epilogue of unit (t, yb) ----------------
            const int grp = p.out_stats ? (int)(((long)t * G_BM) / p.px_per_group) : 0;
            if (p.out_stats && red_live && (grp != red_grp || (!ybfast && yb != red_yb))) stats_flush();
            float bq[4][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = n0 + nt * 16 + cq + r;
                    bq[nt][r] = biasl[min(co, G_BIAS_FLOATS - 1)];          // (zero beyond Cout)
                }
            float ssum[4][4], ssq[4][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }
            const size_t ooff = (((size_t)t * G_BM + wm * 128 + (lane & 15)) * p.Cout_p + n0 + (lane >> 4) * 8) * 2;
            char* obase = p.out + ooff;
            const char* abase = ((FUSE && p.acc_src) ? p.acc_src : (const char*)p.out) + ooff;      // where the running sum of an accumulating launch lives
            const bool bnm = FUSE && p.bn_y != nullptr;                // fused BatchNorm / ReLU backward of the tensor this launch completes (wave-uniform)
            // transposed-layout sums of the fused form: lane (row g = lane >> 4) owns channels n0 + pr*32 + 8g .. +7
            // (the BatchNorm's coefficient block [G][4][Cout_p] sits in LDS: the region of the input-transform coefficients, which a
            //  data gradient does not have)
            float tsum[2][8], tsq[2][8];
            int bco[2] = {0, 0};
            if constexpr (FUSE) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { tsum[pr][e] = 0.f; tsq[pr][e] = 0.f; }
                    bco[pr] = grp * 4 * p.Cout_p + min(n0 + pr * 32 + (lane >> 4) * 8, p.Cout_p - 8);
                }
            }
            // accumulate: the old values (and, fused, the pre-BatchNorm tensor and the mask bytes) are requested four pixel tiles (8 stores)
            // ahead, branch-free with a clamped address, so their latency is paid twice per unit and not once per store
            constexpr int PF = FUSE ? 2 : 4;          // (the fused form prefetches three tensors: two tiles ahead keeps it out of scratch)
            uint4 oldv[PF][2], pf_y[FUSE ? PF : 1][2]; unsigned pf_b[FUSE ? PF : 1][2];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                if ((p.accumulate || bnm) && (mt & (PF - 1)) == 0) {
#pragma unroll
                    for (int m2 = 0; m2 < PF; ++m2)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            const bool vc0 = n0 + pr * 32 + (lane >> 4) * 8 < p.Cout_p;
                            const size_t off = (size_t)(mt + m2) * 16 * p.Cout_p * 2 + (vc0 ? pr * 64 : 0);
                            oldv[m2][pr] = p.accumulate ? *(const uint4*)(abase + off) : make_uint4(0, 0, 0, 0);
                            if constexpr (FUSE) {
                                if (bnm) {
                                    pf_y[m2][pr] = *(const uint4*)(p.bn_y + ooff + off);
                                    if (p.bn_mode == 3) pf_b[m2][pr] = p.bn_bits[(ooff + off) >> 4];
                                }
                            }
                        }
                }
                float v[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (FUSE) v[nt][r] = acc[mt][nt][r]; else v[nt][r] = acc[mt][nt][r] + bq[nt][r];          // (no bias with the fusions)
                        acc[mt][nt][r] = 0.f;
                        if constexpr (!FUSE) { ssum[nt][r] += v[nt][r]; ssq[nt][r] += v[nt][r] * v[nt][r]; }
                    }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    // after the transpose lane (row g = lane>>4) owns channels n0 + pr*32 + 8g .. +7 of its pixel
                    char* oaddr = obase + (size_t)mt * 16 * p.Cout_p * 2 + pr * 64;
                    const bool vc = n0 + pr * 32 + (lane >> 4) * 8 < p.Cout_p;
                    if (p.accumulate || bnm) {
                        // data gradient added to an existing one: in fp32 on the transposed layout (raw dwords transposed), rounded once
                        const uint4 oldq = oldv[mt & (PF - 1)][pr];
                        unsigned t0[4], t1[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            auto x32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * pr][r]), __float_as_uint(v[2 * pr + 1][r]), false, false);
                            auto x16 = __builtin_amdgcn_permlane16_swap(x32[0], x32[1], false, false);
                            t0[r] = x16[0]; t1[r] = x16[1];
                        }
                        float o8[8], w8[8];
                        Gran<TE>::unpack(oldq, o8);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { w8[r] = __uint_as_float(t0[r]) + o8[r]; w8[4 + r] = __uint_as_float(t1[r]) + o8[4 + r]; }
                        if constexpr (FUSE) {
                            if (bnm) {
                                // mask, store the MASKED gradient, keep sum g*m and sum g*m*yhat of the lane's 8 channels (what mfc_bnbwd_reduce
                                // would sweep the tensor for again)
                                float yv[8];
                                Gran<TE>::unpack(pf_y[mt & (PF - 1)][pr], yv);
                                const float* cf = cfl + bco[pr];
                                const float4 m0 = *(const float4*)(cf + 2 * p.Cout_p), m1 = *(const float4*)(cf + 2 * p.Cout_p + 4);
                                const float4 r0 = *(const float4*)(cf + 3 * p.Cout_p), r1 = *(const float4*)(cf + 3 * p.Cout_p + 4);
                                const float bmean[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
                                const float brstd[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
                                unsigned mk = 0xffu;
                                if (p.bn_mode == 2) {
                                    const float4 s0 = *(const float4*)cf, s1 = *(const float4*)(cf + 4);
                                    const float4 h0 = *(const float4*)(cf + p.Cout_p), h1 = *(const float4*)(cf + p.Cout_p + 4);
                                    const float bsc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, bsh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                                    mk = 0;
#pragma unroll
                                    for (int e = 0; e < 8; ++e) mk |= ((yv[e] * bsc[e] + bsh[e]) > 0.f ? 1u : 0u) << e;
                                } else if (p.bn_mode == 3) {
                                    mk = pf_b[mt & (PF - 1)][pr];             // one byte per 8-channel granule (mfc_combine_fwd)
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const float gmv = ((mk >> e) & 1u) ? w8[e] : 0.f;
                                    w8[e] = gmv;
                                    if (vc) { tsum[pr][e] += gmv; tsq[pr][e] += gmv * ((yv[e] - bmean[e]) * brstd[e]); }
                                }
                            }
                        }
                        if (vc) mfc_st16_if(oaddr, Gran<TE>::pack(w8), p.wt);
                    } else {
                        const unsigned p0 = pack2<TE>(v[2 * pr][0], v[2 * pr][1]), p1 = pack2<TE>(v[2 * pr][2], v[2 * pr][3]);
                        const unsigned q0 = pack2<TE>(v[2 * pr + 1][0], v[2 * pr + 1][1]), q1 = pack2<TE>(v[2 * pr + 1][2], v[2 * pr + 1][3]);
                        auto a32 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                        auto a16 = __builtin_amdgcn_permlane16_swap(a32[0], a32[1], false, false);
                        auto b32 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                        auto b16 = __builtin_amdgcn_permlane16_swap(b32[0], b32[1], false, false);
                        if (vc) mfc_st16_if(oaddr, make_uint4(a16[0], b16[0], a16[1], b16[1]), p.wt);
                    }
                }
            }
            if (p.out_stats) {
                // lanes -> row sums -> this wave's LDS slots (only this wave touches them until the flush)
                float* rw = red + (ybfast ? yb : 0) * G_RED1;
                if (bnm) {
                    if constexpr (FUSE) {
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float sa = g_row16_sum(tsum[pr][e]), sb = g_row16_sum(tsq[pr][e]);
                                if ((lane & 15) == 0) {
                                    rw[(wave * 2 + 0) * 64 + pr * 32 + (lane >> 4) * 8 + e] += sa;
                                    rw[(wave * 2 + 1) * 64 + pr * 32 + (lane >> 4) * 8 + e] += sb;
                                }
                            }
                    }
                } else if constexpr (!FUSE) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sa = g_row16_sum(ssum[nt][r]), sb = g_row16_sum(ssq[nt][r]);
                            if ((lane & 15) == 0) {
                                rw[(wave * 2 + 0) * 64 + nt * 16 + cq + r] += sa;
                                rw[(wave * 2 + 1) * 64 + nt * 16 + cq + r] += sb;
                            }
                        }
                }
                red_live = true; red_grp = grp; red_yb = yb;
            }
        }
        if (++h == 2) { h = 0; if (++c == p.nchunks) { c = 0; ++u; if (u < nunits) unit_of(u, t, yb); } }
    }
    if (p.out_stats && red_live) stats_flush();
}

int g_conv_gemm = 1;                 // big 1x1 convolutions through conv_gemm1x1_kernel; tuning: mfc_set_flag(23, v)
int g_conv_gemm_minc = 128;          // smallest Cin / Cout (channels) sent there; tuning: mfc_set_flag(24, n)

bool gemm1x1_eligible(const mfc_conv_desc* d) {
    if (!g_conv_gemm || !mfc_is16(d->dtype) || (d->flags & MFC_CONV_S2_CLASSES)) return false;
    if (d->TA != 1 || d->TB != 1 || d->in_stride != 1 || d->dh0 != 0 || d->dw0 != 0) return false;
    if (d->out_sh != 1 || d->out_sw != 1 || d->out_oh != 0 || d->out_ow != 0) return false;
    if (d->Hl != d->Hout || d->Wl != d->Wout || d->Hin != d->Hout || d->Win != d->Wout) return false;
    if (d->Cin_p % 8 || d->Cout_p % 8 || d->Cout > G_BIAS_FLOATS) return false;
    // wide enough for the 256-channel tile: >= 128 channels on both sides, or a full 256-channel block from >= 64 (measured:
    // 64 -> 256 at 120x160: 93 -> 74 us; 64 -> 128: 45 -> 49 us; 256 -> 64: 66 -> 94 us)
    if (!((d->Cin >= g_conv_gemm_minc && d->Cout >= g_conv_gemm_minc) || (d->Cin >= 64 && d->Cout >= 256))) return false;
    const long M = (long)d->N * d->Hout * d->Wout;
    if (M % G_BM || M / G_BM > 0x3fffffff) return false;
    if (d->out_stats || d->in_coef) {
        if (d->images_per_group <= 0 || d->N % d->images_per_group) return false;
        if (d->in_coef && (long)(d->N / d->images_per_group) * 2 * d->Cin_p > G_COEF_FLOATS) return false;      // coefficients staged in LDS
        if (((long)d->images_per_group * d->Hout * d->Wout) % G_BM) return false;      // a pixel tile never straddles a statistics group
    }
    return true;
}

// the acc_src / bn_y epilogue fusions: the BatchNorm's coefficient block is staged where the input-transform coefficients would be
static bool gemm1x1_fusable(const mfc_conv_desc* d) {
    if (d->in_coef || d->bias || d->images_per_group <= 0 || d->N % d->images_per_group) return false;
    if (((long)d->images_per_group * d->Hout * d->Wout) % G_BM) return false;
    return (long)(d->N / d->images_per_group) * 4 * d->Cout_p <= G_COEF_FLOATS;
}

int gemm1x1_layout(const mfc_conv_desc* d, mfc_conv_layout* out) {
    const int Cin_g = d->Cin_p / 8;
    out->KG = G_KG; out->nchunks = ceil_div(Cin_g, G_KG); out->NT16 = G_BN; out->Yblocks = ceil_div(d->Cout, G_BN);
    out->nslots = G_KG; out->TA = 1; out->TB = 1; out->TAS = 1; out->lds_bytes = G_LDS;
    out->bytes = (int64_t)out->nchunks * out->Yblocks * G_BBYTES;
    const int ntiles = (int)(((long)d->N * d->Hout * d->Wout) / G_BM);
    out->MT = 8; out->TH = 1; out->TW = G_BM; out->grid = ntiles < 256 ? ntiles : 256; out->per_block = ceil_div(ntiles, out->grid); out->NW = 8;
    out->fa = gemm1x1_fusable(d) ? 1 : 0;           // (the acc_src / bn_y epilogue fusions: the FUSE instantiation)
    return MFC_OK;
}

int gemm1x1_launch(const mfc_conv_desc* d, hipStream_t st) {
    if (!d->in || !d->wp || !d->out) return MFC_ERR_INVALID_ARG;
    const bool fused = d->acc_src || d->bn_y;
    if (fused) {
        if (d->acc_src && !d->accumulate) return MFC_ERR_INVALID_ARG;
        if (d->bn_y && (!d->bn_coef || !d->out_stats || (d->bn_mask_mode != 0 && d->bn_mask_mode != 2 && d->bn_mask_mode != 3) ||
                        (d->bn_mask_mode == 3 && !d->bn_bits) || d->bias)) return MFC_ERR_INVALID_ARG;
        if (d->out_stats && !d->bn_y) return MFC_ERR_UNSUPPORTED;          // (the FUSE instantiation keeps only the fused form's sums)
        if (!gemm1x1_fusable(d)) return MFC_ERR_UNSUPPORTED;
    }
    GemmK k;
    k.in = (const char*)d->in; k.wp = (const char*)d->wp; k.out = (char*)d->out; k.bias = d->bias; k.out_stats = d->out_stats;
    k.M = d->N * d->Hout * d->Wout; k.Cin_p = d->Cin_p; k.Cin_g = d->Cin_p / 8; k.Cout_p = d->Cout_p; k.Cout = d->Cout;
    k.nchunks = ceil_div(k.Cin_g, G_KG); k.Yblocks = ceil_div(d->Cout, G_BN);
    k.ntiles = k.M / G_BM;
    int grid = k.ntiles < 256 ? k.ntiles : 256;
    k.tiles_per_block = ceil_div(k.ntiles, grid);
    grid = ceil_div(k.ntiles, k.tiles_per_block);
    k.in_coef = d->in_coef; k.in_relu = d->in_relu;
    const bool grouped = d->out_stats || d->in_coef;
    k.G = grouped ? d->N / d->images_per_group : 1;
    k.accumulate = d->accumulate;
    k.wt = 0;          // (measured slower with write-through stores: 1.52 -> 1.58 ms per serial step; common.h)
    k.px_per_group = grouped ? d->images_per_group * d->Hout * d->Wout : k.M;
    k.acc_src = (const char*)d->acc_src; k.bn_y = (const char*)d->bn_y; k.bn_coef = d->bn_coef; k.bn_bits = (const unsigned char*)d->bn_bits; k.bn_mode = d->bn_mask_mode;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_gemm1x1_kernel<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_gemm1x1_kernel<f16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_gemm1x1_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_gemm1x1_kernel<f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        const double flops = 2.0 * k.M * (double)d->Cout * (double)d->Cin;
        const double bytes = (double)k.M * (k.Cin_p + k.Cout_p * (1.0 + (d->accumulate ? 1.0 : 0.0) + (d->bn_y ? 1.0 : 0.0))) * 2.0;
        mfc_prof_before(st, d->dtype == MFC_F16 ? (fused ? "conv_gemm1x1_kernel<_Float16, true>" : "conv_gemm1x1_kernel<_Float16, false>")
                                                : (fused ? "conv_gemm1x1_kernel<__bf16, true>" : "conv_gemm1x1_kernel<__bf16, false>"), flops, bytes);
    }
    if (fused) { MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL((conv_gemm1x1_kernel<T_, true>), dim3(grid), dim3(512), G_LDS, st, k)); }
    else { MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL((conv_gemm1x1_kernel<T_, false>), dim3(grid), dim3(512), G_LDS, st, k)); }
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}
