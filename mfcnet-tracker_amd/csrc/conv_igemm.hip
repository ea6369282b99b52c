// Implicit-GEMM convolution on MFMA for gfx950 (forward, and data-gradient by re-use).
//
// PERSISTENT, software-pipelined design:
//   * a workgroup (4 waves, one per SIMD) owns a contiguous run of output tiles; a tile is a TH x TW patch
//     of one image (<= 64*MT pixels) x NT*16 output channels; wave w computes MT 16-pixel m-tiles x NT n-tiles
//     with v_mfma_f32_16x16x32_bf16 (bf16) or v_mfma_f32_16x16x4_f32 (exact fp32), fp32 accumulate;
//   * work is flattened into STAGES = (tile, channel chunk, tap row).  Per stage the packed weights of that
//     tap row/chunk are needed in LDS, per (tile, chunk) the input halo patch.  While stage s computes from
//     LDS, the global loads of stage s+1 (weights, and the next patch when it changes) are already in flight
//     into registers; they are written to the other weight buffer after the MFMAs, one barrier per stage.
//     So HBM/L2 latency hides under MFMA work, across tile boundaries too;
//   * the k*k taps re-read the halo patch from LDS, never from L2/HBM; the producer's BatchNorm-apply + ReLU
//     is fused into the patch store (zero padding AFTER the transform);
//   * MFMA roles: A = weights (rows = cout), B = pixels (cols = pixel): each lane ends with 4 consecutive
//     output channels of one pixel -> 8/16-byte vector stores along the NHWC channel axis;
//   * epilogue: + bias, optional accumulate, per-(group,channel) sum / sum-of-squares -> replica atomics.
//   * block ids are remapped so that each XCD (private L2) owns a contiguous range of tiles.
// Replaces: nn.Conv2d forward / cuDNN dgrad of models/hrnet.py:39-42,82-88,200-230,361-386,334-351
// and models/multiframe_model.py:191-201 (see include/mfcnet_hip.h).
#include "common.h"
#include <math.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

struct ConvK {
    const char* in; const char* wp; char* out;
    const float* bias; const float* in_coef; mfc_stat_t* out_stats;
    int N, Hin, Win, Cin_p, Cin_g;      // Cin_g: granules to reduce over
    int Hout, Wout, Cout_p, Cout;
    int Hl, Wl, TA, TB, dh0, dw0, s;
    int osh, osw, ooh, oow;
    int in_relu, ipg, G, accumulate;
    int TH, TW, tilesY, tilesX, ntiles;
    int KG, nchunks, TAS, nstg;         // chunk granules (Cin_g % KG == 0), #chunks, tap rows per stage, stages per chunk
    int PH, PW, pitch;                  // patch dims (pixels) and pixel pitch (bytes)
    int Yblocks, nunits, per_block;     // units = cout blocks (slow) x tiles (fast); units per workgroup
    int nslots, npieces, stage_bytes;   // weight slots per stage, 1-KiB DMA pieces per stage, bytes of one stage image
    int off_w0, off_w1, off_ktab, off_red;   // LDS offsets (bytes)
    int off_dummy;                           // 16-byte LDS slot dead staging pieces are stored to
    int ybfast;                              // unit order: 1 = cout block fastest (the Yblocks units of a pixel tile run back to back on one workgroup: the re-reads of the tile hit L2 instead of HBM)
    int wres;                                // >0: single-stage launch whose Yblocks weight stages ALL stay in LDS (wres = Yblocks); the pixel tile is staged once for its Yblocks units
    int wt;                                  // write-through output stores (common.h: large outputs only)
    int ncls;                                // 4: data gradient of a 3x3 / stride-2 convolution, all four output parity classes in this launch (MFC_CONV_S2_CLASSES); else 1
    long cls_bytes;                          // bytes of one class's packed weight image
    int ablate;                              // tuning only: 1 skip weight DMA, 2 skip patch staging, 4 skip output stores, 8 skip MFMAs
    // data-gradient epilogue fusions (bf16 FA variants; include/mfcnet_hip.h): accumulate from another tensor; BatchNorm-backward statistics
    const char* acc_src; const char* bn_y; const float* bn_coef; const unsigned char* bn_bits; int bn_mode;
};

template <int CTRL> __device__ inline float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (result in every lane): quad xor1, quad xor2, row_half_mirror, row_mirror
__device__ inline float row16_sum(float v) {
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
    return v;
}

// NW = waves per workgroup: 4 (two workgroups per CU, 80 KiB LDS each) or 8 (ONE workgroup per CU with the whole 160 KiB: the
// weight stage is shared by twice the waves, so the channel chunk doubles and the number of stages -- each with its fixed
// prefetch-issue / barrier cost -- halves)
// FUSE: the data-gradient epilogue fusions (acc_src / bn_y) are compiled in (FA variants only; a separate instantiation, so that
// the register allocation of the plain kernels is untouched)
#ifdef MFC_CONV_TRACE
// diagnostic build only (make TRACE=1, never shipped): s_memtime stamps of wave 0 of one workgroup; tools/trace_conv.py prints them
__device__ long long g_conv_trace[4096];
#define TR(tag) do { if (tr_on && tr_n < 2040) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if (lane == 0) { g_conv_trace[2 * tr_n] = (tag); g_conv_trace[2 * tr_n + 1] = (long long)t_; } ++tr_n; } } while (0)
extern "C" int mfc_conv_trace_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_trace), sizeof(long long) * 4096) == hipSuccess ? 0 : -1; }
#else
#define TR(tag) do {} while (0)
#endif

template <typename T, int NT, int MT, int PMAX, int NW, bool FUSE = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void conv_igemm_kernel(ConvK p) {
    constexpr int E = Gran<T>::E;
    constexpr bool BF = (E == 8);
    constexpr int NT16 = NT * 16;
    constexpr bool FA = (NT * MT <= 8) || (MT == 2 && NT <= 4 && PMAX <= 4);      // variants with register slack (see below)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    int* ktab = (int*)(smem + p.off_ktab);
    float* red = (float*)(smem + p.off_red);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
#ifdef MFC_CONV_TRACE
    const bool tr_on = (blockIdx.x == gridDim.x / 2 + 3) && wave == 0; int tr_n = 0;
#endif
    const int u0 = Lb * p.per_block;
    const int nun = min(p.per_block, p.nunits - u0);
    if (nun <= 0) return;
    if (p.ablate & 16) return;
    const int SPT = p.nchunks * p.nstg;
    const int total = nun * SPT;
    const int npix = p.PH * p.PW;
    const int cq = (lane >> 4) * 4;
    const int kg = p.KG;

    // ---------------- one-time per-thread tables (tile shape is fixed for the whole launch) ----------------
    int pbase[MT]; int pty[MT], ptx[MT]; bool pin[MT];
    int eoff16[FA ? MT : 1];
    int eoff[FA ? MT : 1];                   // byte offset of the lane's output cell from the tile origin (used by the FA epilogue)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int pp = (wave * MT + mt) * 16 + (lane & 15);
        int ty = pp / p.TW, tx = pp - ty * p.TW;
        bool v = pp < p.TH * p.TW;
        if (!v) { ty = 0; tx = 0; }
        pty[mt] = ty; ptx[mt] = tx; pin[mt] = v;
        pbase[mt] = ((ty * p.s) * p.PW + tx * p.s) * p.pitch + (BF ? 0 : (lane >> 4) * 4);
        if constexpr (FA) {
            eoff[mt] = (((ty * p.osh) * p.Wout + tx * p.osw) * p.Cout_p + (lane >> 4) * 4) * (int)sizeof(T);
            eoff16[mt] = (((ty * p.osh) * p.Wout + tx * p.osw) * p.Cout_p + (lane >> 4) * 8) * (int)sizeof(T);      // transposed pair layout: 8 channels per lane
        }
    }
    // patch pieces this thread stages: granule p_gi of pixels tid/KGP + i*pstep  (py<<16|px, or -1)
    int KGP = 1; while (KGP < kg) KGP <<= 1;
    const int p_gi = tid & (KGP - 1), pstep = (NW * 64) / KGP;
    int pyx[PMAX];
    // FA ("fast addressing", variants with register slack): per piece the byte offset from the first patch pixel of an interior
    // tile (pgo) and the LDS byte offset (plo, a dummy slot for pieces that carry nothing), so that the prefetch of an interior
    // tile is `scalar base + constant lane offset` (no vector arithmetic) and the LDS store is branch-free
    int pgo[FA ? PMAX : 1], plo[FA ? PMAX : 1];
    unsigned pvalid = 0;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int pix = tid / KGP + i * pstep;
        const int py = pix / p.PW, px = pix - py * p.PW;
        const bool ok = pix < npix && p_gi < kg;
        pyx[i] = ok ? ((py << 16) | px) : -1;
        pvalid |= (ok ? 1u : 0u) << i;
        if constexpr (FA) {
            pgo[i] = ok ? ((py * p.Win + px) * p.Cin_p * (int)sizeof(T) + p_gi * 16) : 0;
            plo[i] = ok ? ((py * p.PW + px) * p.pitch + p_gi * 16) : p.off_dummy;
        }
    }
    // slot -> patch offset table (tail slots repeat the last valid one; their packed weights are zero)
    if (tid < p.nslots) {
        int q = min(tid, p.TAS * p.TB * kg - 1);
        int al = q / (p.TB * kg); q -= al * p.TB * kg;
        int b = q / kg, gi = q - b * kg;
        ktab[tid] = (al * p.PW + b) * p.pitch + gi * 16;
    }

    f32x4 acc[MT][NT];
    float ssum[NT][4], ssq[NT][4];          // running per-lane statistics partials (flushed when (group, cout block) changes)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }
    }

    uint4 preg[PMAX]; unsigned pmask = 0;
    float sc[E], sh[E];
    const bool p_xf = (p.in_coef != nullptr);

    // unit = (cout block yb [slowest], image n, tile row, tile column [fastest]); tracked incrementally (no divisions per stage)
    // (cls = output parity class of a merged stride-2 data gradient, slowest: class c writes pixels (2i + (c >> 1), 2j + (c & 1)) from its own weight image)
    struct UC { int yb, n, tyi, txi, cls; };
    auto uc_init = [&](int u) {
        UC r; int xt;
        r.cls = 0;
        if (p.ncls > 1) { const int per = p.ntiles * p.Yblocks; r.cls = u / per; u -= r.cls * per; }
        if (p.ybfast) { xt = u / p.Yblocks; r.yb = u - xt * p.Yblocks; }
        else { r.yb = u / p.ntiles; xt = u - r.yb * p.ntiles; }
        r.txi = xt % p.tilesX; xt /= p.tilesX; r.tyi = xt % p.tilesY; r.n = xt / p.tilesY;
        return r;
    };
    auto uc_next = [&](UC r) {
        if (p.ybfast) {
            if (++r.yb < p.Yblocks) return r;
            r.yb = 0;
            if (++r.txi == p.tilesX) { r.txi = 0; if (++r.tyi == p.tilesY) { r.tyi = 0; if (++r.n == p.N && p.ncls > 1) { r.n = 0; ++r.cls; } } }
            return r;
        }
        if (++r.txi == p.tilesX) { r.txi = 0; if (++r.tyi == p.tilesY) { r.tyi = 0; if (++r.n == p.N) { r.n = 0; if (++r.yb == p.Yblocks && p.ncls > 1) { r.yb = 0; ++r.cls; } } } }
        return r;
    };
    auto load_patch = [&](const UC& uc, int c) {
        const int n = uc.n, i0 = uc.tyi * p.TH, j0 = uc.txi * p.TW;
        const int g0 = c * kg;
        pmask = 0;
        // Branch-free: every piece issues its load unconditionally from a clamped (always valid) address, so the compiler
        // keeps the loads back to back and can count them (conditional loads force s_waitcnt vmcnt(0) before each one).
        const int gic = min(p_gi, kg - 1);
        if (p_xf) {
            const float* cf = p.in_coef + (size_t)(n / p.ipg) * 4 * p.Cin_p + (g0 + gic) * E;
#pragma unroll
            for (int e = 0; e < E; ++e) { sc[e] = cf[e]; sh[e] = cf[p.Cin_p + e]; }
        }
        const int ih0 = i0 * p.s + p.dh0, iw0 = j0 * p.s + p.dw0;
        const char* base = p.in + (size_t)n * p.Hin * p.Win * p.Cin_p * sizeof(T) + (size_t)(g0 + gic) * 16;
        const int cs = p.Cin_p * (int)sizeof(T);
        if (ih0 >= 0 && iw0 >= 0 && ih0 + p.PH <= p.Hin && iw0 + p.PW <= p.Win) {
            // interior tile (uniform branch): no bounds arithmetic at all
            if constexpr (FA) {
                const char* tb = p.in + ((size_t)n * p.Hin * p.Win + (size_t)(ih0 * p.Win + iw0)) * cs + (size_t)g0 * 16;      // wave-uniform
#pragma unroll
                for (int i = 0; i < PMAX; ++i) preg[i] = *(const uint4*)(tb + (unsigned)pgo[i]);
                pmask = pvalid;
            } else {
                const char* tb = base + (size_t)(ih0 * p.Win + iw0) * cs;
#pragma unroll
                for (int i = 0; i < PMAX; ++i) {
                    const int pv = pyx[i] >= 0 ? pyx[i] : 0;
                    preg[i] = *(const uint4*)(tb + ((pv >> 16) * p.Win + (pv & 0xffff)) * cs);
                    pmask |= (pyx[i] >= 0 ? 1u : 0u) << i;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < PMAX; ++i) {
                const int ih = ih0 + (pyx[i] >> 16), iw = iw0 + (pyx[i] & 0xffff);
                const bool inr = pyx[i] >= 0 && ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Win;
                const int ihc = min(max(ih, 0), p.Hin - 1), iwc = min(max(iw, 0), p.Win - 1);
                preg[i] = *(const uint4*)(base + (size_t)(ihc * p.Win + iwc) * cs);
                pmask |= (inr ? 1u : 0u) << i;
            }
        }
    };
    auto store_patch = [&]() {
        if constexpr (FA) {
            // every lane stores every piece: dead pieces land in the dummy slot, out-of-image pieces store zeros
#pragma unroll
            for (int i = 0; i < PMAX; ++i) {
                uint4 v = preg[i];
                if (p_xf) {
                    float f[E];
                    Gran<T>::unpack(v, f);
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        float t = f[e] * sc[e] + sh[e];
                        f[e] = p.in_relu ? relu_nan(t) : t;
                    }
                    v = Gran<T>::pack(f);
                }
                if (!(pmask & (1u << i))) v = make_uint4(0, 0, 0, 0);
                *(uint4*)(smem + plo[i]) = v;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            if (pyx[i] >= 0) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (pmask & (1u << i)) {
                    v = preg[i];
                    if (p_xf) {
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
                *(uint4*)(patch + ((pyx[i] >> 16) * p.PW + (pyx[i] & 0xffff)) * p.pitch + p_gi * 16) = v;
            }
        }
    };
    // async global -> LDS copy of the packed weights of stage (a, c, cout block): the packed image is laid out
    // [TA][nchunks][Yblocks][nslots][NT16][16 B], i.e. one stage is ONE contiguous block = the LDS image
    auto dma_w = [&](const UC& uc, int c, int ag, char* wl) {
        const char* src = p.wp + (size_t)uc.cls * p.cls_bytes + (size_t)((ag * p.nchunks + c) * p.Yblocks + uc.yb) * p.stage_bytes + lane * 16;
        // Issued as inline asm: through the builtin hipcc assumes the DMA's LDS write may alias every later ds_read and
        // drains it (s_waitcnt vmcnt(0)) before the compute loop, i.e. no overlap.  The buffers are disjoint by
        // construction (double buffer); completion is awaited explicitly (dma_wait) before the stage-end barrier.
        const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wl;
        for (int piece = wave; piece < p.npieces; piece += NW) {
            const char* g = src + piece * 1024;
            const unsigned ldst = __builtin_amdgcn_readfirstlane(lbase + piece * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(ldst) : "memory");
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // statistics: lanes -> row sums -> LDS (per wave) ; the cross-wave sum + atomics happen after the next barrier
    int red_par = 0; bool red_pending = false; int red_n0 = 0, red_grp = 0, red_rep = 0;
    auto stats_to_lds = [&](int n0, int grp, int rep) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sa = row16_sum(ssum[nt][r]), sb = row16_sum(ssq[nt][r]);
                ssum[nt][r] = 0.f; ssq[nt][r] = 0.f;
                if ((lane & 15) == 0) {
                    // (BatchNorm-backward mode keeps the sums in the TRANSPOSED pair layout of the epilogue: entry [nt][r] of row g = lane >> 4
                    //  is channel (nt / 2) * 32 + 8 g + (nt & 1) * 4 + r)
                    const int cl = (FUSE && p.bn_y) ? ((nt >> 1) * 32 + (lane >> 4) * 8 + (nt & 1) * 4 + r) : (nt * 16 + cq + r);
                    red[red_par * 2 * NW * NT16 + (wave * 2 + 0) * NT16 + cl] = sa;
                    red[red_par * 2 * NW * NT16 + (wave * 2 + 1) * NT16 + cl] = sb;
                }
            }
        red_pending = true; red_n0 = n0; red_grp = grp; red_rep = rep; red_par ^= 1;
    };
    auto stats_flush = [&]() {
        if (tid < 2 * NT16) {
            const int which = tid / NT16, cl = tid - which * NT16;
            if (red_n0 + cl < p.Cout_p) {
                const float* rd = red + (red_par ^ 1) * 2 * NW * NT16;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) s += rd[(w * 2 + which) * NT16 + cl];
                atomicAdd(p.out_stats + (((size_t)red_rep * p.G + red_grp) * 2 + which) * p.Cout_p + red_n0 + cl, (mfc_stat_t)s);
            }
        }
        red_pending = false;
    };

    // ---------------- prologue ----------------
    TR(0);
    UC uc = uc_init(u0), uc2 = uc;
    const int wstride = p.npieces * 1024;
    if (p.wres) {
        for (int yb = 0; yb < p.wres; ++yb) { UC t = uc; t.yb = yb; dma_w(t, 0, 0, smem + p.off_w0 + yb * wstride); }
    } else {
        dma_w(uc, 0, 0, smem + p.off_w0);
    }
    load_patch(uc, 0);
    store_patch();
    dma_wait();
    __syncthreads();
    TR(1);

    if (p.ablate & 32) return;
    const bool resident = (p.nchunks == 1 && p.nstg == 1 && p.Yblocks == 1 && p.ncls <= 1) || p.wres;
    int tl = 0, c = 0, a = 0;          // current stage coordinates (a = tap-row group)
    for (int g = 0; g < total; ++g) {
        int tl2 = tl, c2 = c, a2 = a + 1;
        if (a2 == p.nstg) { a2 = 0; ++c2; if (c2 == p.nchunks) { c2 = 0; ++tl2; uc2 = uc_next(uc); } }
        const bool nxt = (g + 1 < total);
        const bool newpatch = nxt && (a2 == 0) && (p.nchunks > 1 || (p.wres ? uc2.yb == 0 : tl2 != tl));
        // (single-stage launches with one cout block use the same weights for every tile: they stay in buffer 0)
        if (nxt && !resident && !(p.ablate & 1)) dma_w(uc2, c2, a2, smem + (((g + 1) & 1) ? p.off_w1 : p.off_w0));
        if (newpatch && !(p.ablate & 2)) load_patch(uc2, c2);
        if (red_pending) stats_flush();
        TR(2);

        // ---------------- compute stage (tl, c, a) ----------------
        {
            const char* wl = smem + (p.wres ? p.off_w0 + uc.yb * wstride : (((g & 1) && !resident) ? p.off_w1 : p.off_w0));
            const char* pa = patch + a * p.TAS * p.PW * p.pitch;
            if constexpr (BF) {
                // Software-pipelined, ping-pong unrolled by two: the LDS fragment reads of k-step k+1 are issued before the
                // MFMAs of k-step k (loop-carried across the back edge, so the compiler cannot sink them next to their use).
                const int nk = p.nslots >> 2;
                const int gl = lane >> 4;
                const char* wlb = wl + (gl * NT16 + (lane & 15)) * 16;
                constexpr int KSS = 4 * NT16 * 16;
                bf16x8 xa[MT], wa[NT], xb[MT], wb[NT];
                auto ldx = [&](bf16x8* x, int ks) {
                    const int kq = ktab[ks * 4 + gl];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) x[mt] = *(const bf16x8*)(pa + pbase[mt] + kq);
                };
                auto ldw = [&](bf16x8* w, int ks) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) w[nt] = *(const bf16x8*)(wlb + ks * KSS + nt * 256);
                };
                auto mm = [&](const bf16x8* w, const bf16x8* x) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = mfma16<T>(w[nt], x[mt], acc[mt][nt]);
                };
                ldx(xa, 0); ldw(wa, 0);
                int ks = 0;
                if (p.ablate & 8) ks = nk;
                for (; ks + 2 <= nk; ks += 2) {
                    ldx(xb, ks + 1); ldw(wb, ks + 1);
                    mm(wa, xa);
                    __builtin_amdgcn_sched_group_barrier(0x100, MT + NT + 1, 0);     // DS reads of the next step first ...
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);         // ... then this step's MFMAs
                    const int k2 = min(ks + 2, nk - 1);
                    ldx(xa, k2); ldw(wa, k2);
                    mm(wb, xb);
                    __builtin_amdgcn_sched_group_barrier(0x100, MT + NT + 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
                }
                if (ks < nk) mm(wa, xa);
            } else {
                for (int q = 0; q < p.nslots; ++q) {
                    const int ko = ktab[q];
                    float xv[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) xv[mt] = *(const float*)(pa + pbase[mt] + ko);
                    const char* wq = wl + (q * NT16 + (lane & 15)) * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float wf = *(const float*)(wq + nt * 256);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, xv[mt], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }

        TR(3);
        // ---------------- tile epilogue ----------------
        if (c == p.nchunks - 1 && a == p.nstg - 1 && !(p.ablate & 64)) {
            const int n = uc.n, i0 = uc.tyi * p.TH, j0 = uc.txi * p.TW, yb = uc.yb;
            const int n0 = yb * NT16;
            // logical grid and output phase of this unit: the launch's, or its parity class's
            const int c_ooh = p.ncls > 1 ? (uc.cls >> 1) : p.ooh, c_oow = p.ncls > 1 ? (uc.cls & 1) : p.oow;
            const int c_Hl = p.ncls > 1 ? ((p.Hout - c_ooh + 1) >> 1) : p.Hl, c_Wl = p.ncls > 1 ? ((p.Wout - c_oow + 1) >> 1) : p.Wl;
            if constexpr (FA) {
                // Variants with register slack: wave-uniform tile origin + the per-lane offsets of the prologue (no 64-bit
                // vector address math per store).  bf16: the MFMA leaves a lane with 4 channels of cout tile nt; two cout tiles
                // are transposed across the four 16-lane rows (v_permlane32_swap + v_permlane16_swap) so that every lane owns
                // 8 CONTIGUOUS channels of its pixel -> one 16-byte store per tile pair, 64 contiguous bytes per pixel (the
                // 8-byte stores to 32-byte half lines cost ~250 cycles of issue each: the memory pipeline works per line touched)
                const size_t toff = ((((size_t)n * p.Hout + (i0 * p.osh + c_ooh)) * p.Wout + (j0 * p.osw + c_oow)) * p.Cout_p + n0) * sizeof(T);
                char* tbase = p.out + toff;
                const char* abase = ((FUSE && p.acc_src) ? p.acc_src : (const char*)p.out) + toff;      // where the running sum of an accumulating launch lives
                const bool bnm = FUSE && BF && p.bn_y != nullptr;                             // fused BatchNorm-backward statistics (wave-uniform)
                const bool full = (i0 + p.TH <= c_Hl) && (j0 + p.TW <= c_Wl);
                constexpr int NP = BF ? NT / 2 : 0;              // cout tile pairs stored transposed
                // Fused data-gradient epilogue: ALL global reads of the tile epilogue (running sum, pre-BN tensor, mask bits) are issued up
                // front, so their latency is paid once per tile and not once per 16-pixel row (issued next to their use they cost
                // +11 us on a 23 us launch: 1-2 us of exposed latency per row, two workgroups per CU to hide it)
                constexpr int NPF = (FUSE && BF) ? (NP > 0 ? NP : 1) : 1;
                uint4 pf_old[FUSE ? MT : 1][NPF], pf_y[FUSE ? MT : 1][NPF]; unsigned pf_bits[FUSE ? MT : 1][NPF];
                if constexpr (FUSE && BF) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const bool vpx = pin[mt] && (full || ((i0 + pty[mt] < c_Hl) && (j0 + ptx[mt] < c_Wl)));
#pragma unroll
                        for (int pr = 0; pr < NP; ++pr) {
                            const bool vc = (n0 + pr * 32 + (lane >> 4) * 8) < p.Cout_p;
                            const unsigned lo = (vpx && vc) ? (unsigned)(eoff16[mt] + pr * 64) : 0u;
                            if (p.accumulate) pf_old[mt][pr] = *(const uint4*)(abase + lo);
                            if (bnm) pf_y[mt][pr] = *(const uint4*)(p.bn_y + toff + lo);
                            if (bnm && p.bn_mode == 3) pf_bits[mt][pr] = p.bn_bits[(toff + lo) >> 4];
                        }
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bool vpx = pin[mt] && (full || ((i0 + pty[mt] < c_Hl) && (j0 + ptx[mt] < c_Wl)));
                    float v[NT][4];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[nt][r] = acc[mt][nt][r];
                            acc[mt][nt][r] = 0.f;
                        }
                        if (p.bias) {          // (only the last_layer and head convolutions carry a bias)
                            float bq[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) bq[r] = p.bias[min(n0 + nt * 16 + cq + r, p.Cout - 1)];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[nt][r] += (n0 + nt * 16 + cq + r) < p.Cout ? bq[r] : 0.f;
                        }
                    }
                    if constexpr (BF) {
#pragma unroll
                        for (int pr = 0; pr < NP; ++pr) {
                            // lane (row g = lane>>4) after the transpose: channels pr*32 + 8g .. +7 of its pixel
                            const unsigned off = (unsigned)(eoff16[mt] + pr * 64);
                            const bool vc = (n0 + pr * 32 + (lane >> 4) * 8) < p.Cout_p;
                            uint4 oldq = make_uint4(0, 0, 0, 0);
                            if constexpr (FUSE) { if (p.accumulate) oldq = pf_old[mt][pr]; }
                            else { if (p.accumulate) oldq = *(const uint4*)(abase + ((vpx && vc) ? off : 0u)); }
                            if (p.accumulate || bnm) {
                                // dgrad accumulation happens in fp32 on the transposed layout: expand, transpose, add, round once
                                float w[8];
                                // (the sums must be rounded once: transpose the fp32 values as raw dwords)
                                unsigned t0[4], t1[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    auto x32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * pr][r]), __float_as_uint(v[2 * pr + 1][r]), false, false);
                                    auto x16 = __builtin_amdgcn_permlane16_swap(x32[0], x32[1], false, false);
                                    t0[r] = x16[0]; t1[r] = x16[1];
                                }
                                float o8[8];
                                Gran<T>::unpack(oldq, o8);
#pragma unroll
                                for (int r = 0; r < 4; ++r) { w[r] = __uint_as_float(t0[r]) + o8[r]; w[4 + r] = __uint_as_float(t1[r]) + o8[4 + r]; }
                                if (bnm) {
                                    // BatchNorm / ReLU backward of the tensor this launch completes: mask, store the MASKED gradient, and keep
                                    // sum g*m, sum g*m*yhat of the lane's 8 channels (what mfc_bnbwd_reduce would sweep the tensor for again)
                                    const bool live = vpx && vc;
                                    float yv[8];
                                    Gran<T>::unpack(pf_y[FUSE ? mt : 0][pr], yv);
                                    const int ch = min(n0 + pr * 32 + (lane >> 4) * 8, p.Cout_p - 8);
                                    const float* cf = p.bn_coef + (size_t)(n / p.ipg) * 4 * p.Cout_p + ch;
                                    const float4 m0 = *(const float4*)(cf + 2 * p.Cout_p), m1 = *(const float4*)(cf + 2 * p.Cout_p + 4);
                                    const float4 r0 = *(const float4*)(cf + 3 * p.Cout_p), r1 = *(const float4*)(cf + 3 * p.Cout_p + 4);
                                    const float mean[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
                                    const float rstd[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
                                    unsigned mk = 0xffu;
                                    if (p.bn_mode == 2) {
                                        const float4 s0 = *(const float4*)cf, s1 = *(const float4*)(cf + 4);
                                        const float4 h0 = *(const float4*)(cf + p.Cout_p), h1 = *(const float4*)(cf + p.Cout_p + 4);
                                        const float sc8[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                                        const float sh8[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                                        mk = 0;
#pragma unroll
                                        for (int e = 0; e < 8; ++e) mk |= ((yv[e] * sc8[e] + sh8[e]) > 0.f ? 1u : 0u) << e;
                                    } else if (p.bn_mode == 3) {
                                        mk = pf_bits[FUSE ? mt : 0][pr];           // one byte per 8-channel granule (mfc_combine_fwd)
                                    }
#pragma unroll
                                    for (int e = 0; e < 8; ++e) {
                                        const float gmv = ((mk >> e) & 1u) ? w[e] : 0.f;
                                        w[e] = gmv;
                                        if (live) {
                                            ssum[2 * pr + (e >> 2)][e & 3] += gmv;
                                            ssq[2 * pr + (e >> 2)][e & 3] += gmv * ((yv[e] - mean[e]) * rstd[e]);
                                        }
                                    }
                                } else if (p.out_stats) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        if (vpx && (n0 + 2 * pr * 16 + cq) < p.Cout_p) { ssum[2 * pr][r] += v[2 * pr][r]; ssq[2 * pr][r] += v[2 * pr][r] * v[2 * pr][r]; }
                                        if (vpx && (n0 + (2 * pr + 1) * 16 + cq) < p.Cout_p) { ssum[2 * pr + 1][r] += v[2 * pr + 1][r]; ssq[2 * pr + 1][r] += v[2 * pr + 1][r] * v[2 * pr + 1][r]; }
                                    }
                                }
                                if (vpx && vc && !(p.ablate & 4)) mfc_st16_if((tbase + off), Gran<T>::pack(w), p.wt);
                            } else {
                                const unsigned p0 = pack2<T>(v[2 * pr][0], v[2 * pr][1]), p1 = pack2<T>(v[2 * pr][2], v[2 * pr][3]);
                                const unsigned q0 = pack2<T>(v[2 * pr + 1][0], v[2 * pr + 1][1]), q1 = pack2<T>(v[2 * pr + 1][2], v[2 * pr + 1][3]);
                                auto a32 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                                auto a16 = __builtin_amdgcn_permlane16_swap(a32[0], a32[1], false, false);
                                auto b32 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                                auto b16 = __builtin_amdgcn_permlane16_swap(b32[0], b32[1], false, false);
                                if (p.out_stats) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        if (vpx && (n0 + 2 * pr * 16 + cq) < p.Cout_p) { ssum[2 * pr][r] += v[2 * pr][r]; ssq[2 * pr][r] += v[2 * pr][r] * v[2 * pr][r]; }
                                        if (vpx && (n0 + (2 * pr + 1) * 16 + cq) < p.Cout_p) { ssum[2 * pr + 1][r] += v[2 * pr + 1][r]; ssq[2 * pr + 1][r] += v[2 * pr + 1][r] * v[2 * pr + 1][r]; }
                                    }
                                }
                                if (vpx && vc && !(p.ablate & 4)) mfc_st16_if((tbase + off), make_uint4(a16[0], b16[0], a16[1], b16[1]), p.wt);
                            }
                        }
                    }
#pragma unroll
                    for (int nt = 2 * NP; nt < NT; ++nt) {      // tiles without a partner (odd NT, fp32): 4 channels per lane
                        const bool vc = (n0 + nt * 16 + cq) < p.Cout_p;
                        const unsigned off = (unsigned)(eoff[mt] + nt * 16 * (int)sizeof(T));
                        if (p.accumulate) {
                            float o4[4];
                            Gran<T>::unquad(*(const typename Gran<T>::Quad*)(tbase + ((vpx && vc) ? off : 0u)), o4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[nt][r] += o4[r];
                        }
                        if (vpx && vc && !(p.ablate & 4)) {
                            *(typename Gran<T>::Quad*)(tbase + off) = Gran<T>::quad(v[nt]);
                            if (p.out_stats) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) { ssum[nt][r] += v[nt][r]; ssq[nt][r] += v[nt][r] * v[nt][r]; }
                            }
                        }
                    }
                }
            } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool valid = pin[mt] && (i0 + pty[mt] < c_Hl) && (j0 + ptx[mt] < c_Wl);
                const int oi = (i0 + pty[mt]) * p.osh + c_ooh, oj = (j0 + ptx[mt]) * p.osw + c_oow;
                T* orow = (T*)(p.out + (((size_t)n * p.Hout + oi) * p.Wout + oj) * p.Cout_p * sizeof(T));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int co = n0 + nt * 16 + cq;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[mt][nt][r];
                        acc[mt][nt][r] = 0.f;
                    }
                    if (p.bias) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (co + r < p.Cout) v[r] += p.bias[co + r];
                    }
                    if (valid && co < p.Cout_p && !(p.ablate & 4)) {
                        T* o = orow + co;
                        if (p.accumulate) {
                            if constexpr (BF) {
                                const uint2 ov = *(const uint2*)o;
                                float o4[4];
                                Gran<T>::unquad(ov, o4);
                                v[0] += o4[0]; v[1] += o4[1]; v[2] += o4[2]; v[3] += o4[3];
                            } else {
                                const float4 ov = *(const float4*)o;
                                v[0] += ov.x; v[1] += ov.y; v[2] += ov.z; v[3] += ov.w;
                            }
                        }
                        if constexpr (BF) {
                            *(uint2*)o = Gran<T>::quad(v);
                        } else {
                            *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                        }
                        if (p.out_stats) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { ssum[nt][r] += v[r]; ssq[nt][r] += v[r] * v[r]; }
                        }
                    }
                }
            }
            }
            if (p.out_stats) {
                // flush the running sums when the next unit belongs to another (statistic group, cout block) or the run ends
                bool flush = (tl + 1 >= nun);
                if (!flush) {
                    const UC un = uc_next(uc);
                    flush = (un.yb != yb) || (un.n / p.ipg != n / p.ipg);        // (a class change of a merged stride-2 launch keeps (group, channel): no flush needed)
                }
                if (flush) stats_to_lds(n0, n / p.ipg, (u0 + tl + blockIdx.x) % MFC_R);
            }
        }

        // ---------------- hand over to the next stage ----------------
        TR(4);
        if (newpatch) {
            __syncthreads();           // every wave has finished reading the current patch
            TR(5);
            if (!(p.ablate & 2)) store_patch();
            TR(6);
        }
        if (!resident) dma_wait();     // the weight DMA of stage g+1 has landed (this wave's pieces) ...
        TR(7);
        __syncthreads();               // ... and everybody's
        TR(8);
        tl = tl2; c = c2; a = a2; uc = uc2;
    }
    if (red_pending) stats_flush();
#ifdef MFC_CONV_TRACE
    TR(9);
    if (tr_on && lane == 0) g_conv_trace[4095] = tr_n;
#endif
}

// ------------------------------------------------------------------------------------------
static void choose_tile(int Hl, int Wl, int cap, int& TH, int& TW, double& eff_out) {
    // maximise useful pixels per tile of `cap` pixels; prefer wide / square-ish tiles
    double best = -1; int bh = 1, bw = 1; double be = 0;
    for (int tw = 1; tw <= cap && tw <= Wl; ++tw) {
        int th = cap / tw; if (th > Hl) th = Hl; if (th < 1) continue;
        double tiles = (double)ceil_div(Hl, th) * ceil_div(Wl, tw);
        double eff = (double)Hl * Wl / (tiles * cap);
        double halo = (double)(th * tw) / ((th + 2.0) * (tw + 2.0));
        double score = eff + 0.05 * halo + ((tw % 16 == 0) ? 0.01 : 0.0);
        if (score > best) { best = score; bh = th; bw = tw; be = eff; }
    }
    TH = bh; TW = bw; eff_out = be;
}

static int g_conv_num_cu = 256;
static int g_conv_force_mt = 0;      // tuning: mfc_set_flag(2, 2|4)
int mfc_conv_set_force_mt(int v) { g_conv_force_mt = v; return 0; }
static int g_conv_lds_kb = 80;       // LDS budget per workgroup (80 KiB -> 2 workgroups per CU); tuning: mfc_set_flag(6, kb)
int mfc_conv_set_lds_kb(int v) { g_conv_lds_kb = v > 0 ? v : 80; return 0; }
int g_conv_wres = 1;                 // keep all cout blocks' weights of a single-stage launch in LDS; tuning: mfc_set_flag(20, v)
static int g_conv_ybfast = -1;        // -1 auto, 0 never, 1 always; tuning: mfc_set_flag(8, v)
int mfc_conv_set_ybfast(int v) { g_conv_ybfast = v; return 0; }
static int g_conv_fill_pct = 100;     // exponent (in %) on the under-fill penalty of the geometry search; tuning: mfc_set_flag(18, pct)
int mfc_conv_set_fill_pct(int v) { g_conv_fill_pct = v; return 0; }
static int g_conv_ablate = 0;
int mfc_conv_set_ablate(int v) { g_conv_ablate = v; return 0; }
static int g_conv_grid = 512;        // persistent workgroups per launch (2 per CU); tuning: mfc_set_flag(4, n)
int mfc_conv_set_grid(int v) { g_conv_grid = v > 0 ? v : 512; return 0; }

static int g_conv_nw8 = 90;           // 8-wave (one workgroup per CU) geometries: score weight in % (0 = never; 90 sends ties to the 4-wave form); tuning: mfc_set_flag(19, pct)
int mfc_conv_set_nw8(int v) { g_conv_nw8 = v; return 0; }
int g_conv_nw8_fused = 90;            // the same weight for launches that ask for the fused data-gradient epilogue (MFC_CONV_WANT_FA); -1 = g_conv_nw8, 0 = 4-wave geometries only.
                                      // Measured twice in round 4: before the fuse stage had one join per module and the 64-channel data gradients moved to the ring
                                      // launch, 4-wave geometries won (two workgroups per CU hide each other's exposed epilogue reads: 36.78 -> 36.60 ms); on the final
                                      // structure the 8-wave ones do (701.4 / 704.6 -> 710.0 / 711.1 frames/s, same box, alternating); mfc_set_flag(55, pct)

static bool conv_variant_fa(int dtype, int NT, int MT, int PM, int NW);

static int conv_setup(const mfc_conv_desc* d, ConvK& k, int& NT, int& MT, int& PM, size_t& lds, int& grid, int& NWsel, int pass = 0) {
    // pass 0 honours MFC_CONV_WANT_FA (fusable geometries only); if none exists the search is repeated unrestricted (pass 1)
    if (!d || !d->in || !d->wp || !d->out) return MFC_ERR_INVALID_ARG;          // (before anything reads the descriptor)
    bool want_fa = pass == 0 && (d->flags & MFC_CONV_WANT_FA) && mfc_is16(d->dtype);
    if (want_fa) {
        // only where the fused epilogue's cout blocking (NT in {2, 4}) is also the natural one: for 48 / 96 / 192 / 384 channels (HRNet-W48)
        // the natural blocks are 3 or 6 tiles wide, and splitting them into blocks of 2 / 4 costs more than the fused reduce pass saves
        // (96 -> 96 at 60x80: 73 us fused in three blocks vs 39.5 us + a 20 us reduce pass)
        const int n16n = ceil_div(d->Cout, 16);
        const int cand[5] = {6, 4, 3, 2, 1};
        double bestc = 1e30; int ntn = 1;
        for (int i = 0; i < 5; ++i) {
            double c = (double)ceil_div(n16n, cand[i]) * cand[i] * (1.0 + 0.5 / cand[i]);
            if (c < bestc - 1e-9) { bestc = c; ntn = cand[i]; }
        }
        // (measured as well: last_layer[3]'s data gradient, 5 -> 480 channels, all stores -- blocked 4 tiles wide with the fused reduce it takes
        //  ~0.33 ms longer and saves a 0.27 ms reduce pass)
        if (ntn != 2 && ntn != 4) want_fa = false;
    }
    if (!mfc_dtype_ok(d->dtype)) return MFC_ERR_INVALID_ARG;
    const int E = mfc_is16(d->dtype) ? 8 : 4;
    if (d->Cin_p % 8 || d->Cout_p % 8 || d->Cin > d->Cin_p || d->Cout > d->Cout_p) return MFC_ERR_INVALID_ARG;
    // MFC_CONV_S2_CLASSES: the data gradient of a 3x3 / stride-2 / pad-1 convolution as ONE launch over its four output parity classes (round 4: they
    // used to be four launches of 9-26 us each for a few us of work -- 204 launches, 2.8 ms of a W32 step).  Every class is the same 2x2-tap
    // problem on the dy grid (taps the class does not have are zero in its weight image): logical pixel (i, j) of class (ph, pw) -> tensor
    // pixel (2i + ph, 2j + pw); classes differ only in their weight image (consecutive in wp) and in where they store.
    const bool s2c = (d->flags & MFC_CONV_S2_CLASSES) != 0;
    if (s2c && (d->TA != 2 || d->TB != 2 || d->dh0 != 0 || d->dw0 != 0 || d->in_stride != 1 || d->out_sh != 2 || d->out_sw != 2 || d->out_oh != 0 ||
                d->out_ow != 0 || d->Hl != (d->Hout + 1) / 2 || d->Wl != (d->Wout + 1) / 2 || d->bias || d->TH > 0 || d->TW > 0)) return MFC_ERR_INVALID_ARG;
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
    k.acc_src = (const char*)d->acc_src; k.bn_y = (const char*)d->bn_y; k.bn_coef = d->bn_coef; k.bn_bits = (const unsigned char*)d->bn_bits; k.bn_mode = d->bn_mask_mode;
    const int n16 = ceil_div(d->Cout, 16);
    {   // N tile: fewest computed n-tiles, mild preference for wide tiles (more reuse of the staged patch)
        const int cand[5] = {6, 4, 3, 2, 1};
        double bestc = 1e30; NT = 1;
        for (int i = 0; i < 5; ++i) {
            if (want_fa && n16 % 2 == 0 && cand[i] != 4 && cand[i] != 2) continue;      // the fused epilogue pairs cout tiles: NT in {2, 4}
            double c = (double)ceil_div(n16, cand[i]) * cand[i] * (1.0 + 0.5 / cand[i]);
            if (c < bestc - 1e-9) { bestc = c; NT = cand[i]; }
        }
    }
    k.Yblocks = ceil_div(n16, NT);
    // Search the launch geometry: pixel tile (256 or 128 pixels), tap rows per stage (all or one) and channel chunk KG,
    // subject to LDS <= 80 KiB (2 workgroups per CU) and the per-thread patch-piece budget; prefer the most MFMAs per
    // stage (fewer barriers / less per-stage bookkeeping per MFMA), discounted by padding waste.
    double best_score = -1; ConvK bk = k; int bMT = 2, bPM = 0, bNW = 4; size_t blds = 0;
    for (int nwi = 0; nwi < 2; ++nwi) {
    const int nw = nwi == 0 ? 4 : 8;
    const int nw8w = (want_fa && g_conv_nw8_fused >= 0) ? g_conv_nw8_fused : g_conv_nw8;
    if (nw == 8 && (nw8w <= 0 || E != 8)) continue;
    const int slots = nw == 8 ? g_conv_grid / 2 : g_conv_grid;          // resident workgroups on the chip
    for (int mi = 0; mi < 2; ++mi) {
        const int mt = mi == 0 ? 4 : 2;
        if (g_conv_force_mt && mt != g_conv_force_mt) continue;
        const int cap = nw * 16 * mt;
        ConvK c = k; double eff;
        if (d->TH > 0 && d->TW > 0) {
            if (d->TH * d->TW > cap || (mt == 4 && d->TH * d->TW <= cap / 2)) continue;
            c.TH = d->TH; c.TW = d->TW;
            eff = (double)d->Hl * d->Wl / ((double)ceil_div(d->Hl, c.TH) * ceil_div(d->Wl, c.TW) * cap);
        } else {
            choose_tile(d->Hl, d->Wl, cap, c.TH, c.TW, eff);
        }
        c.tilesY = ceil_div(d->Hl, c.TH); c.tilesX = ceil_div(d->Wl, c.TW);
        c.PH = (c.TH - 1) * c.s + c.TA; c.PW = (c.TW - 1) * c.s + c.TB;
        const long units = (long)c.N * c.tilesY * c.tilesX * c.Yblocks * (s2c ? 4 : 1);
        // parallelism: a launch wants >= ~2 units per CU-slot (512 slots); fewer units -> idle CUs
        const double fill = units >= slots ? 1.0 : pow((double)units / (double)slots, g_conv_fill_pct / 100.0);
        for (int tas = c.TA; tas >= 1; tas = (tas == 1 ? 0 : 1)) {
            const int kgcap = (c.TB * tas == 1) ? 16 : 8;
            for (int kg = (c.Cin_g < kgcap ? c.Cin_g : kgcap); kg >= 1; --kg) {
                if (c.Cin_g % kg) continue;
                ConvK t = c;
                t.KG = kg; t.nchunks = t.Cin_g / kg; t.TAS = tas; t.nstg = t.TA / tas;
                int P = kg; while (P % 4 != 2) ++P;      // pixel pitch in 16-B slots: conflict-free ds_read_b128 fragments
                if (E == 4) P = kg + 1;
                t.pitch = P * 16;
                const int real = tas * t.TB * kg;
                t.nslots = (E == 8) ? ((real + 3) & ~3) : real;
                int KGP = 1; while (KGP < kg) KGP <<= 1;
                const size_t patch = (size_t)t.PH * t.PW * t.pitch;
                t.stage_bytes = t.nslots * NT * 16 * 16;
                const size_t wbytes = ((size_t)t.stage_bytes + 1023) & ~(size_t)1023;
                t.npieces = (int)(wbytes / 1024);
                t.off_w0 = (int)((patch + 1023) & ~(size_t)1023);
                const bool one_w = (t.nchunks == 1 && t.nstg == 1 && t.Yblocks == 1 && !s2c);      // resident weights: no second buffer (classes of a merged stride-2 launch have their own weights)
                t.off_w1 = t.off_w0 + (one_w ? 0 : (int)wbytes);
                t.off_ktab = t.off_w1 + (int)wbytes;
                t.off_red = t.off_ktab + ((t.nslots * 4 + 15) & ~15);
                t.off_dummy = t.off_red + 2 * nw * 2 * NT * 16 * 4;
                const size_t l = (size_t)t.off_dummy + 16;
                const int pm = ceil_div(t.PH * t.PW, (nw * 64) / KGP);
                // register budget (no spills: a spill in the prefetch path serialises it): wide N tiles keep fewer
                // pixel tiles / prefetch pieces per thread
                int pm_max = mt == 4 ? (NT == 6 ? 0 : NT == 4 ? 3 : 6) : (NT == 6 ? 4 : 10);
                if (nw == 8) pm_max = mt == 4 ? (NT == 6 ? 0 : 3) : (NT <= 4 ? 6 : 4);      // (the 8-wave instantiations: <MT 4, PMAX 3>, <MT 2, PMAX 4> and,
                                                                                              //  for NT <= 4, <MT 2, PMAX 6>: a 64-channel 3x3 then keeps ALL its weights
                                                                                              //  in LDS and runs one stage of 144 MFMAs per wave and tile)
                const size_t lds_cap = nw == 8 ? (size_t)156 * 1024 : (size_t)g_conv_lds_kb * 1024;
                if (l > lds_cap || pm > pm_max || t.nslots > 256) continue;
                if (want_fa && !conv_variant_fa(d->dtype, NT, mt, pm, nw)) continue;
                const double mf = (E == 8 ? t.nslots / 4 : t.nslots) * mt * NT;      // MFMAs per wave per stage
                const double score = eff * fill * ((double)real / t.nslots) * (mf / (mf + 40.0)) * (mt == 4 ? 1.0 : 0.93) * (nw == 8 ? nw8w / 100.0 : 1.0);
                if (score > best_score) { best_score = score; bk = t; bMT = mt; bPM = pm; blds = l; bNW = nw; }
            }
        }
    }
    }
    const bool ok = best_score > 0;
    if (!ok && want_fa) return conv_setup(d, k, NT, MT, PM, lds, grid, NWsel, 1);
    k = bk; MT = bMT; PM = bPM; lds = blds; NWsel = bNW;
    if (!ok) return MFC_ERR_UNSUPPORTED;
    k.ntiles = k.N * k.tilesY * k.tilesX;
    k.ncls = s2c ? 4 : 1;
    k.cls_bytes = (long)k.nstg * k.nchunks * k.Yblocks * k.stage_bytes;
    k.nunits = k.ntiles * k.Yblocks * k.ncls;
    k.ablate = g_conv_ablate;
    k.wt = 0;          // (measured: this family is SLOWER with write-through stores at every threshold -- 15.78 -> 16.05 ms per serial step; common.h)
    {   // cout-block-fastest order when the Yblocks passes over the input would otherwise each stream it from HBM again
        const double in_bytes = (double)k.N * k.Hin * k.Win * k.Cin_p * (mfc_is16(d->dtype) ? 2.0 : 4.0);
        k.ybfast = g_conv_ybfast >= 0 ? (g_conv_ybfast && k.Yblocks > 1) : (k.Yblocks > 1 && in_bytes * (k.Yblocks - 1) > 128e6);
    }
    k.wres = 0;
    if (g_conv_wres && k.nchunks == 1 && k.nstg == 1 && k.Yblocks > 1 && !s2c) {
        // every cout block's weights fit next to the patch: keep them all in LDS and stage each pixel tile once
        const int wb = k.npieces * 1024;
        const int ktab = k.off_w0 + k.Yblocks * wb, redo = ktab + (k.off_red - k.off_ktab), dum = redo + (k.off_dummy - k.off_red);
        const size_t cap = NWsel == 8 ? (size_t)156 * 1024 : (size_t)g_conv_lds_kb * 1024;
        if ((size_t)dum + 16 <= cap) {
            k.wres = k.Yblocks; k.ybfast = 1;
            k.off_ktab = ktab; k.off_red = redo; k.off_dummy = dum;
            if ((size_t)dum + 16 > lds) lds = (size_t)dum + 16;
        }
    }
    grid = NWsel == 8 ? g_conv_grid / 2 : g_conv_grid;
    if (grid > k.nunits) grid = k.nunits;
    k.per_block = ceil_div(k.nunits, grid);
    grid = ceil_div(k.nunits, k.per_block);
    return MFC_OK;
}

// The instantiation conv_dispatch picks for a geometry: its template PMAX (0 = none) and whether it is an FA variant (the epilogue
// with the transposed 8-channel layout, the only one that implements acc_src / bn_y; even NT so that every cout tile has a partner).
static bool conv_variant_fa(int dtype, int NT, int MT, int PM, int NW) {
    int PMAXt = 0;
    if (mfc_is16(dtype) && NW == 8) PMAXt = (NT != 6 && MT == 4 && PM <= 3) ? 3 : ((MT == 2 && PM <= 4) ? 4 : ((MT == 2 && NT <= 4 && PM <= 6) ? 6 : 0));
    else if (NT == 6) PMAXt = (MT == 2 && PM <= 4) ? 4 : 0;
    else if (MT == 4) PMAXt = PM <= 3 ? 3 : ((NT <= 3 && PM <= 6) ? 6 : 0);
    else PMAXt = PM <= 4 ? 4 : 10;
    if (!PMAXt || !mfc_is16(dtype) || (NT != 2 && NT != 4)) return false;
    if (NT == 4 && MT == 2 && PMAXt == 10) return false;          // (<4,2,10,4> with the fusions compiled in spills: not offered)
    return (NT * MT <= 8) || (MT == 2 && NT <= 4 && PMAXt <= 4);
}

template <typename T, int NT, int MT, int PMAX, int NW, bool FUSE = false>
static int conv_launch(const ConvK& k, size_t lds, int grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, NT, MT, PMAX, NW, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        MFC_PROF_NAME(pname, "conv_igemm_kernel<%s, %d, %d, %d, %d, %s>", mfc_tname<T>(), NT, MT, PMAX, NW, FUSE ? "true" : "false");
        const double E = sizeof(T) == 2 ? 8.0 : 4.0;
        // (a merged stride-2 data gradient computes 4 classes x 4 taps, of which 9 are real: the algorithmic work is the 9)
        const double flops = 2.0 * k.N * k.Hl * k.Wl * (double)k.Cout * (k.ncls > 1 ? 9.0 : (double)(k.TA * k.TB)) * (k.Cin_g * E);
        const double bytes = k.ncls > 1 ? ((double)k.N * k.Hin * k.Win * k.Cin_g * 16.0 + (double)k.N * k.Hout * k.Wout * k.Cout_p * sizeof(T))
                                        : (((double)k.N * k.Hin * k.Win * k.Cin_g * 16.0) / (k.osh * k.osw) + (double)k.N * k.Hl * k.Wl * k.Cout_p * sizeof(T));
        mfc_prof_before(st, pname, flops, bytes);
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, NT, MT, PMAX, NW, FUSE>), dim3(grid), dim3(NW * 64), lds, st, k);
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// conv3x3_ring.hip
bool ring_eligible(const mfc_conv_desc* d);
int ring_layout(const mfc_conv_desc* d, mfc_conv_layout* out);
int ring_launch(const mfc_conv_desc* d, hipStream_t st);
// conv_gemm1x1.hip
bool gemm1x1_eligible(const mfc_conv_desc* d);
int gemm1x1_layout(const mfc_conv_desc* d, mfc_conv_layout* out);
int gemm1x1_launch(const mfc_conv_desc* d, hipStream_t st);

extern "C" int mfc_conv2d_layout(const mfc_conv_desc* d, mfc_conv_layout* out) {
    if (!d) return MFC_ERR_INVALID_ARG;
    if (out && ring_eligible(d)) return ring_layout(d, out);
    if (out && gemm1x1_eligible(d)) return gemm1x1_layout(d, out);
    ConvK k; int NT, MT, PM, grid, NW; size_t lds;
    mfc_conv_desc t = *d;
    if (!t.in) t.in = (const void*)16;
    if (!t.wp) t.wp = (const void*)16;
    if (!t.out) t.out = (void*)16;
    int rc = conv_setup(&t, k, NT, MT, PM, lds, grid, NW);
    if (rc < 0 || !out) return rc < 0 ? rc : MFC_ERR_INVALID_ARG;
    out->KG = k.KG; out->nchunks = k.nchunks; out->NT16 = NT * 16; out->Yblocks = k.Yblocks; out->nslots = k.nslots;
    out->TA = k.TA; out->TB = k.TB; out->lds_bytes = (int32_t)lds; out->TAS = k.TAS;
    out->bytes = (int64_t)k.nstg * k.nchunks * k.Yblocks * k.stage_bytes * k.ncls;        // (a merged stride-2 data gradient: four class images, one after the other)
    out->MT = MT; out->TH = k.TH; out->TW = k.TW; out->grid = grid; out->per_block = k.per_block; out->NW = NW;
    out->fa = conv_variant_fa(d->dtype, NT, MT, PM, NW) ? 1 : 0;
    return MFC_OK;
}

extern "C" int mfc_conv2d_lds_bytes(const mfc_conv_desc* d) {
    if (!d) return MFC_ERR_INVALID_ARG;
    if (ring_eligible(d)) { mfc_conv_layout l; const int rc = ring_layout(d, &l); return rc < 0 ? rc : l.lds_bytes; }
    if (gemm1x1_eligible(d)) { mfc_conv_layout l; gemm1x1_layout(d, &l); return l.lds_bytes; }
    ConvK k; int NT, MT, PM, grid, NW; size_t lds;
    int rc = conv_setup(d, k, NT, MT, PM, lds, grid, NW);
    return rc < 0 ? rc : (int)lds;
}

// FA instantiations with the epilogue fusions compiled in (16-bit types, even NT): exactly the set conv_variant_fa() accepts
template <typename T, int NT>
static int conv_dispatch_fused(const ConvK& k, int MT, int PM, int NW, size_t lds, int grid, hipStream_t st) {
    if constexpr (NT == 2 || NT == 4) {
        if (NW == 8) {
            if constexpr (NT == 2) { if (MT == 4 && PM <= 3) return conv_launch<T, NT, 4, 3, 8, true>(k, lds, grid, st); }
            if (MT == 2 && PM <= 4) return conv_launch<T, NT, 2, 4, 8, true>(k, lds, grid, st);
            if (MT == 2 && PM <= 6) return conv_launch<T, NT, 2, 6, 8, true>(k, lds, grid, st);
            return MFC_ERR_UNSUPPORTED;
        }
        if (MT == 4) {
            if constexpr (NT == 2) {
                if (PM <= 3) return conv_launch<T, NT, 4, 3, 4, true>(k, lds, grid, st);
                if (PM <= 6) return conv_launch<T, NT, 4, 6, 4, true>(k, lds, grid, st);
            }
            return MFC_ERR_UNSUPPORTED;
        }
        if (PM <= 4) return conv_launch<T, NT, 2, 4, 4, true>(k, lds, grid, st);
        if constexpr (NT == 2) return conv_launch<T, NT, 2, 10, 4, true>(k, lds, grid, st);
    }
    return MFC_ERR_UNSUPPORTED;
}

template <typename T, int NT>
static int conv_dispatch(const ConvK& k, int MT, int PM, int NW, size_t lds, int grid, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (NW == 8) {       // one workgroup per CU (bf16 only)
            if constexpr (NT == 6) {
                if (MT == 2 && PM <= 4) return conv_launch<T, NT, 2, 4, 8>(k, lds, grid, st);
                return MFC_ERR_UNSUPPORTED;
            } else {
                if (MT == 4 && PM <= 3) return conv_launch<T, NT, 4, 3, 8>(k, lds, grid, st);
                if (MT == 2 && PM <= 4) return conv_launch<T, NT, 2, 4, 8>(k, lds, grid, st);
                if constexpr (NT <= 4) { if (MT == 2 && PM <= 6) return conv_launch<T, NT, 2, 6, 8>(k, lds, grid, st); }
                return MFC_ERR_UNSUPPORTED;
            }
        }
    }
    if constexpr (NT == 6) {
        if (MT == 2 && PM <= 4) return conv_launch<T, NT, 2, 4, 4>(k, lds, grid, st);
        return MFC_ERR_UNSUPPORTED;
    } else {
        if (MT == 4) {
            if (PM <= 3) return conv_launch<T, NT, 4, 3, 4>(k, lds, grid, st);
            if constexpr (NT <= 3) { if (PM <= 6) return conv_launch<T, NT, 4, 6, 4>(k, lds, grid, st); }
            return MFC_ERR_UNSUPPORTED;
        }
        if (PM <= 4) return conv_launch<T, NT, 2, 4, 4>(k, lds, grid, st);
        return conv_launch<T, NT, 2, 10, 4>(k, lds, grid, st);
    }
}

extern "C" int mfc_conv2d_fwd(const mfc_conv_desc* d, void* stream) {
    if (!d) return MFC_ERR_INVALID_ARG;
    if (!mfc_ptrs_ok(d->in, d->wp, d->out, d->bias, d->in_coef, d->out_stats, d->acc_src, d->bn_y, d->bn_coef, d->bn_bits)) return MFC_ERR_INVALID_ARG;
    if (ring_eligible(d)) return ring_launch(d, (hipStream_t)stream);
    if ((d->flags & MFC_CONV_NEVER_ACC) && (d->accumulate || d->acc_src)) return MFC_ERR_INVALID_ARG;
    const bool fused = d && (d->acc_src || d->bn_y);
    if (d && gemm1x1_eligible(d)) return gemm1x1_launch(d, (hipStream_t)stream);
    ConvK k; int NT, MT, PM, grid, NW; size_t lds;
    int rc = conv_setup(d, k, NT, MT, PM, lds, grid, NW);
    if (rc < 0) return rc;
    if (fused) {      // epilogue fusions: only where mfc_conv2d_layout reported fa = 1; never silently dropped
        if (!conv_variant_fa(d->dtype, NT, MT, PM, NW)) return MFC_ERR_UNSUPPORTED;
        if (d->acc_src && !d->accumulate) return MFC_ERR_INVALID_ARG;
        if (d->bn_y && (!d->bn_coef || !d->out_stats || (d->bn_mask_mode != 0 && d->bn_mask_mode != 2 && d->bn_mask_mode != 3) ||
                        (d->bn_mask_mode == 3 && !d->bn_bits) || d->bias)) return MFC_ERR_INVALID_ARG;
        hipStream_t stf = (hipStream_t)stream;
        int rcf = MFC_ERR_UNSUPPORTED;
        if (NT == 2) MFC_TYPED16(d->dtype, T_, rcf = conv_dispatch_fused<T_, 2>(k, MT, PM, NW, lds, grid, stf));
        else if (NT == 4) MFC_TYPED16(d->dtype, T_, rcf = conv_dispatch_fused<T_, 4>(k, MT, PM, NW, lds, grid, stf));
        return rcf;
    }
    hipStream_t st = (hipStream_t)stream;
#define MFC_CONV_CASE(nt) \
    case nt: { int rcd; MFC_TYPED(d->dtype, T_, rcd = conv_dispatch<T_, nt>(k, MT, PM, NW, lds, grid, st)); return rcd; }
    switch (NT) {
        MFC_CONV_CASE(1) MFC_CONV_CASE(2) MFC_CONV_CASE(3) MFC_CONV_CASE(4) MFC_CONV_CASE(6)
    }
#undef MFC_CONV_CASE
    return MFC_ERR_UNSUPPORTED;
}
