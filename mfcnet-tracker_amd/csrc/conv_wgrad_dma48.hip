// Weight gradient of the 3x3 / stride-1 / pad-1 convolutions whose channel counts are multiples of 48 but not of 32 -- the 48 -> 48 BasicBlocks
// of HRNet-W48's 120x160 branch (models/hrnet.py:58-74, 297-333; 64 launches of a W48 step: 61 / 88 us each on the register-staged wave
// kernel, 4.8 ms of a 72 ms serial step).
//   dWp[a,b][co][ci] = sum_{n,i,j} dy[n,i,j,co] * f(x[n, i-1+a, j-1+b, ci])
// conv_wgrad_dma.hip gives every WAVE a private staging ring and a 32 x 32 channel block; with 48 channels a wave cannot hold the block (81
// accumulator tiles), and private 48 x 16 blocks move 7 KB of DMA per 27 MFMAs -- the LDS-DMA rate of a CU (measured slower than the wave
// kernel, profiles/r03_wgrad_dma48.txt).  Here the WORKGROUP shares one ring: a stage holds the dy sub-tile (4 x 8 pixels x 48 couts) and the
// 6 x 10 patch (48 cins) ONCE, and three waves each multiply it by their own third of the input channels (27 accumulator tiles each:
// 81 MFMAs per 10.5 KB of DMA).  Protocol of conv3x3_ring.hip: all four waves issue the DMA pieces of the sub-tile two ahead
// (global_load_lds_dwordx4), the compute waves run their MFMAs, every wave waits for ITS OWN pieces of the next sub-tile (counted
// s_waitcnt vmcnt), fixes them up in place (zero what lies outside the image; the producer's BatchNorm + ReLU) and ONE raw s_barrier ends
// the step.  No accumulator reduction at the end: the three waves own disjoint parts of the 48 x 48 block.
//   * LDS layouts (tools/probe/c48_banks.py, brute force over taps, thirds and both transpose reads): dy rows of 8 pixels x 96 B at a pitch
//     of 896 B, patch rows of 10 pixels x 96 B at a pitch of 1152 B: conflict free for ds_read_b64_tr_b16 without any swizzle; the padding
//     slots of a (lane-contiguous) DMA piece fetch a valid granule nobody reads; LDS slot s of a region IS byte 16 s of it.
#include "common.h"

int g_wgrad_dma48_x2 = 0;            // mfc_set_flag(48, 1): twice the workgroups (and partial-sum slices) per launch
int g_wgrad_dma48 = 1;               // mfc_set_flag(47, v): 0 = register-staged wave kernel (conv_wgrad.hip) for these launches
extern int g_wgrad_blocks;           // target workgroups per launch (conv_wgrad.hip, mfc_set_flag(11))

#define W48_DPITCH 896               // bytes between dy tile rows (8 pixels x 96 B + 128)
#define W48_DSLOTS 56
#define W48_DP 4                     // DMA pieces of the dy sub-tile (4 x 896 = 3584 B of 4096)
#define W48_XPITCH 1152              // bytes between patch rows (10 pixels x 96 B + 192)
#define W48_XSLOTS 72
#define W48_XP 7                     // DMA pieces of the patch (6 x 1152 = 6912 B of 7168)
#define W48_NP (W48_DP + W48_XP)     // 11 pieces per sub-tile, piece q issued by wave q & 3
#define W48_DBYTES (W48_DP * 1024)
#define W48_STAGE (W48_NP * 1024)
#define W48_NST 3

struct Wgrad48 {
    const char* x; const char* dy; float* dwp; const float* in_coef;
    int N, H, W, Cin_p, Cout_p;
    int in_relu, ipg, G;
    int tilesY, tilesX, ntiles, splits;
    int Co16, Ci16, co_blocks, ci_blocks;
    int slice;                       // floats per partial-sum slice of dwp
};

__device__ inline void w48_dma(const char* base, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds) : "memory");
}
__device__ inline void w48_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // (leaves the DMAs in flight)

template <typename TE, bool XF>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma48_kernel(Wgrad48 p) {
    typedef TE T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* coefs = (float*)(smem + W48_NST * W48_STAGE);      // [G][2][48] scale / shift of this block's input channels
    const int Ytot = p.co_blocks * p.ci_blocks;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int y = Lb % Ytot, bsplit = Lb / Ytot;
    const int ib = y % p.ci_blocks, cb = y / p.ci_blocks;
    const int co0 = cb * 48, ci0 = ib * 48;
    if constexpr (XF) {
        for (int i = tid; i < p.G * 96; i += 256) {
            const int g = i / 96, r = i - g * 96, w = r / 48, ch = r - w * 48;
            coefs[i] = p.in_coef[((size_t)g * 4 + w) * p.Cin_p + ci0 + ch];
        }
    }
    const int drow = p.Cout_p * 2, xrow = p.Cin_p * 2;
    const int npw = (W48_NP - wave + 3) / 4;                 // pieces this wave issues per sub-tile: 3, 3, 3, 2 (wave-uniform)
    // ---- slot -> (pixel, granule) of piece q = wave + 4 i; slot s of a region is byte 16 s of it
    auto d_slot = [&](int pq, int ln, int& ty, int& tx, int& g) {
        const int s = 64 * pq + ln;
        const int q = s / W48_DSLOTS, rem = s - q * W48_DSLOTS;
        ty = min(q, 3);
        tx = min(rem / 6, 7);
        g = rem < 48 ? rem - (rem / 6) * 6 : 0;
        return q <= 3 && rem < 48;                          // a real slot (not row padding, not the tail of the last piece)
    };
    auto x_slot = [&](int pq, int ln, int& py, int& px, int& g) {
        const int s = 64 * pq + ln;
        const int q = s / W48_XSLOTS, rem = s - q * W48_XSLOTS;
        py = min(q, 5);
        px = min(rem / 6, 9);
        g = rem < 60 ? rem - (rem / 6) * 6 : 0;
        return q <= 5 && rem < 60;
    };
    int vof[3];                                             // interior tiles: byte offset of this lane's granule from the tile's first pixel
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = wave + 4 * i;
        int a, b, g;
        if (q < W48_DP) { d_slot(q, lane, a, b, g); vof[i] = (a * p.W + b) * drow + g * 16; }
        else { x_slot(min(q, W48_NP - 1) - W48_DP, lane, a, b, g); vof[i] = (a * p.W + b) * xrow + g * 16; }
    }

    f32x4 acc[9][3];
#pragma unroll
    for (int b = 0; b < 9; ++b)
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[b][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // sub-tile cursor of the WORKGROUP (4 x 8 output pixels), tracked incrementally
    const int stride = p.splits;
    const int st_x = stride % p.tilesX, st_y = (stride / p.tilesX) % p.tilesY, st_n = stride / (p.tilesX * p.tilesY);
    struct TC { int n, tyi, txi; };
    auto tc_next = [&](TC c) {
        c.txi += st_x; if (c.txi >= p.tilesX) { c.txi -= p.tilesX; ++c.tyi; }
        c.tyi += st_y; if (c.tyi >= p.tilesY) { c.tyi -= p.tilesY; ++c.n; }
        c.n += st_n;
        return c;
    };
    auto is_interior = [&](const TC& c) {
        const int i0 = c.tyi * 4, j0 = c.txi * 8;
        return i0 >= 1 && j0 >= 1 && i0 + 5 <= p.H && j0 + 9 <= p.W;
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](const TC& c, int slot) {
        const int i0 = c.tyi * 4, j0 = c.txi * 8;
        const unsigned lds = lds0 + slot * W48_STAGE;
        const char* dimg = p.dy + ((size_t)c.n * p.H * p.W * p.Cout_p + co0) * sizeof(T);
        const char* ximg = p.x + ((size_t)c.n * p.H * p.W * p.Cin_p + ci0) * sizeof(T);
        if (is_interior(c)) {           // workgroup-uniform
            const char* dt = dimg + (size_t)(i0 * p.W + j0) * drow;
            const char* xt = ximg + (size_t)((i0 - 1) * p.W + (j0 - 1)) * xrow;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = wave + 4 * i;
                if (i < npw) w48_dma(q < W48_DP ? dt : xt, (unsigned)vof[i], lds + 1024 * q);
            }
        } else {                        // clamped (always valid) addresses; the out-of-image slots are zeroed after landing
            int ln = lane;
            asm volatile("" : "+v"(ln));          // (keeps the recomputed slot coordinates inside this branch)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = wave + 4 * i;
                if (i < npw) {
                    int a, b, g;
                    if (q < W48_DP) {
                        d_slot(q, ln, a, b, g);
                        const int oy = min(i0 + a, p.H - 1), ox = min(j0 + b, p.W - 1);
                        w48_dma(dimg, (unsigned)((oy * p.W + ox) * drow + g * 16), lds + 1024 * q);
                    } else {
                        x_slot(q - W48_DP, ln, a, b, g);
                        const int iy = min(max(i0 - 1 + a, 0), p.H - 1), ix = min(max(j0 - 1 + b, 0), p.W - 1);
                        w48_dma(ximg, (unsigned)((iy * p.W + ix) * xrow + g * 16), lds + 1024 * q);
                    }
                }
            }
        }
    };
    // all but this wave's k youngest vector-memory operations are done
    auto vm_wait = [&](int k) {
        if (k >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (k == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // ---- in-LDS fix-up of the pieces THIS wave issued into `slot` (after its own counted wait: no barrier between landing and fix-up)
    const float relu_floor = p.in_relu ? 0.f : -3.0e38f;
    auto fixup = [&](const TC& c, int slot) {
        const bool interior = is_interior(c);
        if (!XF && interior) return;
        const int i0 = c.tyi * 4, j0 = c.txi * 8;
        char* buf = smem + slot * W48_STAGE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = wave + 4 * i;
            if (i < npw) {
                char* a = buf + 1024 * q + lane * 16;
                int u, v, g;
                if (q < W48_DP) {
                    const bool real = d_slot(q, lane, u, v, g);
                    if (!interior && real && (i0 + u >= p.H || j0 + v >= p.W)) *(uint4*)a = make_uint4(0, 0, 0, 0);
                } else {
                    const bool real = x_slot(q - W48_DP, lane, u, v, g);
                    const int iy = i0 - 1 + u, ix = j0 - 1 + v;
                    const bool inr = interior || ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W);
                    if constexpr (XF) {
                        if (real) {
                            const float* cf = coefs + (c.n / p.ipg) * 96 + g * 8;
                            const float4 s0 = *(const float4*)cf, s1 = *(const float4*)(cf + 4), h0 = *(const float4*)(cf + 48), h1 = *(const float4*)(cf + 52);
                            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                            uint4 val = *(const uint4*)a;
                            float f[8];
                            Gran<T>::unpack(val, f);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] * sc[e] + sh[e], relu_floor);      // (v_max swallows NaN: see conv_wgrad.hip store_tile)
                            val = Gran<T>::pack(f);
                            *(uint4*)a = inr ? val : make_uint4(0, 0, 0, 0);
                        }
                    } else {
                        if (real && !inr) *(uint4*)a = make_uint4(0, 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- MFMAs of one landed slot (waves 0-2: input-channel third `wave`).  k index of lane = pixel 8 kq + r (+4 for the second read)
    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int kq = lane >> 4, r = (lane & 15) >> 2, csub = (lane & 3) * 8;
    const int dbase = kq * W48_DPITCH + r * 96 + csub;
    const int xbase = W48_DBYTES + kq * W48_XPITCH + r * 96 + min(wave, 2) * 32 + csub;
    auto compute = [&](int slot) {
        const char* buf = smem + slot * W48_STAGE;
        bf16x8 af[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const char* a = buf + dbase + i * 32;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)a);
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a + 4 * 96));
            af[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int b = 0; b < 9; ++b) {
            const int ta = b / 3, tb = b - 3 * ta;
            const char* a = buf + xbase + ta * W48_XPITCH + tb * 96;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)a);
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a + 4 * 96));
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[b][i] = mfma16<T>(af[i], bfr, acc[b][i]);
        }
    };

    int tile = bsplit;
    const int nmine = tile < p.ntiles ? (p.ntiles - tile + stride - 1) / stride : 0;      // sub-tiles of this workgroup (workgroup-uniform)
    TC t0;
    { t0.txi = tile % p.tilesX; const int q = tile / p.tilesX; t0.tyi = q % p.tilesY; t0.n = q / p.tilesY; }
    TC t1 = tc_next(t0), t2 = tc_next(t1);
    if (nmine > 0) {
        // ---- prologue: two slots in flight
        issue(t0, 0);
        if (nmine > 1) issue(t1, 1);
        w48_barrier();                                       // coefficient table visible (its ds_writes are waited for; the DMAs stay in flight)
        vm_wait(nmine > 1 ? npw : 0);
        fixup(t0, 0);
        w48_barrier();
        int slot = 0;
        for (int it = 0; it < nmine; ++it) {
            const bool has1 = it + 1 < nmine, has2 = it + 2 < nmine;
            int s1 = slot + 1; if (s1 >= W48_NST) s1 -= W48_NST;
            int s2 = slot + 2; if (s2 >= W48_NST) s2 -= W48_NST;
            // (slot s2 was read by the MFMAs of the previous sub-tile; every wave has passed the barrier behind them)
            if (has2) issue(t2, s2);
            if (wave < 3) compute(slot);
            if (has1) {
                vm_wait(has2 ? npw : 0);                     // this wave's pieces of the next sub-tile have landed
                fixup(t1, s1);
            }
            w48_barrier();                                   // everybody's pieces of the next sub-tile are in place; everybody is done with `slot`
            t0 = t1; t1 = t2; t2 = tc_next(t2);
            slot = s1;
        }
    }
    // ---- the three compute waves own disjoint thirds of the 48 x 48 block: plain stores of the partial sums, no reduction
    if (wave < 3) {
#pragma unroll
        for (int b = 0; b < 9; ++b)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int co = co0 + i * 16 + (lane >> 4) * 4, ci = ci0 + wave * 16 + (lane & 15);
                float* o = p.dwp + (size_t)bsplit * p.slice + ((size_t)b * p.Co16 + co) * p.Ci16 + ci;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) o[(size_t)rr * p.Ci16] = acc[b][i][rr];
            }
    }
}

bool wgrad_dma48_eligible(const mfc_wgrad_desc* d) {
    if (!g_wgrad_dma48 || !mfc_is16(d->dtype) || d->batch > 1) return false;
    if (d->TA != 3 || d->TB != 3 || d->in_stride != 1 || d->dh0 != -1 || d->dw0 != -1) return false;
    if (d->Hin != d->Hout || d->Win != d->Wout) return false;
    if (d->Cout % 48 || d->Cin % 48 || d->Cin_p != d->Cin || d->Cout_p != d->Cout) return false;       // whole 48 x 48 blocks
    if (d->Cin % 32 == 0 && d->Cout % 32 == 0) return false;                                            // (conv_wgrad_dma.hip's 32 x 32 blocks fit)
    if (d->N / d->images_per_group > 8) return false;
    if ((double)d->Hin * d->Win * (d->Cin_p > d->Cout_p ? d->Cin_p : d->Cout_p) * 2.0 >= 2.0e9) return false;      // 32-bit lane offsets inside one image
    return true;
}

static void wgrad_dma48_setup(const mfc_wgrad_desc* d, Wgrad48& f) {
    f.x = (const char*)d->x; f.dy = (const char*)d->dy; f.dwp = d->dwp; f.in_coef = d->in_coef;
    f.N = d->N; f.H = d->Hin; f.W = d->Win; f.Cin_p = d->Cin_p; f.Cout_p = d->Cout_p;
    f.in_relu = d->in_relu; f.ipg = d->images_per_group; f.G = d->N / d->images_per_group;
    f.tilesY = ceil_div(f.H, 4); f.tilesX = ceil_div(f.W, 8);
    f.ntiles = f.N * f.tilesY * f.tilesX;
    f.Co16 = d->Cout; f.Ci16 = d->Cin;
    f.co_blocks = d->Cout / 48; f.ci_blocks = d->Cin / 48;
    const int Y = f.co_blocks * f.ci_blocks;
    int S = d->splits;
    if (S <= 0) S = ceil_div(g_wgrad_dma48_x2 ? 2 * g_wgrad_blocks : g_wgrad_blocks, Y);
    if (S > f.ntiles) S = f.ntiles;
    if (S < 1) S = 1;
    f.splits = S;
    f.slice = 9 * f.Co16 * f.Ci16;
}

int wgrad_dma48_parts(const mfc_wgrad_desc* d) { Wgrad48 f; wgrad_dma48_setup(d, f); return f.splits; }

int wgrad_dma48_launch(const mfc_wgrad_desc* d, hipStream_t st) {
    Wgrad48 f; wgrad_dma48_setup(d, f);
    const bool xf = d->in_coef != nullptr;
    const size_t lds = (size_t)W48_NST * W48_STAGE + (xf ? (size_t)f.G * 96 * 4 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma48_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma48_kernel<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma48_kernel<f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma48_kernel<f16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        const double flops = 2.0 * f.N * f.H * f.W * (double)f.Co16 * f.Ci16 * 9.0;
        const double bytes = ((double)f.N * f.H * f.W * (f.Cin_p + f.Cout_p)) * 2.0;
        const bool h = d->dtype == MFC_F16;
        mfc_prof_before(st, h ? (xf ? "conv_wgrad_dma48_kernel<_Float16, true>" : "conv_wgrad_dma48_kernel<_Float16, false>")
                              : (xf ? "conv_wgrad_dma48_kernel<__bf16, true>" : "conv_wgrad_dma48_kernel<__bf16, false>"), flops, bytes);
    }
    const int grid = f.splits * f.co_blocks * f.ci_blocks;
    if (xf) MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL((conv_wgrad_dma48_kernel<T_, true>), dim3(grid), dim3(256), lds, st, f));
    else MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL((conv_wgrad_dma48_kernel<T_, false>), dim3(grid), dim3(256), lds, st, f));
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}
