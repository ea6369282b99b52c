// Weight gradient of the 3x3 / stride-1 / pad-1 encoder convolutions (bf16), LDS-DMA fed.
//   dWp[a,b][co][ci] = sum_{n,i,j} dy[n,i,j,co] * f(x[n, i-1+a, j-1+b, ci])          (models/hrnet.py:58-74 under loss.backward())
// Same arithmetic decomposition as conv_wgrad_wave_kernel<3,3,2,2,..> (GEMM M = cout, N = 9 x cin, K = pixels; a workgroup owns a
// 32 x 32 channel block for ALL nine taps, each of its 4 waves walks its own 4 x 8-pixel sub-tiles = one MFMA k-step and keeps the
// 36 accumulator tiles in registers), but the operands no longer pass through registers on their way to LDS:
//   * every wave owns a ring of THREE staging slots; a slot is filled by six global_load_lds_dwordx4 (2 for the dy sub-tile, 4 for
//     the 6 x 10 input patch) issued TWO sub-tiles ahead of the MFMAs that read it, so two slots' worth of loads (12 KiB per wave) is
//     always in flight -- the register-staged kernel had one sub-tile in flight per wave and spent 84 % of its time waiting for it;
//   * no workgroup barrier and no VALU address arithmetic in the main loop (interior tiles: scalar tile base + constant lane offset);
//     completion is a counted s_waitcnt vmcnt(12);
//   * an LDS-DMA writes 64 x 16 contiguous bytes, so pixel rows cannot be padded against bank conflicts; instead a pixel row is
//     exactly 64 B and the two 32-byte channel halves of the pixels of ODD tile / patch rows are swapped (on the source side of the
//     DMA: each lane fetches the granule that belongs in its slot, free).  With an 8-pixel-wide tile the 32 lanes of a
//     ds_read_b64_tr_b16 group read two consecutive rows, i.e. both halves of every 64-byte bank slot: conflict free for every tap;
//   * out-of-image halo / tail pixels are fetched from a clamped address and zeroed in LDS after landing (edge tiles only); the fused
//     BatchNorm + ReLU of the producer is applied in place in LDS by the wave that will read the slot.
// Partial sums leave the workgroup as one fp32 slice (mfc_conv2d_wgrad_parts / mfc_unpack_wgrad), as in conv_wgrad.hip.
#include "common.h"
#include <cstdlib>

int g_wgrad_dma_xf8 = 0;             // mfc_set_flag(38, 1): 8-wave workgroups for the launches with an input transform (faster alone, slower in the step: see DESIGN.md)
int g_wgrad_dma = 1;                 // mfc_set_flag(29, v): 0 = register-staged wave kernel (conv_wgrad.hip) for these launches
extern int g_wgrad_blocks;           // target workgroups per launch (conv_wgrad.hip, mfc_set_flag(11))

#define WD_PW 10
#define WD_NPX 60
#define WD_DBYTES 2048
#define WD_XBYTES 4096               // 60 pixels x 64 B = 3840 B + a 256-byte tail the last DMA piece spills into
#define WD_STAGE (WD_DBYTES + WD_XBYTES)
#define WD_NST 3
#define WD_WAVE (WD_NST * WD_STAGE)

struct WgradD {
    const char* x; const char* dy; float* dwp; const float* in_coef;
    int N, H, W, Cin_p, Cout_p;
    int in_relu, ipg, G;
    int tilesY, tilesX, ntiles, splits;
    int Co16, Ci16, co_blocks, ci_blocks;
    int slice;                       // floats per partial-sum slice of dwp
};

// one 1-KiB LDS-DMA piece: lane l copies 16 B from (base + voff) to LDS byte (lds + 16 l)
__device__ inline void wd_dma(const char* base, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds) : "memory");
}

// NW = waves per workgroup.  4: two workgroups fit a CU.  8 (input-transform variant): the in-LDS BatchNorm + ReLU of a sub-tile is VALU /
// LDS work of the wave that is about to multiply it, and with one wave per SIMD nothing runs under it (27.9 vs 18.7 us at 256
// workgroups); eight self-contained waves = two per SIMD let one wave's transform issue under the other's MFMAs without doubling the
// number of workgroups, i.e. of partial-sum slices the unpack has to add.
template <typename TE, bool XF, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void conv_wgrad_dma_kernel(WgradD p) {
    typedef TE T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* wbase = smem + wave * WD_WAVE;
    float* coefs = (float*)(smem + NW * WD_WAVE);          // [G][2][32] scale / shift of this block's input channels
    // 1-D grid, weight block fastest, XCD-contiguous (conv_wgrad.hip): the workgroups that walk the SAME pixel tiles run on one XCD
    const int Ytot = p.co_blocks * p.ci_blocks;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int y = Lb % Ytot, bsplit = Lb / Ytot;
    const int ib = y % p.ci_blocks, cb = y / p.ci_blocks;
    const int co0 = cb * 32, ci0 = ib * 32;
    if constexpr (XF) {
        for (int i = tid; i < p.G * 64; i += NW * 64) {
            const int ch = i & 31, w = (i >> 5) & 1, g = i >> 6;
            coefs[i] = (ci0 + ch < p.Cin_p) ? p.in_coef[((size_t)g * 4 + w) * p.Cin_p + ci0 + ch] : 0.f;
        }
    }
    // ---- per-lane DMA tables: slot s = 64 i + lane of a piece sequence -> (pixel, physical granule); the granule fetched is the
    //      logical one of that slot: physical ^ 2 on odd rows (the 32-byte halves of odd rows are swapped)
    int d_vof[2], x_vof[4]; unsigned d_pk[2], x_pk[4];
    const int drow = p.Cout_p * 2, xrow = p.Cin_p * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = 64 * i + lane, pp = s >> 2, gp = s & 3;
        const int ty = pp >> 3, tx = pp & 7, gl = gp ^ ((ty & 1) << 1);
        d_pk[i] = ty | (tx << 4) | (gl << 8);
        d_vof[i] = (ty * p.W + tx) * drow + gl * 16;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = 64 * i + lane, q = min(s >> 2, WD_NPX - 1), gp = s & 3;
        const int py = q / WD_PW, px = q - py * WD_PW, gl = gp ^ ((py & 1) << 1);
        x_pk[i] = py | (px << 4) | (gl << 8);
        x_vof[i] = (py * p.W + px) * xrow + gl * 16;
    }

    f32x4 acc[9][2][2];
#pragma unroll
    for (int b = 0; b < 9; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // tile coordinates, tracked incrementally (stride decomposed once)
    const int stride = p.splits * NW;
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
    auto issue = [&](const TC& c, int slot) {
        const int i0 = c.tyi * 4, j0 = c.txi * 8;
        const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(wbase + slot * WD_STAGE);
        const char* dimg = p.dy + ((size_t)c.n * p.H * p.W * p.Cout_p + co0) * sizeof(T);
        const char* ximg = p.x + ((size_t)c.n * p.H * p.W * p.Cin_p + ci0) * sizeof(T);
        if (is_interior(c)) {           // wave-uniform
            const char* dt = dimg + (size_t)(i0 * p.W + j0) * drow;
            const char* xt = ximg + (size_t)((i0 - 1) * p.W + (j0 - 1)) * xrow;
#pragma unroll
            for (int i = 0; i < 2; ++i) wd_dma(dt, (unsigned)d_vof[i], lds + 1024 * i);
#pragma unroll
            for (int i = 0; i < 4; ++i) wd_dma(xt, (unsigned)x_vof[i], lds + WD_DBYTES + 1024 * i);
        } else {                        // clamped (always valid) addresses; the out-of-image slots are zeroed after landing
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = min(i0 + (int)(d_pk[i] & 15), p.H - 1), ox = min(j0 + (int)((d_pk[i] >> 4) & 15), p.W - 1);
                wd_dma(dimg, (unsigned)((oy * p.W + ox) * drow + (int)(d_pk[i] >> 8) * 16), lds + 1024 * i);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int iy = min(max(i0 - 1 + (int)(x_pk[i] & 15), 0), p.H - 1), ix = min(max(j0 - 1 + (int)((x_pk[i] >> 4) & 15), 0), p.W - 1);
                wd_dma(ximg, (unsigned)((iy * p.W + ix) * xrow + (int)(x_pk[i] >> 8) * 16), lds + WD_DBYTES + 1024 * i);
            }
        }
    };

    // ---- in-LDS fix-up of a landed slot (the reading wave does it): zero what lies outside the image; apply the producer's BN + ReLU
    //      to the patch.  Patch pass mapping: lanes 0-31 take the even patch rows, 32-63 the odd ones, so a lane's logical granule
    //      (physical ^ 2 on odd rows) -- hence its 8 channels' coefficients -- is the same for all its pieces.
    const int xh = lane >> 5, xli = lane & 31;
    const int xg = xli & 3, xgl = xg ^ (xh << 1);
    const float relu_floor = p.in_relu ? 0.f : -3.0e38f;
    auto fixup = [&](const TC& c, int slot) {
        const bool interior = is_interior(c);
        if (!XF && interior) return;
        const int i0 = c.tyi * 4, j0 = c.txi * 8;
        char* buf = wbase + slot * WD_STAGE;
        if (!interior) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = 64 * i + lane, pp = s >> 2;
                if (i0 + (pp >> 3) >= p.H || j0 + (pp & 7) >= p.W) *(uint4*)(buf + s * 16) = make_uint4(0, 0, 0, 0);
            }
        }
        float sc[8], sh[8];
        if constexpr (XF) {
            const float* cf = coefs + (c.n / p.ipg) * 64 + xgl * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = cf[e]; sh[e] = cf[32 + e]; }
        }
        if constexpr (XF) {
            // all four granules are READ first and transformed afterwards: as read-modify-write chains one after the other the in-place stores
            // (which hipcc cannot prove disjoint from the next piece's load) put four LDS round trips in a row in front of every MFMA block
            uint4 vv[4]; char* aa[4]; bool okk[4], inn[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = xli + 32 * it;              // (pixel-in-parity-class, granule), granule = idx & 3 = xg
                const int pc = min(idx >> 2, 29);
                const int r3 = pc / WD_PW, px = pc - r3 * WD_PW, py = 2 * r3 + xh;
                aa[it] = buf + WD_DBYTES + (py * WD_PW + px) * 64 + xg * 16;
                const int iy = i0 - 1 + py, ix = j0 - 1 + px;
                okk[it] = (idx >> 2) < 30;
                inn[it] = interior || ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W);
                vv[it] = *(const uint4*)aa[it];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float f[8];
                Gran<T>::unpack(vv[it], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] * sc[e] + sh[e], relu_floor);      // (v_max swallows NaN: see conv_wgrad.hip store_tile)
                const uint4 v = Gran<T>::pack(f);
                if (okk[it]) *(uint4*)aa[it] = inn[it] ? v : make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = xli + 32 * it;
                const int pc = idx >> 2;
                if (pc < 30) {
                    const int r3 = pc / WD_PW, px = pc - r3 * WD_PW, py = 2 * r3 + xh;
                    char* a = buf + WD_DBYTES + (py * WD_PW + px) * 64 + xg * 16;
                    const int iy = i0 - 1 + py, ix = j0 - 1 + px;
                    const bool inr = interior || ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W);
                    if (!inr) *(uint4*)a = make_uint4(0, 0, 0, 0);
                }
            }
        }
    };

    // ---- MFMAs of one landed slot.  k index of lane = pixel 8 kq + r (+4 for the second transpose read): tile row kq, column r.
    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int kq = lane >> 4, r = (lane & 15) >> 2, csub = (lane & 3) * 8, par = kq & 1;
    const int dbase = (8 * kq + r) * 64 + csub;
    const int xb0 = WD_DBYTES + (kq * WD_PW + r) * 64 + csub + par * 32;          // logical half 0 on even tap rows
    const int xb1 = WD_DBYTES + (kq * WD_PW + r) * 64 + csub + (par ^ 1) * 32;    // logical half 1 on even tap rows (and half 0 on odd ones)
    auto compute = [&](int slot) {
        const char* buf = wbase + slot * WD_STAGE;
        bf16x8 af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* a = buf + dbase + (i ? (par ^ 1) : par) * 32;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)a);
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a + 256));
            af[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int b = 0; b < 9; ++b) {
            const int ta = b / 3, tb = b - 3 * ta;
            const int toff = (ta * WD_PW + tb) * 64;
            bf16x8 bfr[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char* a = buf + (((j ^ (ta & 1)) != 0) ? xb1 : xb0) + toff;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)a);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a + 256));
                bfr[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[b][i][j] = mfma16<T>(af[i], bfr[j], acc[b][i][j]);
        }
    };

    if constexpr (XF) __syncthreads();             // coefficient table visible
    int tile = bsplit * NW + wave;
    TC t0;
    { t0.txi = tile % p.tilesX; const int q = tile / p.tilesX; t0.tyi = q % p.tilesY; t0.n = q / p.tilesY; }
    TC t1 = tc_next(t0), t2 = tc_next(t1);
    // ring prologue: two slots in flight
    if (tile < p.ntiles) issue(t0, 0);
    if (tile + stride < p.ntiles) issue(t1, 1);
    int slot = 0;
    for (; tile < p.ntiles; tile += stride) {
        const bool has1 = tile + stride < p.ntiles, has2 = tile + 2 * stride < p.ntiles;
        int s2 = slot + 2; if (s2 >= WD_NST) s2 -= WD_NST;
        // (the slot refilled here was read by the MFMAs of the previous iteration; their operands are in registers by now)
        if (has2) { issue(t2, s2); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        else if (has1) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        fixup(t0, slot);
        compute(slot);
        t0 = t1; t1 = t2; t2 = tc_next(t2);
        if (++slot == WD_NST) slot = 0;
    }
    // ---- tree-reduce the waves' accumulators through LDS, then one wave stores the slice ----
    constexpr int NTW = 36;
    bool flusher = true;
    for (int half = NW / 2; half >= 1; half >>= 1) {
        __syncthreads();
        const bool dump = flusher && wave >= half && wave < 2 * half;
        const bool take = flusher && wave < half;
        char* region = smem + (size_t)(wave % half) * (NTW * 1024);
        if (dump) {
            int t = 0;
#pragma unroll
            for (int b = 0; b < 9; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j, ++t) *(f32x4*)(region + (t * 64 + lane) * 16) = acc[b][i][j];
            flusher = false;
        }
        __syncthreads();
        if (take) {
            int t = 0;
#pragma unroll
            for (int b = 0; b < 9; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j, ++t) acc[b][i][j] += *(const f32x4*)(region + (t * 64 + lane) * 16);
        }
    }
    if (flusher) {
#pragma unroll
        for (int b = 0; b < 9; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int co = co0 + i * 16 + (lane >> 4) * 4, ci = ci0 + j * 16 + (lane & 15);
                    if (co < p.Co16 && ci < p.Ci16) {
                        float* o = p.dwp + (size_t)bsplit * p.slice + ((size_t)b * p.Co16 + co) * p.Ci16 + ci;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) o[(size_t)rr * p.Ci16] = acc[b][i][j][rr];
                    }
                }
    }
}

bool wgrad_dma_eligible(const mfc_wgrad_desc* d) {
    if (!g_wgrad_dma || !mfc_is16(d->dtype) || d->batch > 1) return false;
    if (d->TA != 3 || d->TB != 3 || d->in_stride != 1 || d->dh0 != -1 || d->dw0 != -1) return false;
    if (d->Hin != d->Hout || d->Win != d->Wout) return false;
    if (d->Cin % 32 || d->Cout % 32 || d->Cin_p != d->Cin || d->Cout_p != d->Cout) return false;      // whole 32-channel blocks
    if (d->N / d->images_per_group > 8) return false;
    // 32-bit lane offsets inside one image
    if ((double)d->Hin * d->Win * (d->Cin_p > d->Cout_p ? d->Cin_p : d->Cout_p) * 2.0 >= 2.0e9) return false;
    return true;
}

static void wgrad_dma_setup(const mfc_wgrad_desc* d, WgradD& f) {
    f.x = (const char*)d->x; f.dy = (const char*)d->dy; f.dwp = d->dwp; f.in_coef = d->in_coef;
    f.N = d->N; f.H = d->Hin; f.W = d->Win; f.Cin_p = d->Cin_p; f.Cout_p = d->Cout_p;
    f.in_relu = d->in_relu; f.ipg = d->images_per_group; f.G = d->N / d->images_per_group;
    f.tilesY = ceil_div(f.H, 4); f.tilesX = ceil_div(f.W, 8);
    f.ntiles = f.N * f.tilesY * f.tilesX;
    f.Co16 = d->Cout; f.Ci16 = d->Cin;
    f.co_blocks = d->Cout / 32; f.ci_blocks = d->Cin / 32;
    const int Y = f.co_blocks * f.ci_blocks;
    int S = d->splits;
    if (S <= 0) S = ceil_div(g_wgrad_blocks, Y);
    if (S * 4 > f.ntiles) S = ceil_div(f.ntiles, 4);
    if (S < 1) S = 1;
    f.splits = S;
    f.slice = 9 * f.Co16 * f.Ci16;
}

int wgrad_dma_parts(const mfc_wgrad_desc* d) { WgradD f; wgrad_dma_setup(d, f); return f.splits; }

int wgrad_dma_launch(const mfc_wgrad_desc* d, hipStream_t st) {
    WgradD f; wgrad_dma_setup(d, f);
    const bool xf = d->in_coef != nullptr;
    const int nw = (xf && g_wgrad_dma_xf8) ? 8 : 4;
    const size_t lds = (size_t)nw * WD_WAVE + (xf ? (size_t)f.G * 64 * 4 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<bf16_t, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<bf16_t, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<bf16_t, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<f16_t, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<f16_t, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<f16_t, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        const double flops = 2.0 * f.N * f.H * f.W * (double)f.Co16 * f.Ci16 * 9.0;
        const double bytes = ((double)f.N * f.H * f.W * (f.Cin_p + f.Cout_p)) * 2.0;
        const bool h = d->dtype == MFC_F16;
        // (the profiler keeps the pointer: literals only)
        const char* pname = h ? (xf ? (nw == 8 ? "conv_wgrad_dma_kernel<_Float16, true, 8>" : "conv_wgrad_dma_kernel<_Float16, true, 4>") : "conv_wgrad_dma_kernel<_Float16, false, 4>")
                              : (xf ? (nw == 8 ? "conv_wgrad_dma_kernel<__bf16, true, 8>" : "conv_wgrad_dma_kernel<__bf16, true, 4>") : "conv_wgrad_dma_kernel<__bf16, false, 4>");
        mfc_prof_before(st, pname, flops, bytes);
    }
    const int grid = f.splits * f.co_blocks * f.ci_blocks;
    if (xf && nw == 8) MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL((conv_wgrad_dma_kernel<T_, true, 8>), dim3(grid), dim3(512), lds, st, f));
    else if (xf) MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL((conv_wgrad_dma_kernel<T_, true, 4>), dim3(grid), dim3(256), lds, st, f));
    else MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL((conv_wgrad_dma_kernel<T_, false, 4>), dim3(grid), dim3(256), lds, st, f));
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}
