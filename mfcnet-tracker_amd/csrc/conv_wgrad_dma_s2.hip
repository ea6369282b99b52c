// Weight gradient of the 3x3 / STRIDE-2 / pad-1 convolutions (16-bit storage), LDS-DMA fed: the down-sampling paths of the fuse layers
// and the transitions (models/hrnet.py:200-230, 361-386, under loss.backward()).
//   dWp[a,b][co][ci] = sum_{n,i,j} dy[n,i,j,co] * f(x[n, 2i-1+a, 2j-1+b, ci])
// Until round 3 these launches ran on the register-staged wave kernel (conv_wgrad.hip) at 56-157 TFLOP/s: 35-115 us for 37-44 MB of
// operands, 4-6x their HBM time -- one sub-tile in flight per wave.  This is conv_wgrad_dma.hip's structure (workgroup = 32 x 32 channel
// block for all nine taps, each of 4 waves walks its own 4 x 8-pixel OUTPUT sub-tiles = one MFMA k-step, 36 accumulator tiles, a private
// ring of three staging slots filled by global_load_lds_dwordx4 two sub-tiles ahead, counted s_waitcnt, no workgroup barrier in the loop)
// with the stride-2 patch: 9 x 17 input pixels per sub-tile (10 DMA pieces) instead of 6 x 10.
//   * LDS layout of the patch (tools/probe/s2_banks.py, brute force over all taps): a lane of a ds_read_b64_tr_b16 reads patch pixel
//     (2 kq + a, 2 r + b), i.e. every OTHER pixel of every OTHER row; with 64-byte pixels that puts the 4 pixels of a 16-lane group 128 B
//     apart (two of them on the same banks) and the two row groups of a 32-lane pass a whole number of 256-byte bank rows apart.
//     Conflict-free form: rows 1120 B apart (17.5 pixels: the half pixel moves the second row group onto the other half of the banks)
//     and the two 32-byte channel halves of a pixel swapped where (px >> 2) is odd -- both applied on the SOURCE side of the DMA (each lane
//     fetches the granule that belongs in its slot), free;
//   * channel counts need only be multiples of 16 (HRNet-W48: 48 -> 96, 48 -> 48, ...): a 32-channel block that sticks out of the tensor
//     fetches granule 0 of the same pixel for the missing channels -- real, finite data whose products land in cells of the 32 x 32
//     block that are never stored;
//   * out-of-image pixels come from clamped addresses and are zeroed in LDS after landing (edge tiles only); the producer's BatchNorm +
//     ReLU is applied in place in LDS by the wave that will read the slot (lanes 0-31 take the pixels with unswapped halves, 32-63 the
//     swapped ones, so that a lane's 8 channels -- and their coefficients -- are the same for all its pieces).
#include "common.h"

int g_wgrad_dma_s2 = 1;              // mfc_set_flag(46, v): 0 = register-staged wave kernel (conv_wgrad.hip) for these launches
extern int g_wgrad_blocks;           // target workgroups per launch (conv_wgrad.hip, mfc_set_flag(11))

#define S2_PH 9
#define S2_PW 17
#define S2_PITCH 1120                // bytes between patch rows
#define S2_PSLOTS 70                 // 16-byte slots per patch row (68 used)
#define S2_DBYTES 2048
#define S2_XP 10                     // DMA pieces of the patch (9 x 1120 = 10 080 B of 10 240)
#define S2_STAGE (S2_DBYTES + S2_XP * 1024)
#define S2_NST 3
#define S2_WAVE (S2_NST * S2_STAGE)

struct WgradS2 {
    const char* x; const char* dy; float* dwp; const float* in_coef;
    int N, H, W, Ho, Wo, Cin_p, Cout_p;
    int in_relu, ipg, G;
    int tilesY, tilesX, ntiles, splits;
    int Co16, Ci16, co_blocks, ci_blocks;
    int slice;                       // floats per partial-sum slice of dwp
};

__device__ inline void s2_dma(const char* base, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds) : "memory");
}

template <typename TE, bool XF>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_s2_kernel(WgradS2 p) {
    typedef TE T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* wbase = smem + wave * S2_WAVE;
    float* coefs = (float*)(smem + 4 * S2_WAVE);          // [G][2][32] scale / shift of this block's input channels
    const int Ytot = p.co_blocks * p.ci_blocks;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int y = Lb % Ytot, bsplit = Lb / Ytot;
    const int ib = y % p.ci_blocks, cb = y / p.ci_blocks;
    const int co0 = cb * 32, ci0 = ib * 32;
    if constexpr (XF) {
        for (int i = tid; i < p.G * 64; i += 256) {
            const int ch = i & 31, w = (i >> 5) & 1, g = i >> 6;
            coefs[i] = (ci0 + ch < p.Cin_p) ? p.in_coef[((size_t)g * 4 + w) * p.Cin_p + ci0 + ch] : 0.f;
        }
    }
    const int drow = p.Cout_p * 2, xrow = p.Cin_p * 2;
    // granules of the 32-channel block that exist in the tensor (a block may stick out by 16 channels: module comment)
    const int dgn = min(4, (p.Cout_p - co0) >> 3), xgn = min(4, (p.Cin_p - ci0) >> 3);
    // ---- per-lane DMA tables: LDS slot 64 i + lane of a piece sequence -> (pixel, source granule)
    int d_vof[2]; unsigned d_pk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = 64 * i + lane, pp = s >> 2, gp = s & 3;
        const int ty = pp >> 3, tx = pp & 7;
        int gl = gp ^ ((ty & 1) << 1);
        if (gl >= dgn) gl = 0;
        d_pk[i] = ty | (tx << 4) | (gl << 8);
        d_vof[i] = (ty * p.Wo + tx) * drow + gl * 16;
    }
    auto x_slot = [&](int i, int ln, int& py, int& px, int& gl) {
        const int s = 64 * i + ln;
        py = min(s / S2_PSLOTS, S2_PH - 1);
        const int rem = s - (s / S2_PSLOTS) * S2_PSLOTS;
        px = min(rem >> 2, S2_PW - 1);
        const int gp = rem & 3;
        gl = gp ^ (((px >> 2) & 1) << 1);
        if (gl >= xgn) gl = 0;
    };
    int x_vof[S2_XP];
#pragma unroll
    for (int i = 0; i < S2_XP; ++i) {
        int py, px, gl;
        x_slot(i, lane, py, px, gl);
        x_vof[i] = (py * p.W + px) * xrow + gl * 16;
    }

    f32x4 acc[9][2][2];
#pragma unroll
    for (int b = 0; b < 9; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // tile coordinates (output sub-tiles of 4 x 8 pixels), tracked incrementally
    const int stride = p.splits * 4;
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
        return i0 >= 1 && j0 >= 1 && i0 + 4 <= p.Ho && j0 + 8 <= p.Wo && 2 * i0 + 7 <= p.H - 1 && 2 * j0 + 15 <= p.W - 1;
    };
    auto issue = [&](const TC& c, int slot) {
        const int i0 = c.tyi * 4, j0 = c.txi * 8;
        const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(wbase + slot * S2_STAGE);
        const char* dimg = p.dy + ((size_t)c.n * p.Ho * p.Wo * p.Cout_p + co0) * sizeof(T);
        const char* ximg = p.x + ((size_t)c.n * p.H * p.W * p.Cin_p + ci0) * sizeof(T);
        if (is_interior(c)) {           // wave-uniform
            const char* dt = dimg + (size_t)(i0 * p.Wo + j0) * drow;
            const char* xt = ximg + (size_t)((2 * i0 - 1) * p.W + (2 * j0 - 1)) * xrow;
#pragma unroll
            for (int i = 0; i < 2; ++i) s2_dma(dt, (unsigned)d_vof[i], lds + 1024 * i);
#pragma unroll
            for (int i = 0; i < S2_XP; ++i) s2_dma(xt, (unsigned)x_vof[i], lds + S2_DBYTES + 1024 * i);
        } else {                        // clamped (always valid) addresses; the out-of-image slots are zeroed after landing
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = min(i0 + (int)(d_pk[i] & 15), p.Ho - 1), ox = min(j0 + (int)((d_pk[i] >> 4) & 15), p.Wo - 1);
                s2_dma(dimg, (unsigned)((oy * p.Wo + ox) * drow + (int)(d_pk[i] >> 8) * 16), lds + 1024 * i);
            }
            int ln = lane;
            asm volatile("" : "+v"(ln));          // (keeps the recomputed slot coordinates inside this branch: hoisted they would be spilled)
#pragma unroll
            for (int i = 0; i < S2_XP; ++i) {
                int py, px, gl;
                x_slot(i, ln, py, px, gl);
                const int iy = min(max(2 * i0 - 1 + py, 0), p.H - 1), ix = min(max(2 * j0 - 1 + px, 0), p.W - 1);
                s2_dma(ximg, (unsigned)((iy * p.W + ix) * xrow + gl * 16), lds + S2_DBYTES + 1024 * i);
            }
        }
    };

    // ---- in-LDS fix-up of a landed slot (the reading wave does it): zero what lies outside the image; apply the producer's BN + ReLU.
    //      Lanes 0-31 take the pixels whose halves are in place ((px >> 2) even: 9 per row), lanes 32-63 the swapped ones (8 per row)
    const int xh = lane >> 5, xli = lane & 31;
    const int xg = xli & 3, xgl = xg ^ (xh << 1);
    const float relu_floor = p.in_relu ? 0.f : -3.0e38f;
    auto fixup = [&](const TC& c, int slot) {
        const bool interior = is_interior(c);
        if (!XF && interior) return;
        const int i0 = c.tyi * 4, j0 = c.txi * 8;
        char* buf = wbase + slot * S2_STAGE;
        if (!interior) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = 64 * i + lane, pp = s >> 2;
                if (i0 + (pp >> 3) >= p.Ho || j0 + (pp & 7) >= p.Wo) *(uint4*)(buf + s * 16) = make_uint4(0, 0, 0, 0);
            }
        }
        float sc[8], sh[8];
        if constexpr (XF) {
            const float* cf = coefs + (c.n / p.ipg) * 64 + xgl * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = cf[e]; sh[e] = cf[32 + e]; }
        }
        const int per_row = xh ? 8 : 9, count = per_row * S2_PH;
        for (int it = 0; it < 11; ++it) {
            const int cidx = (xli >> 2) + 8 * it;
            if (cidx < count) {
                const int py = cidx / per_row, k = cidx - py * per_row;
                const int px = xh ? (k < 4 ? k + 4 : k + 8) : (k < 4 ? k : (k < 8 ? k + 4 : 16));
                char* a = buf + S2_DBYTES + py * S2_PITCH + px * 64 + xg * 16;
                const int iy = 2 * i0 - 1 + py, ix = 2 * j0 - 1 + px;
                const bool inr = interior || ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W);
                if constexpr (XF) {
                    uint4 v = *(const uint4*)a;
                    float f[8];
                    Gran<T>::unpack(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] * sc[e] + sh[e], relu_floor);      // (v_max swallows NaN: see conv_wgrad.hip store_tile)
                    v = Gran<T>::pack(f);
                    *(uint4*)a = inr ? v : make_uint4(0, 0, 0, 0);
                } else {
                    if (!inr) *(uint4*)a = make_uint4(0, 0, 0, 0);
                }
            }
        }
    };

    // ---- MFMAs of one landed slot.  k index of lane = output pixel 8 kq + r (+4 for the second transpose read): tile row kq, column r
    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int kq = lane >> 4, r = (lane & 15) >> 2, csub = (lane & 3) * 8, par = kq & 1;
    const int dbase = (8 * kq + r) * 64 + csub;
    const int xbase = S2_DBYTES + (2 * kq) * S2_PITCH + (2 * r) * 64 + csub;
    int hsel[3];                                            // byte offset of LOGICAL half 0 of patch column 2 r + b (its other half: ^ 32)
#pragma unroll
    for (int b = 0; b < 3; ++b) hsel[b] = (((2 * r + b) >> 2) & 1) * 32;
    auto compute = [&](int slot) {
        const char* buf = wbase + slot * S2_STAGE;
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
            const int toff = ta * S2_PITCH + tb * 64;
            bf16x8 bfr[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char* a = buf + xbase + toff + (j ? (hsel[tb] ^ 32) : hsel[tb]);
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)a);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a + 512));      // columns 2 (r + 4) + b: same half swap ((px + 8) >> 2 has the parity of px >> 2)
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
    int tile = bsplit * 4 + wave;
    TC t0;
    { t0.txi = tile % p.tilesX; const int q = tile / p.tilesX; t0.tyi = q % p.tilesY; t0.n = q / p.tilesY; }
    TC t1 = tc_next(t0), t2 = tc_next(t1);
    // ring prologue: two slots in flight (12 DMA pieces each)
    if (tile < p.ntiles) issue(t0, 0);
    if (tile + stride < p.ntiles) issue(t1, 1);
    int slot = 0;
    for (; tile < p.ntiles; tile += stride) {
        const bool has1 = tile + stride < p.ntiles, has2 = tile + 2 * stride < p.ntiles;
        int s2 = slot + 2; if (s2 >= S2_NST) s2 -= S2_NST;
        // (the slot refilled here was read by the MFMAs of the previous iteration; their operands are in registers by now)
        if (has2) { issue(t2, s2); asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
        else if (has1) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        fixup(t0, slot);
        compute(slot);
        t0 = t1; t1 = t2; t2 = tc_next(t2);
        if (++slot == S2_NST) slot = 0;
    }
    // ---- tree-reduce the four waves' accumulators through LDS, then one wave stores the slice ----
    constexpr int NTW = 36;
    bool flusher = true;
    for (int half = 2; half >= 1; half >>= 1) {
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

bool wgrad_dma_s2_eligible(const mfc_wgrad_desc* d) {
    if (!g_wgrad_dma_s2 || !mfc_is16(d->dtype) || d->batch > 1) return false;
    if (d->TA != 3 || d->TB != 3 || d->in_stride != 2 || d->dh0 != -1 || d->dw0 != -1) return false;
    if (d->Hout != (d->Hin - 1) / 2 + 1 || d->Wout != (d->Win - 1) / 2 + 1) return false;
    if (d->Cin % 16 || d->Cout % 16 || d->Cin_p != d->Cin || d->Cout_p != d->Cout) return false;      // whole 16-channel half blocks
    if (d->N / d->images_per_group > 8) return false;
    if (d->Hin < 2 || d->Win < 2) return false;
    // Where both channel counts are multiples of 48 AND there are many 32 x 32 blocks (HRNet-W48: 96 -> 192, 192 -> 384, 48 -> 192 ...), the wave
    // kernel's 48 x 48 register blocks fit the problem exactly and re-read the 9 x 17 patch less often: measured 65.7 vs 74.1 us (96 -> 192 at
    // 60x80), 51.3 vs 82.6 (192 -> 384), 57.7 vs 75.4 (48 -> 192); with few blocks the ring wins (48 -> 96: 113.5 -> 62.7, 48 -> 48: 68.6 -> 27.0)
    if (d->Cin % 48 == 0 && d->Cout % 48 == 0 && ceil_div(d->Cin, 32) * ceil_div(d->Cout, 32) > 6) return false;
    // 32-bit lane offsets inside one image
    if ((double)d->Hin * d->Win * (d->Cin_p > d->Cout_p ? d->Cin_p : d->Cout_p) * 2.0 >= 2.0e9) return false;
    return true;
}

static void wgrad_dma_s2_setup(const mfc_wgrad_desc* d, WgradS2& f) {
    f.x = (const char*)d->x; f.dy = (const char*)d->dy; f.dwp = d->dwp; f.in_coef = d->in_coef;
    f.N = d->N; f.H = d->Hin; f.W = d->Win; f.Ho = d->Hout; f.Wo = d->Wout; f.Cin_p = d->Cin_p; f.Cout_p = d->Cout_p;
    f.in_relu = d->in_relu; f.ipg = d->images_per_group; f.G = d->N / d->images_per_group;
    f.tilesY = ceil_div(f.Ho, 4); f.tilesX = ceil_div(f.Wo, 8);
    f.ntiles = f.N * f.tilesY * f.tilesX;
    f.Co16 = d->Cout; f.Ci16 = d->Cin;
    f.co_blocks = ceil_div(d->Cout, 32); f.ci_blocks = ceil_div(d->Cin, 32);
    const int Y = f.co_blocks * f.ci_blocks;
    int S = d->splits;
    if (S <= 0) S = ceil_div(g_wgrad_blocks, Y);
    if (S * 4 > f.ntiles) S = ceil_div(f.ntiles, 4);
    if (S < 1) S = 1;
    f.splits = S;
    f.slice = 9 * f.Co16 * f.Ci16;
}

int wgrad_dma_s2_parts(const mfc_wgrad_desc* d) { WgradS2 f; wgrad_dma_s2_setup(d, f); return f.splits; }

int wgrad_dma_s2_launch(const mfc_wgrad_desc* d, hipStream_t st) {
    WgradS2 f; wgrad_dma_s2_setup(d, f);
    const bool xf = d->in_coef != nullptr;
    const size_t lds = (size_t)4 * S2_WAVE + (xf ? (size_t)f.G * 64 * 4 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_s2_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_s2_kernel<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_s2_kernel<f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_s2_kernel<f16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        const double flops = 2.0 * f.N * f.Ho * f.Wo * (double)f.Co16 * f.Ci16 * 9.0;
        const double bytes = ((double)f.N * f.H * f.W * f.Cin_p + (double)f.N * f.Ho * f.Wo * f.Cout_p) * 2.0;
        const bool h = d->dtype == MFC_F16;
        mfc_prof_before(st, h ? (xf ? "conv_wgrad_dma_s2_kernel<_Float16, true>" : "conv_wgrad_dma_s2_kernel<_Float16, false>")
                              : (xf ? "conv_wgrad_dma_s2_kernel<__bf16, true>" : "conv_wgrad_dma_s2_kernel<__bf16, false>"), flops, bytes);
    }
    const int grid = f.splits * f.co_blocks * f.ci_blocks;
    if (xf) MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL((conv_wgrad_dma_s2_kernel<T_, true>), dim3(grid), dim3(256), lds, st, f));
    else MFC_TYPED16(d->dtype, T_, hipLaunchKernelGGL((conv_wgrad_dma_s2_kernel<T_, false>), dim3(grid), dim3(256), lds, st, f));
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}
