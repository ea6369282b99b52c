// 3x3 / stride-1 / pad-1 convolutions with 48 channels on both sides (16-bit storage), forward and data gradient: the BasicBlocks of
// HRNet-W48's 120x160 branch (models/hrnet.py:58-74, 297-333; 128 launches of a W48 step on conv_igemm's two-stage <3,4,6,4> geometry,
// 39-52 us each for 88 MB of tensors).  conv3x3_ring.hip's design -- weights resident for the whole launch, the halo patch of a pixel tile
// DMA-copied (global_load_lds_dwordx4) into a ring of three LDS slots two tiles ahead, counted s_waitcnt + ONE raw s_barrier per tile,
// zero padding and the producer's BatchNorm + ReLU applied in place in LDS by the wave that issued the piece -- for a channel count
// that is one and a half MFMA k-steps:
//   * per tap, channels 0-31 go through v_mfma_f32_16x16x32 (A = weights in REGISTERS: 9 taps x 3 cout tiles = 108 VGPRs) and channels
//     32-47 through v_mfma_f32_16x16x16 (A = 13.5 KiB of weights resident in LDS, one ds_read_b64 per tap and cout tile);
//   * a patch slot is two planes: 64 B per pixel (channels 0-31, the half-swap swizzle of conv3x3_ring.hip) and 48 B per pixel (channels
//     32-47 + 16 B of padding: with 32-byte pixels the 16 pixels of a ds_read_b64 pass would wrap onto the same banks; 48 B is conflict
//     free without any swizzle).  A DMA writes lane-contiguous LDS, so the padding slot of a pixel just fetches a valid granule;
//   * workgroup = 4 waves = 4 pixel groups of MT = 4 rows x 16 pixels x all 48 couts (16 x 16-pixel tiles, one workgroup per CU: 111 KiB of
//     ring); epilogue: cout tiles 0 and 1 transposed across the 16-lane rows (16-byte stores of 8 channels), tile 2 stored as 8-byte
//     quads; per-channel sum / sum of squares in registers across tiles; `accumulate` (data gradient into an existing tensor) with the
//     old values requested in front of the MFMA block.
// Dispatched from mfc_conv2d_fwd / mfc_conv2d_layout when ring48_eligible() (mfc_set_flag(50, 1)).
//
// STATUS (round 3): correct (48 parity cases, tests/test_gpu_ring.py) but NOT faster yet, so OFF by default: 48 -> 48 at 120x160, N = 24:
// 44-46 us (MT = 2, two workgroups per CU, fragments of tap t + 1 requested under the MFMAs of tap t) against conv_igemm's 40.5 us; the first
// version (one workgroup per CU, two waits for LDS fragments per tap) took 61 us.  Ablation (tools/dbg48.py, mfc_set_flag(32, mask): 1 / 2 the
// two MFMA shapes, 4 DMA, 8 stores, 16 fragment reads, 32 epilogue, 64 fix-up): launch + prologue + barriers 8.9 us, fragment reads 8.9,
// border fix-up 7.2 (30 % of the 8 x 16 tiles are edge tiles and every piece recomputes its pixel coordinates), epilogue 3.1, DMA + stores
// ~10, MFMAs ~6.  The fix-up and the read block are where the remaining 10-15 us are.
// One finding worth keeping whatever happens to this kernel: a v_mfma_f32_16x16x16 issued directly behind the v_mfma_f32_16x16x32 that
// writes its SrcC returned stale rows 0-1 of the accumulator (hipcc 7.2 / gfx950 inserts no wait states between the two shapes); the two
// phases below keep such pairs 6-12 MFMAs apart.
#include "common.h"

int g_conv_ring48 = 0;                // OFF: correct but slower than conv_igemm (STATUS below); mfc_set_flag(50, 1) turns it on
extern int g_ring_ablate;

struct Ring48K {
    const char* in; const char* wp; char* out;
    const float* in_coef; mfc_stat_t* out_stats;
    const mfc_bnfin_desc* in_fin;
    int N, H, W;
    int in_relu, ipg, G, accumulate;
    int tilesY, tilesX, ntiles, per_block;
    int off_w16, off_coef, off_red;          // LDS byte offsets behind the ring
    int ablate;                              // tuning / debugging only (mfc_set_flag(32, mask)): 1 skip the k = 16 MFMAs, 2 skip the k = 32 MFMAs
};

template <int CTRL> __device__ inline float q_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ inline float q_row16_sum(float v) {
    v = q_dpp_add<0xB1>(v); v = q_dpp_add<0x4E>(v); v = q_dpp_add<0x141>(v); v = q_dpp_add<0x140>(v);
    return v;
}
__device__ inline void q_dma(const char* base, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds) : "memory");
}
__device__ inline void q_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((ext_vector_type(4))) unsigned q_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned q_u32x2;
typedef __attribute__((ext_vector_type(4))) short q_s16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 q_h4;
template <typename T> __device__ inline f32x4 q_mfma_k16(q_s16x4 a, q_s16x4 b, f32x4 c);
template <> __device__ inline f32x4 q_mfma_k16<bf16_t>(q_s16x4 a, q_s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
template <> __device__ inline f32x4 q_mfma_k16<f16_t>(q_s16x4 a, q_s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(q_h4, a), __builtin_bit_cast(q_h4, b), c, 0, 0, 0);
}
template <typename T> __device__ inline unsigned q_pack2(float a, float b);
template <> __device__ inline unsigned q_pack2<bf16_t>(float a, float b) { return pack_bf16x2(a, b); }
template <> __device__ inline unsigned q_pack2<f16_t>(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}

constexpr int Q_PW = 18;                    // patch width: 16-pixel tile rows + halo
constexpr int Q_NSLOT = 3;
constexpr int Q_C = 48;
template <int MT> struct Ring48Geo {
    static constexpr int TH = 4 * MT, PH = TH + 2, NPX = PH * Q_PW;
    static constexpr int PA = ((NPX * 64 + 1023) / 1024) * 1024, PPA = PA / 1024;      // plane of channels 0-31
    static constexpr int PB = ((NPX * 48 + 1023) / 1024) * 1024, PPB = PB / 1024;      // plane of channels 32-47 (48-byte pixels)
    static constexpr int SLOT = PA + PB, NP = PPA + PPB, NPW = (NP + 3) / 4, RING = Q_NSLOT * SLOT;
};

template <typename T, int MT, bool ACC>
__global__ __launch_bounds__(256, 2) void conv3x3_ring48_kernel(Ring48K p) {
    typedef Ring48Geo<MT> Geo;
    constexpr int C = Q_C;
    constexpr int TH = Geo::TH, NPX = Geo::NPX, PA = Geo::PA, PPA = Geo::PPA, NP = Geo::NP, NPW = Geo::NPW, SLOT = Geo::SLOT;
    constexpr int ROWA = Q_PW * 64, ROWB = Q_PW * 48;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w16 = smem + p.off_w16;                                // [9 taps][3 cout tiles][64 lanes][8 B]: A fragments of the 16-channel remainder
    float* coefl = (float*)(smem + p.off_coef);                  // [G][2][48] scale / shift of the fused input transform
    float* red = (float*)(smem + p.off_red);                     // [2 parities][4 waves][2][48]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave;
    const int gl = lane >> 4, lx = lane & 15;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int u0 = Lb * p.per_block;
    const int nun = min(p.per_block, p.ntiles - u0);
    if (nun <= 0) return;
    if (p.in_fin) bn_fold_prologue(p.in_fin, Lb == 0);
    const int H = p.H, W = p.W;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    struct TC { int n, tyi, txi; };
    auto tc_next = [&](TC c) {
        if (++c.txi == p.tilesX) { c.txi = 0; if (++c.tyi == p.tilesY) { c.tyi = 0; ++c.n; } }
        return c;
    };
    auto tc_interior = [&](const TC& c) {
        const int i0 = c.tyi * TH, j0 = c.txi * 16;
        return i0 >= 1 && j0 >= 1 && i0 + TH + 1 <= H && j0 + 17 <= W;
    };
    TC tc0;
    { const int tpi = p.tilesY * p.tilesX; tc0.n = u0 / tpi; const int r = u0 - tc0.n * tpi; tc0.tyi = r / p.tilesX; tc0.txi = r - tc0.tyi * p.tilesX; }
    TC tc1 = tc_next(tc0), tc2 = tc_next(tc1);

    // ---------------- DMA tables: piece q = wave + 4 i of a slot; its LDS slot 64 q' + lane -> (patch pixel, source granule) ----------------
    const int npw = (NP - wave + 3) / 4;                         // pieces this wave issues per slot (wave-uniform)
    auto piece_src = [&](int i, int ln, int& py, int& px, int& g) {
        const int q = wave + 4 * i;
        bool real;
        int pix;
        if (q < PPA) {
            const int S = q * 64 + ln;
            pix = S >> 2; real = pix < NPX; pix = min(pix, NPX - 1);
            py = pix / Q_PW; px = pix - py * Q_PW;
            g = (S & 3) ^ (((px >> 2) & 1) << 1);
        } else {
            const int S = (q - PPA) * 64 + ln;
            pix = S / 3;
            const int sub = S - 3 * pix;
            real = pix < NPX && sub < 2; pix = min(pix, NPX - 1);
            py = pix / Q_PW; px = pix - py * Q_PW;
            g = 4 + min(sub, 1);
        }
        return real;
    };
    // (the per-lane source offsets are recomputed for every tile from an opaque copy of the lane id: kept in registers across the MFMA block
    //  they spill the resident weights)
    auto issue = [&](const TC& c, int slot) {
        const int i0 = c.tyi * TH, j0 = c.txi * 16;
        const char* img = p.in + (size_t)c.n * H * W * C * 2;
        const unsigned lbase = lds0 + slot * SLOT + wave * 1024;
        if (p.ablate & 4) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (tc_interior(c)) {           // workgroup-uniform
            const char* pb = img + ((size_t)(i0 - 1) * W + (j0 - 1)) * C * 2;
#pragma unroll
            for (int i = 0; i < NPW; ++i)
                if (i < npw) {
                    int py, px, g;
                    piece_src(i, ln, py, px, g);
                    q_dma(pb, (unsigned)(((py * W + px) * C + g * 8) * 2), lbase + i * 4096);
                }
        } else {                        // clamped (always valid) addresses; the out-of-image pixels are zeroed after landing
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                if (i < npw) {
                    int py, px, g;
                    piece_src(i, ln, py, px, g);
                    const int iy = min(max(i0 - 1 + py, 0), H - 1), ix = min(max(j0 - 1 + px, 0), W - 1);
                    q_dma(img, (unsigned)(((iy * W + ix) * C + g * 8) * 2), lbase + i * 4096);
                }
            }
        }
    };
    // all but this wave's `k` youngest vector-memory operations are done
    auto vm_wait = [&](int k) {
#define Q_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
        switch (k) { Q_W(0) Q_W(1) Q_W(2) Q_W(3) Q_W(4) Q_W(5) Q_W(6) Q_W(7) Q_W(8) Q_W(9) default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break; }
#undef Q_W
    };
    static_assert(NPW <= 10, "vm_wait covers up to ten pieces per wave");

    // ---------------- resident weights ----------------
    // packed image [tap][granule 0..5][cout 48][16 B] (mfc_conv2d_layout: KG = 6, nchunks = Yblocks = 1, NT16 = 48, nslots = 54)
    bf16x8 wr[9][3];
    {
        const char* ws = p.wp + ((size_t)gl * C + lx) * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) wr[t][nt] = *(const bf16x8*)(ws + ((size_t)(t * 6) * C + nt * 16) * 16);
    }
    for (int i = tid; i < 9 * 3 * 64; i += 256) {
        const int l = i & 63, tn = i >> 6, t = tn / 3, nt = tn - 3 * t;
        const int g4 = l >> 4, c16 = l & 15;
        *(uint2*)(w16 + i * 8) = *(const uint2*)(p.wp + ((size_t)(t * 6 + 4 + (g4 >> 1)) * C + nt * 16 + c16) * 16 + (g4 & 1) * 8);
    }
    const bool xf = (p.in_coef != nullptr);
    if (xf) {
        for (int i = tid; i < p.G * 2 * C; i += 256) {
            const int g = i / (2 * C), r = i - g * 2 * C;
            coefl[i] = p.in_coef[(size_t)g * 4 * C + r];          // rows 0 (scale) and 1 (shift) of [G][4][C]
        }
    }

    // ---------------- in-LDS fix-up of the pieces this wave issued into `slot` ----------------
    auto fixup = [&](const TC& c, int slot) {
        const bool interior = tc_interior(c);
        if (!xf && interior) return;
        const int i0 = c.tyi * TH, j0 = c.txi * 16;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const float* cfg = coefl + (c.n / p.ipg) * 2 * C;
        char* lb = smem + slot * SLOT + wave * 1024 + ln * 16;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            if (i < npw) {
                int py, px, g;
                const bool real = piece_src(i, ln, py, px, g);
                const bool inr = interior || ((unsigned)(i0 - 1 + py) < (unsigned)H && (unsigned)(j0 - 1 + px) < (unsigned)W);
                char* a = lb + i * 4096;
                if (real) {
                    if (xf) {
                        uint4 v = *(const uint4*)a;
                        const float4 s0 = *(const float4*)(cfg + g * 8), s1 = *(const float4*)(cfg + g * 8 + 4);
                        const float4 h0 = *(const float4*)(cfg + C + g * 8), h1 = *(const float4*)(cfg + C + g * 8 + 4);
                        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                        float f[8];
                        Gran<T>::unpack(v, f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float t = f[e] * sc[e] + sh[e];
                            f[e] = p.in_relu ? relu_nan(t) : t;
                        }
                        v = Gran<T>::pack(f);
                        *(uint4*)a = inr ? v : make_uint4(0, 0, 0, 0);
                    } else {
                        if (!inr) *(uint4*)a = make_uint4(0, 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---------------- accumulators / statistics ----------------
    f32x4 acc[MT][3];
    float ssum[3][4], ssq[3][4];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }
    }
    int red_par = 0; bool red_pending = false; int red_grp = 0, red_rep = 0;
    auto stats_to_lds = [&](int grp, int rep) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sa = q_row16_sum(ssum[nt][r]), sb = q_row16_sum(ssq[nt][r]);
                ssum[nt][r] = 0.f; ssq[nt][r] = 0.f;
                if (lx == 0) {
                    const int cl = nt * 16 + gl * 4 + r;
                    red[red_par * 384 + (wave * 2 + 0) * 48 + cl] = sa;
                    red[red_par * 384 + (wave * 2 + 1) * 48 + cl] = sb;
                }
            }
        red_pending = true; red_grp = grp; red_rep = rep; red_par ^= 1;
    };
    auto stats_flush = [&]() {
        if (tid < 2 * C) {
            const int which = tid / C, c = tid - which * C;
            const float* rd = red + (red_par ^ 1) * 384;
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) s += rd[(g * 2 + which) * 48 + c];
            atomicAdd(p.out_stats + (((size_t)red_rep * p.G + red_grp) * 2 + which) * C + c, (mfc_stat_t)s);
        }
        red_pending = false;
    };

    // ---------------- fragment addressing ----------------
    int adx[3], bdx[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int px = lx + dx;
        adx[dx] = pg * MT * ROWA + px * 64 + ((gl ^ (((px >> 2) & 1) << 1)) * 16);
        bdx[dx] = PA + pg * MT * ROWB + px * 48 + gl * 8;
    }
    const int e_lane = ((pg * MT * W + lx) * C) * 2 + gl * 16;          // the lane's 8 channels of cout tiles 0 / 1 (after the transpose), from the tile origin
    const int e2_lane = ((pg * MT * W + lx) * C) * 2 + 64 + gl * 8;     // its 4 channels of cout tile 2
    const int e_row = W * C * 2;
    const char* w16l = w16 + lane * 8;

    // ---------------- prologue: two slots in flight ----------------
    issue(tc0, 0);
    if (nun > 1) issue(tc1, 1);
    q_barrier();                                                 // weight / coefficient tables visible
    vm_wait(nun > 1 ? npw : 0);
    fixup(tc0, 0);
    q_barrier();

    int slot = 0;
    for (int it = 0; it < nun; ++it) {
        const bool has1 = it + 1 < nun, has2 = it + 2 < nun;
        int s1 = slot + 1; if (s1 >= Q_NSLOT) s1 -= Q_NSLOT;
        int s2 = slot + 2; if (s2 >= Q_NSLOT) s2 -= Q_NSLOT;
        // `accumulate`: the old values of this tile are requested in front of the MFMA block that hides them (asm loads hipcc does not
        // count: they are OLDER than the ring slot issued next, so the counted wait behind the MFMAs covers them; conv3x3_ring.hip)
        q_u32x4 pf16[ACC ? MT : 1]; q_u32x2 pf8[ACC ? MT : 1];
        if constexpr (ACC) {
            const int i0 = tc0.tyi * TH, j0 = tc0.txi * 16;
            const char* ab = p.out + (((size_t)tc0.n * H + i0) * W + j0) * C * 2;
            const bool full = (i0 + TH <= H) && (j0 + 16 <= W);
            const bool vx = full || (j0 + lx < W);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool vpx = vx && (full || (i0 + pg * MT + mt < H));
                const unsigned lo = vpx ? (unsigned)(e_lane + mt * e_row) : 0u, lo2 = vpx ? (unsigned)(e2_lane + mt * e_row) : 0u;
                pf16[mt] = (q_u32x4){0u, 0u, 0u, 0u}; pf8[mt] = (q_u32x2){0u, 0u};
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(pf16[mt]) : "v"(lo), "s"(ab));
                asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "+v"(pf8[mt]) : "v"(lo2), "s"(ab));
            }
        }
        if (has2) issue(tc2, s2);

        // ---------------- MFMAs of tile `it`: per tap a k = 32 phase and a k = 16 phase over the wave's MT pixel rows ----------------
        // Software pipeline over the taps: ALL LDS fragments of tap t + 1 (3 weight fragments of the 16-channel remainder, MT + MT pixel
        // fragments) are requested in front of the 6 MT MFMAs of tap t and land under them; sched_barrier keeps hipcc from sinking them next
        // to their use.  The two phases also keep the k = 16 MFMA of an accumulator 3 MT MFMAs behind the k = 32 one that feeds its SrcC:
        // issued back to back the pair returned stale rows 0-1 of the accumulator (different pass counts; hipcc 7.2 inserts no wait states
        // on gfx950).
        if (!(p.ablate & 16)) {
            const char* pb = smem + slot * SLOT;
            bf16x8 b32[2][MT]; q_s16x4 b16[2][MT], a16[2][3];
            auto frags = [&](int tap, int buf) {
                const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) a16[buf][nt] = *(const q_s16x4*)(w16l + (tap * 3 + nt) * 512);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    b32[buf][mt] = *(const bf16x8*)(pb + adx[dx] + (mt + dy) * ROWA);
                    b16[buf][mt] = *(const q_s16x4*)(pb + bdx[dx] + (mt + dy) * ROWB);
                }
            };
            frags(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cur = tap & 1;
                if (tap < 8) frags(tap + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt)
                        if (!(p.ablate & 2)) acc[mt][nt] = mfma16<T>(wr[tap][nt], b32[cur][mt], acc[mt][nt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt)
                        if (!(p.ablate & 1)) acc[mt][nt] = q_mfma_k16<T>(a16[cur][nt], b16[cur][mt], acc[mt][nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the patch of the next tile has landed (this wave's pieces; the slot issued above stays in flight), and so have the old values
        if (has1 || ACC) vm_wait(has2 ? npw : 0);
        if (has1 && !(p.ablate & 64)) fixup(tc1, s1);

        // ---------------- tile epilogue ----------------
        if (!(p.ablate & 32)) {
            const int n = tc0.n, i0 = tc0.tyi * TH, j0 = tc0.txi * 16;
            char* tbase = p.out + (((size_t)n * H + i0) * W + j0) * C * 2;
            const bool full = (i0 + TH <= H) && (j0 + 16 <= W);
            const bool vx = (full || (j0 + lx < W)) && !(p.ablate & 8);
            if constexpr (ACC) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(pf16[mt]), "+v"(pf8[mt]));       // (pins the destinations behind the wait above)
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool vpx = vx && (full || (i0 + pg * MT + mt < H));
                float v[3][4];
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[nt][r] = acc[mt][nt][r]; acc[mt][nt][r] = 0.f; }
                if (p.out_stats) {
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (vpx) { ssum[nt][r] += v[nt][r]; ssq[nt][r] += v[nt][r] * v[nt][r]; }
                }
                if constexpr (ACC) {
                    // the sum is formed in fp32 on the stored layout: transpose the fp32 values of tiles 0 / 1 as raw dwords, add, round once
                    unsigned t0[4], t1[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        auto x32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0][r]), __float_as_uint(v[1][r]), false, false);
                        auto x16 = __builtin_amdgcn_permlane16_swap(x32[0], x32[1], false, false);
                        t0[r] = x16[0]; t1[r] = x16[1];
                    }
                    float o8[8], w[8];
                    Gran<T>::unpack(make_uint4(pf16[mt][0], pf16[mt][1], pf16[mt][2], pf16[mt][3]), o8);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { w[r] = __uint_as_float(t0[r]) + o8[r]; w[4 + r] = __uint_as_float(t1[r]) + o8[4 + r]; }
                    float o4[4];
                    Gran<T>::unquad(make_uint2(pf8[mt][0], pf8[mt][1]), o4);
                    if (vpx) {
                        *(uint4*)(tbase + (unsigned)(e_lane + mt * e_row)) = Gran<T>::pack(w);
                        *(uint2*)(tbase + (unsigned)(e2_lane + mt * e_row)) =
                            make_uint2(q_pack2<T>(v[2][0] + o4[0], v[2][1] + o4[1]), q_pack2<T>(v[2][2] + o4[2], v[2][3] + o4[3]));
                    }
                } else {
                    const unsigned p0 = q_pack2<T>(v[0][0], v[0][1]), p1 = q_pack2<T>(v[0][2], v[0][3]);
                    const unsigned q0 = q_pack2<T>(v[1][0], v[1][1]), q1 = q_pack2<T>(v[1][2], v[1][3]);
                    auto a32 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                    auto a16s = __builtin_amdgcn_permlane16_swap(a32[0], a32[1], false, false);
                    auto b32s = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                    auto b16s = __builtin_amdgcn_permlane16_swap(b32s[0], b32s[1], false, false);
                    if (vpx) {
                        *(uint4*)(tbase + (unsigned)(e_lane + mt * e_row)) = make_uint4(a16s[0], b16s[0], a16s[1], b16s[1]);
                        *(uint2*)(tbase + (unsigned)(e2_lane + mt * e_row)) = make_uint2(q_pack2<T>(v[2][0], v[2][1]), q_pack2<T>(v[2][2], v[2][3]));
                    }
                }
            }
            if (p.out_stats) {
                if (!has1 || (tc1.n / p.ipg != n / p.ipg)) stats_to_lds(n / p.ipg, (Lb + it) % MFC_R);
            }
        }

        // ---------------- hand over to the next tile ----------------
        q_barrier();
        if (red_pending) stats_flush();
        tc0 = tc1; tc1 = tc2; tc2 = tc_next(tc2);
        slot = s1;
    }
    if (red_pending) stats_flush();
}

// ------------------------------------------------------------------------------------------
bool ring48_eligible(const mfc_conv_desc* d) {
    if (!g_conv_ring48 || !d || !mfc_is16(d->dtype)) return false;
    if (d->TA != 3 || d->TB != 3 || d->dh0 != -1 || d->dw0 != -1 || d->in_stride != 1) return false;
    if (d->out_sh != 1 || d->out_sw != 1 || d->out_oh != 0 || d->out_ow != 0) return false;
    if (d->Hin != d->Hout || d->Win != d->Wout || d->Hl != d->Hout || d->Wl != d->Wout) return false;
    if (d->Cin != Q_C || d->Cout != Q_C || d->Cin_p != Q_C || d->Cout_p != Q_C) return false;
    if (d->bias || d->TH > 0 || d->TW > 0 || d->acc_src || d->bn_y) return false;
    if (d->N <= 0 || d->images_per_group <= 0 || d->N % d->images_per_group || d->N / d->images_per_group > 8) return false;
    if (d->Hin < 2 || d->Win < 2) return false;
    if ((double)d->Hin * d->Win * Q_C * 2.0 >= 2.0e9) return false;             // 32-bit lane offsets inside one image
    return true;
}

int g_ring48_mt = 2;                  // rows of 16 pixels per wave (mfc_set_flag(51, 2 | 4))
template <int MT>
static int ring48_setup_t(const mfc_conv_desc* d, Ring48K& k, size_t& lds, int& grid) {
    typedef Ring48Geo<MT> Geo;
    if (!d->in || !d->wp || !d->out) return MFC_ERR_INVALID_ARG;
    k.in = (const char*)d->in; k.wp = (const char*)d->wp; k.out = (char*)d->out;
    k.in_coef = d->in_coef; k.out_stats = d->out_stats; k.in_fin = (const mfc_bnfin_desc*)d->in_fin;
    k.N = d->N; k.H = d->Hout; k.W = d->Wout;
    k.in_relu = d->in_relu; k.ipg = d->images_per_group; k.G = d->N / d->images_per_group; k.accumulate = d->accumulate;
    k.ablate = g_ring_ablate;
    k.tilesY = ceil_div(d->Hout, Geo::TH); k.tilesX = ceil_div(d->Wout, 16);
    k.ntiles = d->N * k.tilesY * k.tilesX;
    k.off_w16 = Geo::RING;
    k.off_coef = k.off_w16 + 9 * 3 * 512;
    k.off_red = k.off_coef + (d->in_coef ? k.G * 2 * Q_C * 4 : 0);
    lds = (size_t)k.off_red + (d->out_stats ? 2 * 384 * 4 : 0);
    grid = lds <= 80 * 1024 ? 512 : 256;                         // two workgroups per CU where 63 KiB of ring + weights + tables fit twice
    if (grid > k.ntiles) grid = k.ntiles;
    k.per_block = ceil_div(k.ntiles, grid);
    grid = ceil_div(k.ntiles, k.per_block);
    return MFC_OK;
}
static int ring48_setup(const mfc_conv_desc* d, Ring48K& k, size_t& lds, int& grid) {
    return g_ring48_mt == 4 ? ring48_setup_t<4>(d, k, lds, grid) : ring48_setup_t<2>(d, k, lds, grid);
}

int ring48_layout(const mfc_conv_desc* d, mfc_conv_layout* out) {
    Ring48K k; size_t lds; int grid;
    mfc_conv_desc t = *d;
    if (!t.in) t.in = (const void*)16;
    if (!t.wp) t.wp = (const void*)16;
    if (!t.out) t.out = (void*)16;
    const int rc = ring48_setup(&t, k, lds, grid);
    if (rc < 0) return rc;
    out->KG = 6; out->nchunks = 1; out->NT16 = Q_C; out->Yblocks = 1; out->nslots = 54;
    out->TA = 3; out->TB = 3; out->lds_bytes = (int32_t)lds; out->TAS = 3;
    out->bytes = (int64_t)9 * 6 * Q_C * 16;
    out->MT = g_ring48_mt == 4 ? 4 : 2; out->TH = 4 * out->MT; out->TW = 16; out->grid = grid; out->per_block = k.per_block; out->NW = 4;
    out->fa = 0;
    return MFC_OK;
}

template <typename T, int MT, bool ACC>
static int ring48_launch_t(const Ring48K& k, size_t lds, int grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_ring48_kernel<T, MT, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        MFC_PROF_NAME(pname, "conv3x3_ring48_kernel<%s, %d, %s>", mfc_tname<T>(), MT, ACC ? "true" : "false");
        const double px = (double)k.N * k.H * k.W;
        mfc_prof_before(st, pname, 2.0 * px * 9.0 * Q_C * Q_C, px * 2.0 * Q_C * 2.0);
    }
    hipLaunchKernelGGL((conv3x3_ring48_kernel<T, MT, ACC>), dim3(grid), dim3(256), lds, st, k);
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

int ring48_launch(const mfc_conv_desc* d, hipStream_t st) {
    Ring48K k; size_t lds; int grid;
    const int rc = ring48_setup(d, k, lds, grid);
    if (rc < 0) return rc;
    int r = MFC_ERR_UNSUPPORTED;
    if (g_ring48_mt == 4) {
        if (d->accumulate) MFC_TYPED16(d->dtype, T_, r = (ring48_launch_t<T_, 4, true>(k, lds, grid, st)));
        else MFC_TYPED16(d->dtype, T_, r = (ring48_launch_t<T_, 4, false>(k, lds, grid, st)));
    } else {
        if (d->accumulate) MFC_TYPED16(d->dtype, T_, r = (ring48_launch_t<T_, 2, true>(k, lds, grid, st)));
        else MFC_TYPED16(d->dtype, T_, r = (ring48_launch_t<T_, 2, false>(k, lds, grid, st)));
    }
    return r;
}
