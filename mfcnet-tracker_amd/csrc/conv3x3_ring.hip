// 3x3 / stride-1 / pad-1 convolutions of the HRNet BasicBlocks with 32 or 64 channels on both sides (16-bit storage), forward and
// data gradient: models/hrnet.py:58-74 (conv1 / conv2 of every BasicBlock of the 120x160 and 60x80 branches, W32), :95-115
// (Bottleneck conv2) and the cuDNN dgrad of those calls under loss.backward() (src/engine.py:70).
//
// conv_igemm.hip runs these shapes with all waves of the chip in the same phase (load issue -> wait -> MFMA -> epilogue; the MFMA
// loop is 28-41 % of a tile, profiles/r02_d_conv_trace_cycles.log) and needs 246 VGPRs / 80 KiB per workgroup.  This kernel is built
// the other way round:
//   * the WEIGHTS live in registers for the whole launch (9 taps x C/32 k-steps x 2 cout tiles = 72 / 144 VGPRs per wave): no
//     weight stages, no weight reads from LDS, no k-table -- the only LDS traffic of the MFMA loop is one ds_read_b128 of pixels
//     per two MFMAs;
//   * the input halo patch of a pixel tile goes global -> LDS by DMA (global_load_lds_dwordx4) into a ring of THREE slots, issued
//     TWO tiles ahead of the MFMAs that read it (40-46 KiB in flight per workgroup, two workgroups per CU), completion by a
//     counted s_waitcnt vmcnt; no registers and no VALU address arithmetic on the staging path of an interior tile;
//   * an LDS-DMA writes 64 x 16 contiguous bytes, so pixel rows cannot be padded against bank conflicts: a pixel is 64 B per
//     32-channel plane and the two 32-byte halves of the pixels whose patch column has bit 2 set are swapped -- on the SOURCE side
//     of the DMA (each lane fetches the granule that belongs in its slot), which makes every tap's ds_read_b128 conflict free;
//   * one raw s_barrier per tile (s_waitcnt lgkmcnt only, so the DMA ring stays in flight across it); the producer's BatchNorm +
//     ReLU and the zero padding are applied to a landed patch in place in LDS (only where needed: interior tiles of a launch
//     without input transform are never touched);
//   * workgroup = 4 waves = (4 / NCH pixel groups) x (NCH = C/32 cout halves); a wave owns MT rows of 16 pixels x 32 couts;
//     epilogue as in conv_igemm (FA variants): sum / sum of squares in registers across tiles, cout tile pairs transposed across
//     the 16-lane rows so that every lane stores 16 contiguous bytes; data-gradient fusions (accumulate / acc_src / bn_y) included.
// Dispatched from mfc_conv2d_fwd / mfc_conv2d_layout when ring_eligible(); mfc_set_flag(30, 0) sends these launches back to
// conv_igemm.hip.
#include "common.h"

int g_conv_ring = 1;
int g_ring_ablate = 0;
int g_ring_stagger = 0;
int g_ring_wgs = 2;                       // workgroups per CU the grid is sized for (tuning: mfc_set_flag(33, n))
long g_ring_c64_unfused_px = 100000;       // pixels from which a 64-channel data gradient prefers the unfused ring launch (tuning: mfc_set_flag(57, px / 1000)).  Measured at N = 24:
                                          // 120x160 (layer1) -0.1 ms per step, 60x80 (the 64-channel branch's 32 conv2 gradients) another -0.4 ms (34.5 -> 34.1)
int g_ring_grid = 0;                      // > 0: workgroups per launch (tuning / probes: mfc_set_flag(52, n)); 0 = 256 * g_ring_wgs

struct RingK {
    const char* in; const char* wp; char* out;
    const float* in_coef; mfc_stat_t* out_stats;
    const char* acc_src; const char* bn_y; const float* bn_coef; const unsigned char* bn_bits;
    int N, H, W, C;
    int in_relu, ipg, G, accumulate, bn_mode;
    int tilesY, tilesX, ntiles, per_block;
    int off_coef, off_bnc, off_red;          // LDS byte offsets behind the ring
    int stagger;                             // cycles the SECOND workgroup of a CU waits before it starts (de-phases the two; mfc_set_flag(37, cycles))
    int wt;                                  // write-through output stores (common.h: large outputs only)
    int ablate;                              // tuning only (mfc_set_flag(32, mask)): 1 skip MFMAs, 2 skip stores, 4 skip DMA, 8 skip fix-up, 16 skip statistics
};

template <int CTRL> __device__ inline float r_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ inline float r_row16_sum(float v) {      // sum over the 16 lanes of a DPP row (result in every lane)
    v = r_dpp_add<0xB1>(v); v = r_dpp_add<0x4E>(v); v = r_dpp_add<0x141>(v); v = r_dpp_add<0x140>(v);
    return v;
}
// one 1-KiB LDS-DMA piece: lane l copies 16 B from (base + voff) to LDS byte (lds + 16 l)
__device__ inline void r_dma(const char* base, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds) : "memory");
}
// the same with a full 64-bit address per lane (border tiles of launches without input transform: out-of-image pixels fetch a zero granule)
__device__ inline void r_dma64(const char* addr, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(addr), "s"(lds) : "memory");
}
__device__ __attribute__((aligned(16))) const unsigned r_zero16[4] = {0u, 0u, 0u, 0u};
// workgroup barrier that leaves vector-memory operations (the DMA ring, the epilogue's stores) in flight: LDS traffic only
__device__ inline void r_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// output store of the plain epilogue: write-through (common.h, mfc_st16) unless the ablation mask asks for the write-back (32) or nt (64) form
__device__ inline void r_store16(char* addr, uint4 v, int mode) {
    if (mode == 1) {
        *(uint4*)addr = v;
    } else if (mode == 2) {
        __builtin_nontemporal_store(v.x, (unsigned*)addr); __builtin_nontemporal_store(v.y, (unsigned*)addr + 1);
        __builtin_nontemporal_store(v.z, (unsigned*)addr + 2); __builtin_nontemporal_store(v.w, (unsigned*)addr + 3);
    } else {
        mfc_st16(addr, v);
    }
}
constexpr int R_PW = 18;                    // patch width: 16-pixel tile rows + halo
constexpr int R_NSLOT = 3;                  // ring slots

template <int NCH, int MT> struct RingGeo {
    static constexpr int NPG = 4 / NCH;                         // pixel groups (waves along the tile's rows)
    static constexpr int TH = NPG * MT;                         // tile rows
    static constexpr int PH = TH + 2;
    static constexpr int NPX = PH * R_PW;
    static constexpr int PLANE = ((NPX * 64 + 1023) / 1024) * 1024;      // one 32-channel plane of a slot (whole DMA pieces)
    static constexpr int PPP = PLANE / 1024;
    static constexpr int NP = NCH * PPP;                        // DMA pieces per slot
    static constexpr int NPW = (NP + 3) / 4;                    // ... per wave
    static constexpr int SLOT = NCH * PLANE;
    static constexpr int RING = R_NSLOT * SLOT;
};

template <typename T, int NCH, int MT, bool FUSE>
__global__ __launch_bounds__(256, 2) void conv3x3_ring_kernel(RingK p) {
    typedef RingGeo<NCH, MT> Geo;
    constexpr int C = 32 * NCH, KG = C / 8;
    constexpr int NPG = Geo::NPG, TH = Geo::TH, PH = Geo::PH, NPX = Geo::NPX, PLANE = Geo::PLANE, PPP = Geo::PPP, NP = Geo::NP, NPW = Geo::NPW,
                  SLOT = Geo::SLOT;
    constexpr int ROWB = R_PW * 64;                              // bytes of one patch row in a plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* coefl = (float*)(smem + p.off_coef);                  // [G][2][C] scale / shift of the fused input transform
    float* bncl = (float*)(smem + p.off_bnc);                    // [G][4][C] coefficient block of the fused BatchNorm backward
    float* red = (float*)(smem + p.off_red);                     // [2 parities][4 waves][2][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave / NCH, ch = wave % NCH;
    const int gl = lane >> 4, lx = lane & 15;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int u0 = Lb * p.per_block;
    const int nun = min(p.per_block, p.ntiles - u0);
    if (nun <= 0) return;
    const int H = p.H, W = p.W;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    if (p.stagger > 0) {
        // Two workgroups share a CU and start together: left alone they issue their DMA, run their MFMAs and store their tiles in the
        // same phase.  The one in the odd wave slot of its SIMDs (HW_ID.wave_id bit 0: speed only, never correctness) starts late.
        const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));      // HW_REG_HW_ID, bits [3:0] = wave slot
        if (hwid & 1u) for (int t = 0; t < p.stagger; t += 1024) __builtin_amdgcn_s_sleep(16);
    }

    // ---------------- tile cursors (tracked incrementally) ----------------
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

    // ---------------- DMA tables: LDS slot 64 i + lane of piece (wave + 4 i) -> source granule ----------------
    // interior tiles: byte offset from the first patch pixel (constant per lane: kept in registers where the resident weights leave
    // room, i.e. for 32 channels); border tiles recompute with clamping.  Recomputed values hang on an opaque copy of the lane id,
    // so that hipcc does not hoist them out of the tile loop (it then spills them: 38 VGPRs of scratch in the 64-channel variant)
    constexpr bool KEEPV = (NCH == 1);
    const int npw = (NP - wave + 3) / 4;                         // pieces this wave issues per slot (wave-uniform)
    auto piece_src = [&](int i, int ln, int& py, int& px, int& g) {
        const int pid = wave + 4 * i;
        const int kc = pid / PPP, S = (pid - kc * PPP) * 64 + ln;
        const int q = min(S >> 2, NPX - 1), sp = S & 3;
        py = q / R_PW; px = q - py * R_PW;
        g = (sp ^ (((px >> 2) & 1) << 1)) + 4 * kc;
    };
    int voff[KEEPV ? NPW : 1], vpk[KEEPV ? NPW : 1];              // vpk = py | px << 8 | g << 16 of the piece's slot
    if constexpr (KEEPV) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            int py, px, g;
            piece_src(i, lane, py, px, g);
            voff[i] = ((py * W + px) * C + g * 8) * 2;
            vpk[i] = py | (px << 8) | (g << 16);
        }
    }
    auto issue = [&](const TC& c, int slot) {
        const int i0 = c.tyi * TH, j0 = c.txi * 16;
        const char* img = p.in + (size_t)c.n * H * W * C * 2;
        const unsigned lbase = lds0 + slot * SLOT + wave * 1024;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (p.ablate & 4) return;
        if (tc_interior(c)) {           // wave-uniform
            const char* pb = img + ((size_t)(i0 - 1) * W + (j0 - 1)) * C * 2;
#pragma unroll
            for (int i = 0; i < NPW; ++i)
                if (i < npw) {
                    unsigned vo;
                    if constexpr (KEEPV) vo = (unsigned)voff[i];
                    else { int py, px, g; piece_src(i, ln, py, px, g); vo = (unsigned)(((py * W + px) * C + g * 8) * 2); }
                    r_dma(pb, vo, lbase + i * 4096);
                }
        } else if (!p.in_coef) {        // no input transform: out-of-image pixels FETCH zeros (one zero granule, full address per lane) -- no fix-up pass at all
            const char* zp = (const char*)r_zero16;
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                if (i < npw) {
                    int py, px, g;
                    piece_src(i, ln, py, px, g);
                    const int iy = i0 - 1 + py, ix = j0 - 1 + px;
                    const bool inr = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    r_dma64(inr ? img + (size_t)(unsigned)(((iy * W + ix) * C + g * 8) * 2) : zp, lbase + i * 4096);
                }
            }
        } else {                        // clamped (always valid) addresses; the out-of-image pixels are zeroed after landing, behind the transform
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                if (i < npw) {
                    int py, px, g;
                    piece_src(i, ln, py, px, g);
                    const int iy = min(max(i0 - 1 + py, 0), H - 1), ix = min(max(j0 - 1 + px, 0), W - 1);
                    r_dma(img, (unsigned)(((iy * W + ix) * C + g * 8) * 2), lbase + i * 4096);
                }
            }
        }
    };
    // all but this wave's `k` youngest vector-memory operations are done (k = pieces of the slot issued last, or 0)
    auto vm_wait = [&](int k) {
        if (k >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (k == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (k == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (k == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (k == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (k == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(NPW <= 6, "vm_wait covers up to six pieces per wave");

    // ---------------- weights -> registers (A operand: row = cout, k = cin) ----------------
    // packed image [tap][granule][cout][16 B] (mfc_conv2d_layout: nchunks = Yblocks = 1, NT16 = C, nslots = 9 C/8)
    bf16x8 wr[9][NCH][2];
    {
        const char* ws = p.wp + ((size_t)gl * C + ch * 32 + lx) * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    wr[t][kc][nt] = *(const bf16x8*)(ws + ((size_t)((t * KG + kc * 4) * C) + nt * 16) * 16);
    }
    const int st_mode = ((p.ablate & 32) || !p.wt) ? 1 : ((p.ablate & 64) ? 2 : 0);
    const bool xf = (p.in_coef != nullptr);
    if (xf) {
        for (int i = tid; i < p.G * 2 * C; i += 256) {
            const int g = i / (2 * C), r = i - g * 2 * C;
            coefl[i] = p.in_coef[(size_t)g * 4 * C + r];          // rows 0 (scale) and 1 (shift) of [G][4][C]
        }
    }
    const bool bnm = FUSE && p.bn_y != nullptr;
    if constexpr (FUSE) {
        // rows 0 / 1: scale / shift (mask of bn_mode 2); row 2 is stored as -mean * rstd next to row 3 = rstd: yhat = y * rstd + row2, one fma per element
        if (bnm) for (int i = tid; i < p.G * 4 * C; i += 256) {
            const int r = (i / C) & 3;
            bncl[i] = r == 2 ? -p.bn_coef[i] * p.bn_coef[i + C] : p.bn_coef[i];
        }
    }

    // ---------------- in-LDS fix-up of a landed slot: zero what lies outside the image; the producer's BatchNorm + ReLU ----------------
    // Every wave fixes up the bytes its OWN DMA pieces wrote, right behind its own vmcnt wait: no barrier between landing and fix-up
    // (the slot is not read by anybody before the barrier that ends the tile), one barrier per tile whatever the launch fuses.
    auto fixup = [&](const TC& c, int slot) {
        const bool interior = tc_interior(c);
        if (!xf || (p.ablate & 8)) return;               // (without a transform the border pixels arrived as zeros: issue())
        const int i0 = c.tyi * TH, j0 = c.txi * 16;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const float* cfg = coefl + (c.n / p.ipg) * 2 * C;
        char* lb = smem + slot * SLOT + wave * 1024 + ln * 16;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            if (i < npw) {
                int py, px, g;
                if constexpr (KEEPV) { py = vpk[i] & 0xff; px = (vpk[i] >> 8) & 0xff; g = vpk[i] >> 16; }
                else piece_src(i, ln, py, px, g);
                const bool inr = interior || ((unsigned)(i0 - 1 + py) < (unsigned)H && (unsigned)(j0 - 1 + px) < (unsigned)W);
                char* a = lb + i * 4096;
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
    };

    // ---------------- accumulators / statistics ----------------
    f32x4 acc[MT][2];
    float ssum[2][4], ssq[2][4];          // running per-lane statistics partials (flushed when the statistic group changes)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }
    }
    int red_par = 0; bool red_pending = false; int red_grp = 0, red_rep = 0;
    auto stats_to_lds = [&](int grp, int rep) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sa = r_row16_sum(ssum[nt][r]), sb = r_row16_sum(ssq[nt][r]);
                ssum[nt][r] = 0.f; ssq[nt][r] = 0.f;
                if (lx == 0) {
                    // (BatchNorm-backward mode keeps the sums in the TRANSPOSED layout of the epilogue: entry [nt][r] of row gl is channel 8 gl + 4 nt + r)
                    const int cl = bnm ? (gl * 8 + nt * 4 + r) : (nt * 16 + gl * 4 + r);
                    red[red_par * 256 + (wave * 2 + 0) * 32 + cl] = sa;
                    red[red_par * 256 + (wave * 2 + 1) * 32 + cl] = sb;
                }
            }
        red_pending = true; red_grp = grp; red_rep = rep; red_par ^= 1;
    };
    auto stats_flush = [&]() {
        if (tid < 2 * C) {
            const int which = tid / C, c = tid - which * C;
            const int hc = c >> 5, cl = c & 31;
            const float* rd = red + (red_par ^ 1) * 256;
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < NPG; ++g) s += rd[((g * NCH + hc) * 2 + which) * 32 + cl];
            atomicAdd(p.out_stats + (((size_t)red_rep * p.G + red_grp) * 2 + which) * C + c, (mfc_stat_t)s);
        }
        red_pending = false;
    };

    // ---------------- fragment addressing: patch row base of the wave + per-tap-column lane offsets ----------------
    int adx[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int px = lx + dx;
        adx[dx] = pg * MT * ROWB + px * 64 + ((gl ^ (((px >> 2) & 1) << 1)) * 16);
    }
    const int e_lane = ((pg * MT * W + lx) * C) * 2 + ch * 64 + gl * 16;      // byte offset of the lane's 8 output channels from the tile origin
    const int e_row = W * C * 2;

    // ---------------- prologue: two slots in flight ----------------
    issue(tc0, 0);
    if (nun > 1) issue(tc1, 1);
    r_barrier();                                                 // coefficient tables visible
    vm_wait(nun > 1 ? npw : 0);
    fixup(tc0, 0);
    r_barrier();

    int slot = 0;
    for (int it = 0; it < nun; ++it) {
        const bool has1 = it + 1 < nun, has2 = it + 2 < nun;
        int s1 = slot + 1; if (s1 >= R_NSLOT) s1 -= R_NSLOT;
        int s2 = slot + 2; if (s2 >= R_NSLOT) s2 -= R_NSLOT;
        // (slot s2 was read by the MFMAs of the previous tile; every wave has passed the barrier behind them)
        // data-gradient fusions: ALL global reads of this tile's epilogue (running sum, pre-BN tensor, mask bits) are issued here, in
        // front of the MFMA loop that hides them, as asm loads hipcc does not count: they are OLDER than the ring slot issued next, so
        // the one counted wait behind the MFMA loop covers them and still leaves that slot in flight (through compiler-visible loads
        // the epilogue would wait vmcnt(0) and drain the ring; issued next to their use they are 16 exposed round trips per tile:
        // 36.7 us for a 18 us launch)
        u32x4 pf_old[FUSE ? MT : 1], pf_y[FUSE ? MT : 1]; unsigned pf_bits[FUSE ? MT : 1];
        if constexpr (FUSE) {
            const int i0 = tc0.tyi * TH, j0 = tc0.txi * 16;
            const size_t toff = (((size_t)tc0.n * H + i0) * W + j0) * C * 2;
            const char* ab = (p.acc_src ? p.acc_src : (const char*)p.out) + toff;
            const char* yb = p.bn_y + toff;
            const unsigned char* bb = p.bn_bits + (toff >> 4);
            const bool full = (i0 + TH <= H) && (j0 + 16 <= W);
            const bool vx = full || (j0 + lx < W);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool vpx = vx && (full || (i0 + pg * MT + mt < H));
                const unsigned lo = vpx ? (unsigned)(e_lane + mt * e_row) : 0u;
                pf_old[mt] = (u32x4){0u, 0u, 0u, 0u}; pf_y[mt] = (u32x4){0u, 0u, 0u, 0u}; pf_bits[mt] = 0xffu;
                if (p.accumulate) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(pf_old[mt]) : "v"(lo), "s"(ab));
                if (bnm) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(pf_y[mt]) : "v"(lo), "s"(yb));
                if (bnm && p.bn_mode == 3) asm volatile("s_nop 4\n\tglobal_load_ubyte %0, %1, %2" : "+v"(pf_bits[mt]) : "v"(lo >> 4), "s"(bb));
            }
        }
        if (has2) issue(tc2, s2);

        // ---------------- MFMAs of tile `it` ----------------
        // one ds_read_b128 feeds two MFMAs; the reads run RD steps ahead of their MFMAs in a rolling window of RD + 1 fragments
        // (pinned with sched_group_barrier: left alone, hipcc hoists dozens of reads and spills the resident weights)
        {
            const char* pb = smem + slot * SLOT;
            constexpr int NK = MT * 9 * NCH, RD = 3;
            if (!(p.ablate & 1)) {
            bf16x8 bw[RD + 1];
            auto rd = [&](int s) {
                const int mt = s / (9 * NCH), r9 = s - mt * 9 * NCH;
                const int tap = r9 / NCH, kc = r9 - tap * NCH;
                const int dy = tap / 3, dx = tap - dy * 3;
                return *(const bf16x8*)(pb + adx[dx] + (kc * PLANE + (mt + dy) * ROWB));
            };
#pragma unroll
            for (int s = 0; s < NK + RD; ++s) {
                if (s < NK) bw[s % (RD + 1)] = rd(s);
                if (s >= RD) {
                    const int q = s - RD;
                    const int mt = q / (9 * NCH), r9 = q - mt * 9 * NCH;
                    const int tap = r9 / NCH, kc = r9 - tap * NCH;
                    acc[mt][0] = mfma16<T>(wr[tap][kc][0], bw[q % (RD + 1)], acc[mt][0]);
                    acc[mt][1] = mfma16<T>(wr[tap][kc][1], bw[q % (RD + 1)], acc[mt][1]);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
            for (int s = 0; s < NK - RD; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * RD, 0);
            }
        }
        // the patch of the next tile has landed (this wave's pieces; the slot issued above stays in flight), and so have the epilogue's operands
        if (has1 || FUSE) vm_wait(has2 ? npw : 0);
        if (has1) fixup(tc1, s1);

        // ---------------- tile epilogue ----------------
        {
            const int n = tc0.n, i0 = tc0.tyi * TH, j0 = tc0.txi * 16;
            const size_t toff = (((size_t)n * H + i0) * W + j0) * C * 2;
            char* tbase = p.out + toff;
            const bool full = (i0 + TH <= H) && (j0 + 16 <= W);
            const bool vx = (full || (j0 + lx < W)) && !(p.ablate & 2);
            if constexpr (FUSE) {
                // (the operands were requested in front of the MFMA loop; the wait above covers them -- the empty statement pins every
                //  destination register behind that wait, form (ii) of the guide's section 5.7)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(pf_old[mt]), "+v"(pf_y[mt]), "+v"(pf_bits[mt]));
                const float* cf0 = bncl + (n / p.ipg) * 4 * C + ch * 32 + gl * 8;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    __builtin_amdgcn_sched_barrier(0);            // rows one after the other: interleaved they need twice the registers
                    int zo = 0;
                    asm volatile("" : "+v"(zo));                  // (opaque: the coefficient reads stay inside the row instead of being hoisted above the tile)
                    const float* cf = cf0 + zo;
                    const bool vpx = vx && (full || (i0 + pg * MT + mt < H));
                    float v[2][4];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[nt][r] = acc[mt][nt][r]; acc[mt][nt][r] = 0.f; }
                    const unsigned off = (unsigned)(e_lane + mt * e_row);
                    if (p.accumulate || bnm) {
                        // accumulation / masking happen in fp32 on the transposed layout: transpose the fp32 values as raw dwords, add, round once
                        float w[8];
                        unsigned t0[4], t1[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            auto x32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0][r]), __float_as_uint(v[1][r]), false, false);
                            auto x16 = __builtin_amdgcn_permlane16_swap(x32[0], x32[1], false, false);
                            t0[r] = x16[0]; t1[r] = x16[1];
                        }
                        if (!bnm && p.out_stats) {
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (vpx) { ssum[nt][r] += v[nt][r]; ssq[nt][r] += v[nt][r] * v[nt][r]; }
                        }
                        float o8[8];
                        Gran<T>::unpack(make_uint4(pf_old[mt][0], pf_old[mt][1], pf_old[mt][2], pf_old[mt][3]), o8);          // (zeros without `accumulate`)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { w[r] = __uint_as_float(t0[r]) + o8[r]; w[4 + r] = __uint_as_float(t1[r]) + o8[4 + r]; }
                        if (bnm) {
                            // BatchNorm / ReLU backward of the tensor this launch completes: mask, store the MASKED gradient, keep
                            // sum g*m and sum g*m*yhat of the lane's 8 channels (what mfc_bnbwd_reduce would sweep the tensor for again)
                            float yv[8];
                            Gran<T>::unpack(make_uint4(pf_y[mt][0], pf_y[mt][1], pf_y[mt][2], pf_y[mt][3]), yv);
                            unsigned mk = 0xffu;
                            if (p.bn_mode == 2) {
                                mk = 0;
#pragma unroll
                                for (int e = 0; e < 8; ++e) mk |= ((yv[e] * cf[e] + cf[C + e]) > 0.f ? 1u : 0u) << e;
                            } else if (p.bn_mode == 3) {
                                mk = pf_bits[mt];                 // one byte per 8-channel granule (mfc_combine_fwd)
                            }
                            if (!vpx) mk = 0;                     // (pixels outside the image: masked out, so the sums below need no select; their y is finite, clamped address)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float gmv = ((mk >> e) & 1u) ? w[e] : 0.f;
                                w[e] = gmv;
                                ssum[e >> 2][e & 3] += gmv;
                                ssq[e >> 2][e & 3] += gmv * (yv[e] * cf[3 * C + e] + cf[2 * C + e]);
                            }
                        }
                        if (vpx) mfc_st16_if((tbase + off), Gran<T>::pack(w), p.wt);
                        continue;
                    }
                    // (a FUSE launch without accumulate / bn_y stores like the plain kernel)
                    const unsigned p0 = pack2<T>(v[0][0], v[0][1]), p1 = pack2<T>(v[0][2], v[0][3]);
                    const unsigned q0 = pack2<T>(v[1][0], v[1][1]), q1 = pack2<T>(v[1][2], v[1][3]);
                    auto a32 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                    auto a16 = __builtin_amdgcn_permlane16_swap(a32[0], a32[1], false, false);
                    auto b32 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                    auto b16 = __builtin_amdgcn_permlane16_swap(b32[0], b32[1], false, false);
                    if (p.out_stats) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (vpx) { ssum[nt][r] += v[nt][r]; ssq[nt][r] += v[nt][r] * v[nt][r]; }
                    }
                    if (vpx) mfc_st16_if((tbase + off), make_uint4(a16[0], b16[0], a16[1], b16[1]), p.wt);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bool vpx = vx && (full || (i0 + pg * MT + mt < H));
                    float v[2][4];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[nt][r] = acc[mt][nt][r]; acc[mt][nt][r] = 0.f; }
                    // the MFMA leaves a lane with 4 channels of each cout tile; the two tiles are transposed across the four 16-lane rows
                    // (v_permlane32_swap + v_permlane16_swap) so that every lane owns 8 CONTIGUOUS channels of its pixel: one 16-byte store
                    const unsigned p0 = pack2<T>(v[0][0], v[0][1]), p1 = pack2<T>(v[0][2], v[0][3]);
                    const unsigned q0 = pack2<T>(v[1][0], v[1][1]), q1 = pack2<T>(v[1][2], v[1][3]);
                    auto a32 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                    auto a16 = __builtin_amdgcn_permlane16_swap(a32[0], a32[1], false, false);
                    auto b32 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                    auto b16 = __builtin_amdgcn_permlane16_swap(b32[0], b32[1], false, false);
                    if (p.out_stats) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (vpx) { ssum[nt][r] += v[nt][r]; ssq[nt][r] += v[nt][r] * v[nt][r]; }
                    }
                    if (vpx) r_store16(tbase + (unsigned)(e_lane + mt * e_row), make_uint4(a16[0], b16[0], a16[1], b16[1]), st_mode);
                }
            }
            if (p.out_stats) {
                // flush the running sums when the next tile belongs to another statistic group or the run ends
                if (!has1 || (tc1.n / p.ipg != n / p.ipg)) stats_to_lds(n / p.ipg, (Lb + it) % MFC_R);
            }
        }

        // ---------------- hand over to the next tile ----------------
        r_barrier();                                             // everybody's pieces of the next patch have landed and are fixed up; everybody is done with this slot
        if (red_pending) stats_flush();
        tc0 = tc1; tc1 = tc2; tc2 = tc_next(tc2);
        slot = s1;
    }
    if (red_pending) stats_flush();
}

// ------------------------------------------------------------------------------------------
bool ring_eligible(const mfc_conv_desc* d) {
    if (!g_conv_ring || !d || !mfc_is16(d->dtype) || (d->flags & MFC_CONV_S2_CLASSES)) return false;
    if (d->TA != 3 || d->TB != 3 || d->dh0 != -1 || d->dw0 != -1 || d->in_stride != 1) return false;
    if (d->out_sh != 1 || d->out_sw != 1 || d->out_oh != 0 || d->out_ow != 0) return false;
    if (d->Hin != d->Hout || d->Win != d->Wout || d->Hl != d->Hout || d->Wl != d->Wout) return false;
    if (d->Cin != d->Cout || d->Cin_p != d->Cin || d->Cout_p != d->Cout || (d->Cin != 32 && d->Cin != 64)) return false;
    if (d->bias || d->TH > 0 || d->TW > 0) return false;
    // (64 channels with the data-gradient epilogue fusions: 144 VGPRs of resident weights + the fused epilogue's operands do not fit in
    //  256 registers -- hipcc spills 200 of them; those launches stay on conv_igemm.hip)
    if (d->Cin == 64 && !(d->flags & MFC_CONV_NEVER_ACC) && (d->acc_src || d->bn_y || d->accumulate)) return false;
    // a 64-channel data gradient that ASKS for a fusable launch (MFC_CONV_WANT_FA) gets conv_igemm's -- except on large images, where the plain ring launch
    // plus a separate reduce pass is faster than conv_igemm's fused one (layer1's conv2 at 120x160, N = 24: 56 + 30 us against 119 us; at 60x80 17 + 12 against
    // 31, and the ring launches share the chip better with the other lanes); mfc_conv2d_layout
    // then reports fa = 0 and the planner keeps the reduce record
    // (only for descriptors that promise never to accumulate, MFC_CONV_NEVER_ACC: the kernel choice -- and with it the packed weight layout -- must not
    //  depend on a field the planner fills in later)
    if (d->Cin == 64 && (d->flags & MFC_CONV_WANT_FA) && (!(d->flags & MFC_CONV_NEVER_ACC) || (long)d->N * d->Hin * d->Win < g_ring_c64_unfused_px)) return false;
    if (d->N <= 0 || d->images_per_group <= 0 || d->N % d->images_per_group || d->N / d->images_per_group > 8) return false;
    if (d->Hin < 2 || d->Win < 2) return false;
    if ((double)d->Hin * d->Win * d->Cin * 2.0 >= 2.0e9) return false;          // 32-bit lane offsets inside one image
    return true;
}

static int g_ring_mt = 4;            // rows of 16 pixels per wave (tuning: mfc_set_flag(31, 2 | 4))
int mfc_ring_set_mt(int v) { g_ring_mt = (v == 2) ? 2 : 4; return 0; }

template <int NCH, int MT>
static void ring_geo(const mfc_conv_desc* d, RingK& k, size_t& lds, int& grid) {
    typedef RingGeo<NCH, MT> Geo;
    k.tilesY = ceil_div(d->Hout, Geo::TH); k.tilesX = ceil_div(d->Wout, 16);
    k.ntiles = d->N * k.tilesY * k.tilesX;
    const int C = 32 * NCH, G = d->N / d->images_per_group;
    k.off_coef = Geo::RING;
    k.off_bnc = k.off_coef + G * 2 * C * 4;
    k.off_red = k.off_bnc + G * 4 * C * 4;
    lds = (size_t)k.off_red + 2 * 256 * 4;
    grid = g_ring_grid > 0 ? g_ring_grid : 256 * g_ring_wgs;     // two workgroups per CU
    if (grid > k.ntiles) grid = k.ntiles;
    k.per_block = ceil_div(k.ntiles, grid);
    grid = ceil_div(k.ntiles, k.per_block);
}

static int ring_setup(const mfc_conv_desc* d, RingK& k, size_t& lds, int& grid, int& MT) {
    if (!d->in || !d->wp || !d->out) return MFC_ERR_INVALID_ARG;
    k.in = (const char*)d->in; k.wp = (const char*)d->wp; k.out = (char*)d->out;
    k.in_coef = d->in_coef; k.out_stats = d->out_stats;
    k.acc_src = (const char*)d->acc_src; k.bn_y = (const char*)d->bn_y; k.bn_coef = d->bn_coef; k.bn_bits = (const unsigned char*)d->bn_bits;
    k.N = d->N; k.H = d->Hout; k.W = d->Wout; k.C = d->Cin;
    k.in_relu = d->in_relu; k.ipg = d->images_per_group; k.G = d->N / d->images_per_group; k.accumulate = d->accumulate; k.bn_mode = d->bn_mask_mode;
    MT = g_ring_mt;
    k.ablate = g_ring_ablate; k.stagger = g_ring_stagger;
    k.wt = mfc_wt_for((double)d->N * d->Hout * d->Wout * d->Cout_p * 2.0);
    if (d->Cin == 32) { if (MT == 4) ring_geo<1, 4>(d, k, lds, grid); else ring_geo<1, 2>(d, k, lds, grid); }
    else { MT = 4; ring_geo<2, 4>(d, k, lds, grid); }
    return MFC_OK;
}

int ring_layout(const mfc_conv_desc* d, mfc_conv_layout* out) {
    RingK k; size_t lds; int grid, MT;
    mfc_conv_desc t = *d;
    if (!t.in) t.in = (const void*)16;
    if (!t.wp) t.wp = (const void*)16;
    if (!t.out) t.out = (void*)16;
    const int rc = ring_setup(&t, k, lds, grid, MT);
    if (rc < 0) return rc;
    const int C = d->Cin;
    out->KG = C / 8; out->nchunks = 1; out->NT16 = C; out->Yblocks = 1; out->nslots = 9 * (C / 8);
    out->TA = 3; out->TB = 3; out->TAS = 3; out->lds_bytes = (int32_t)lds;
    out->bytes = (int64_t)9 * (C / 8) * C * 16;
    out->MT = MT; out->TH = (C == 32 ? 4 : 2) * MT; out->TW = 16; out->grid = grid; out->per_block = k.per_block; out->NW = 4;
    out->fa = (C == 32) ? 1 : 0;            // (the 64-channel fused variants spill: no fusions offered)
    return MFC_OK;
}

template <typename T, int NCH, int MT, bool FUSE>
static int ring_launch_t(const RingK& k, size_t lds, int grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_ring_kernel<T, NCH, MT, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        MFC_PROF_NAME(pname, "conv3x3_ring_kernel<%s, %d, %d, %s>", mfc_tname<T>(), NCH, MT, FUSE ? "true" : "false");
        const double px = (double)k.N * k.H * k.W;
        mfc_prof_before(st, pname, 2.0 * px * 9.0 * k.C * k.C, px * 2.0 * k.C * 2.0);
    }
    hipLaunchKernelGGL((conv3x3_ring_kernel<T, NCH, MT, FUSE>), dim3(grid), dim3(256), lds, st, k);
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

int ring_launch(const mfc_conv_desc* d, hipStream_t st) {
    RingK k; size_t lds; int grid, MT;
    const int rc = ring_setup(d, k, lds, grid, MT);
    if (rc < 0) return rc;
    const bool fused = d->acc_src || d->bn_y || d->accumulate;
    if ((d->flags & MFC_CONV_NEVER_ACC) && (d->accumulate || d->acc_src)) return MFC_ERR_INVALID_ARG;          // (the promise the layout was chosen on)
    if (d->Cin == 64 && fused) return MFC_ERR_UNSUPPORTED;                                                      // (no fused 64-channel form: mfc_conv2d_layout reported fa = 0)
    if (d->acc_src && !d->accumulate) return MFC_ERR_INVALID_ARG;
    if (d->bn_y && (!d->bn_coef || !d->out_stats || (d->bn_mask_mode != 0 && d->bn_mask_mode != 2 && d->bn_mask_mode != 3) ||
                    (d->bn_mask_mode == 3 && !d->bn_bits))) return MFC_ERR_INVALID_ARG;
    int r = MFC_ERR_UNSUPPORTED;
    if (d->Cin == 32) {
        if (MT == 4) {
            if (fused) MFC_TYPED16(d->dtype, T_, r = (ring_launch_t<T_, 1, 4, true>(k, lds, grid, st)));
            else MFC_TYPED16(d->dtype, T_, r = (ring_launch_t<T_, 1, 4, false>(k, lds, grid, st)));
        } else {
            if (fused) MFC_TYPED16(d->dtype, T_, r = (ring_launch_t<T_, 1, 2, true>(k, lds, grid, st)));
            else MFC_TYPED16(d->dtype, T_, r = (ring_launch_t<T_, 1, 2, false>(k, lds, grid, st)));
        }
    } else {
        if (fused) MFC_TYPED16(d->dtype, T_, r = (ring_launch_t<T_, 2, 4, true>(k, lds, grid, st)));
        else MFC_TYPED16(d->dtype, T_, r = (ring_launch_t<T_, 2, 4, false>(k, lds, grid, st)));
    }
    return r;
}
