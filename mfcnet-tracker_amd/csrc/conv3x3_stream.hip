// 3x3 / stride-1 / pad-1 convolutions with 128 or 256 channels on both sides (16-bit storage), forward and data gradient: the
// BasicBlocks of the 30x40 / 15x20 branches (models/hrnet.py:58-74, W32) -- 8.5 GFLOP on 28 800 / 7 200 pixels, MFMA-bound, where
// conv_igemm.hip reaches 240-420 TFLOP/s: its register-staged patch prefetch costs as many cycles per stage as the MFMAs
// (profiles/r02_d_conv_trace_cycles.log: issue 2500 + wait 2200 around a 2500-cycle MFMA loop), all waves in the same phase.
//
// Here NOTHING is staged through registers and the two operands run in LDS-DMA rings of their own:
//   * workgroup = 4 waves, each MT m-tiles of 16 pixels x 64 couts = MT x 4 MFMA tiles (v_mfma_f32_16x16x32: (MT + 4) fragment reads
//     per 4 MT MFMAs, 32 B/clk/SIMD of LDS at MT = 4); a unit = (pixel tile of 64 MT pixels, block of 64 couts); one workgroup per CU;
//     an m-tile is 16 pixels of one row (MSH 0) or 2 rows x 8 columns (MSH 1: 40- and 20-pixel rows divide by 8, not by 16);
//   * K is walked as stages (32-channel plane c of the patch, tap row a): 3 taps x MT x 4 MFMAs per wave from plane c and a
//     [3 taps][4 granules][64 couts] block of packed weights (12 KiB, contiguous in the packed image);
//   * WEIGHT RING of 6 slots, each stage's block DMA-copied (global_load_lds_dwordx4) FIVE stages ahead: 60 KiB of weights in flight
//     per CU -- a CU streams the 147 KiB (128 channels) of its cout block once per unit and an LDS-DMA stream needs that depth
//     (with two stages in flight the first version of this kernel sat at 15 GB/s per CU); PLANE RING of 3 slots, a plane copied
//     two planes = six stages ahead of its first use;
//   * one raw s_barrier per stage (s_waitcnt lgkmcnt only; the rings stay in flight across it) and one counted s_waitcnt vmcnt:
//     "everything but what this wave issued during the last four stages"; stage / plane / unit indices advance incrementally (integer
//     divisions in the stage loop cost 1.5 us per stage in the first version);
//   * plane layout, source-side half swap, in-LDS fix-up (zero padding, producer's BatchNorm + ReLU) and epilogue (statistics in
//     registers, transposed 16-byte stores, data-gradient fusions) as in conv3x3_ring.hip.  The fused epilogue's operands are plain
//     loads issued in front of the unit's last stage: with one workgroup per CU hipcc keeps part of the register file in AGPRs, and an
//     asm load whose destination lives there is copied out before it has landed (guide 5.7, item 1).
// Dispatched from mfc_conv2d_fwd / mfc_conv2d_layout when stream_eligible().
//
// STATUS (round 3): correct (bit-identical to conv_igemm on every variant, tools/bench_ring.py) but NOT faster, so OFF by default
// (mfc_set_flag(34, 1) turns it on): 128 -> 128 at 30x40, N = 24: 24.4 us against conv_igemm's 18.8 us; 256 -> 256 at 15x20: 43 against 29.
// Ablation (tools/ablate_stream.py, profiles/r03_stream_ablation.txt): the MFMAs + the weight stream cost 6.8 us (4 us = the MFMA rate of
// one wave per SIMD), i.e. the rings do their job; the other 18 us are the unit's fixed costs -- first patch + weight round trip,
// 12 barriers with their counted waits (1.7 us), the per-plane issue / border fix-up arithmetic (5.4 us), epilogue -- spread over
// ONE unit per workgroup: a launch with 8.5 GFLOP on 28 800 pixels gives every CU a single 256-pixel x 64-cout tile, and a deeper
// pipeline cannot amortise its prologue over one tile.  What these shapes need is more work per launch (several problems in one
// grid), not a better kernel for one problem.
#include "common.h"

int g_conv_stream = 0;
int g_stream_ablate = 0;

struct StreamK {
    const char* in; const char* wp; char* out;
    const float* in_coef; mfc_stat_t* out_stats;
    const char* acc_src; const char* bn_y; const float* bn_coef; const unsigned char* bn_bits;
    int N, H, W, C;
    int in_relu, ipg, G, accumulate, bn_mode;
    int tilesY, tilesX, npt, Yb, nunits, per_block;      // pixel tiles per launch, cout blocks, units = npt * Yb (cout block fastest)
    int NPL;                                             // 32-channel planes = C / 32
    int off_w, off_coef, off_bnc, off_red;               // LDS byte offsets
    int ablate;                                          // tuning only (mfc_set_flag(35, mask)): 1 skip MFMAs, 2 skip stores, 4 skip weight DMA
};

template <int CTRL> __device__ inline float s_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ inline float s_row16_sum(float v) {
    v = s_dpp_add<0xB1>(v); v = s_dpp_add<0x4E>(v); v = s_dpp_add<0x141>(v); v = s_dpp_add<0x140>(v);
    return v;
}
__device__ inline void s_dma(const char* base, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds) : "memory");
}
__device__ inline void s_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// all but this wave's k youngest vector-memory operations are done (k <= 40; larger values wait for a few more than necessary)
__device__ inline void s_vm_wait(int k) {
#define S_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    switch (k) {
        S_W(0) S_W(1) S_W(2) S_W(3) S_W(4) S_W(5) S_W(6) S_W(7) S_W(8) S_W(9) S_W(10) S_W(11) S_W(12) S_W(13) S_W(14) S_W(15)
        S_W(16) S_W(17) S_W(18) S_W(19) S_W(20) S_W(21) S_W(22) S_W(23) S_W(24) S_W(25) S_W(26) S_W(27) S_W(28) S_W(29) S_W(30)
        S_W(31) S_W(32) S_W(33) S_W(34) S_W(35) S_W(36) S_W(37) S_W(38) S_W(39)
        default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    }
#undef S_W
}

constexpr int S_NPR = 3;                     // plane ring slots
constexpr int S_NWR = 6;                     // weight ring slots
constexpr int S_CB = 64;                     // couts per unit
constexpr int S_STG = 3 * 4 * S_CB * 16;     // bytes of one weight stage: [3 taps][4 granules][64 couts][16 B] = 12 KiB
constexpr int S_NWPW = S_STG / 1024 / 4;     // DMA pieces per wave and weight stage (3)
constexpr int S_WAHEAD = S_NWR - 1;          // stages a weight block is issued ahead of its use
static_assert(S_STG % 4096 == 0, "every wave issues the same number of weight pieces");

template <int MSH, int MT> struct StreamGeo {
    static constexpr int TWc = MSH ? 8 : 16;                     // tile columns
    static constexpr int MR = MSH ? 2 : 1;                       // rows of an m-tile
    static constexpr int TH = 4 * MT * MR;                       // tile rows: 4 waves x MT m-tiles
    static constexpr int PW = TWc + 2, PH = TH + 2;
    static constexpr int NPX = PH * PW;
    static constexpr int PLANE = ((NPX * 64 + 1023) / 1024) * 1024;
    static constexpr int PPP = PLANE / 1024;                     // DMA pieces per plane
    static constexpr int NPW = (PPP + 3) / 4;                    // ... per wave
    static constexpr int OFF_W = S_NPR * PLANE;
};

template <typename T, int MSH, int MT>
__global__ __launch_bounds__(256, 2) void conv3x3_stream_kernel(StreamK p) {      // (2: a 256-register budget with NO AGPR half -- with 512 hipcc keeps the accumulators in AGPRs and copies them to and fro: 113 v_accvgpr moves per stage)
    typedef StreamGeo<MSH, MT> Geo;
    constexpr int TWc = Geo::TWc, MR = Geo::MR, TH = Geo::TH, PW = Geo::PW, NPX = Geo::NPX, PLANE = Geo::PLANE, PPP = Geo::PPP, NPW = Geo::NPW;
    constexpr int NT = 4;
    constexpr int ROWB = PW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* coefl = (float*)(smem + p.off_coef);                  // [G][2][C]
    float* bncl = (float*)(smem + p.off_bnc);                    // [G][4][C]
    float* red = (float*)(smem + p.off_red);                     // [2 parities][4 waves][2][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gl = lane >> 4, lx = lane & 15;
    const int ry = MSH ? (lx >> 3) : 0, cx = MSH ? (lx & 7) : lx;      // the lane's pixel inside an m-tile
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int u0 = Lb * p.per_block;
    const int nun = min(p.per_block, p.nunits - u0);
    if (nun <= 0) return;
    const int H = p.H, W = p.W, C = p.C, NPL = p.NPL, Yb = p.Yb;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int NPtot = nun * NPL, NG = 3 * NPtot;

    // ---------------- unit cursors (cout block fastest), advanced incrementally: no divisions in the stage loop ----------------
    struct UC { int n, ty, tx, yb; };
    auto uc_next = [&](UC c) {
        if (++c.yb == Yb) { c.yb = 0; if (++c.tx == p.tilesX) { c.tx = 0; if (++c.ty == p.tilesY) { c.ty = 0; ++c.n; } } }
        return c;
    };
    UC ucur;
    {
        const int pt = u0 / Yb; ucur.yb = u0 - pt * Yb;
        const int tpi = p.tilesY * p.tilesX;
        ucur.n = pt / tpi; const int q = pt - ucur.n * tpi;
        ucur.ty = q / p.tilesX; ucur.tx = q - ucur.ty * p.tilesX;
    }
    auto interior = [&](const UC& c) {
        const int i0 = c.ty * TH, j0 = c.tx * TWc;
        return i0 >= 1 && j0 >= 1 && i0 + TH + 1 <= H && j0 + TWc + 1 <= W;
    };
    // source-side slot permutation of a patch pixel (conflict-free ds_read_b128 fragments): swap the 32-byte halves where bit 2 (rows
    // of 16 pixels) / bit 1 (2 x 8 m-tiles) of the patch column is set
    auto swz = [&](int px) { return MSH ? (((px >> 1) & 1) << 1) : (((px >> 2) & 1) << 1); };

    // ---------------- plane DMA ----------------
    const int npp = (PPP - wave + 3) / 4;                        // pieces of a plane this wave issues
    auto piece_src = [&](int i, int ln, int& py, int& px, int& g) {
        const int S = (wave + 4 * i) * 64 + ln;
        const int q = min(S >> 2, NPX - 1), sp = S & 3;
        py = q / PW; px = q - py * PW;
        g = sp ^ swz(px);
    };
    auto issue_plane = [&](const UC& uc, int c, int slot) {
        const int i0 = uc.ty * TH, j0 = uc.tx * TWc;
        const char* img = p.in + ((size_t)uc.n * H * W * C + c * 32) * 2;
        const unsigned lbase = lds0 + slot * PLANE + wave * 1024;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const bool inter = interior(uc);
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            if (k < npp) {
                int py, px, g;
                piece_src(k, ln, py, px, g);
                int iy = i0 - 1 + py, ix = j0 - 1 + px;
                if (!inter) { iy = min(max(iy, 0), H - 1); ix = min(max(ix, 0), W - 1); }
                s_dma(img, (unsigned)(((iy * W + ix) * C + g * 8) * 2), lbase + k * 4096);
            }
        }
        return npp;
    };
    const bool xf = (p.in_coef != nullptr);
    auto fixup_plane = [&](const UC& uc, int c, int slot) {      // this wave's own pieces, after they have landed
        const bool inter = interior(uc);
        if ((!xf && inter) || (p.ablate & 8)) return;
        const int i0 = uc.ty * TH, j0 = uc.tx * TWc;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const float* cfg = coefl + (uc.n / p.ipg) * 2 * C + c * 32;
        char* lb = smem + slot * PLANE + wave * 1024 + ln * 16;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            if (k < npp) {
                int py, px, g;
                piece_src(k, ln, py, px, g);
                const bool inr = inter || ((unsigned)(i0 - 1 + py) < (unsigned)H && (unsigned)(j0 - 1 + px) < (unsigned)W);
                char* a = lb + k * 4096;
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
    // ---------------- weight-stage DMA: stage (plane c, tap row a, cout block yb) = one contiguous 12-KiB block of the packed image
    //                  [tap row][chunk][cout block][3 taps x 4 granules][64 couts][16 B] (mfc_conv2d_layout: TAS = 1, nslots = 12) ----------------
    auto issue_w = [&](int a, int c, int yb, int slot) {
        if (p.ablate & 4) return 0;
        const char* src = p.wp + (size_t)((a * NPL + c) * Yb + yb) * S_STG;
        const unsigned lbase = lds0 + p.off_w + slot * S_STG + wave * 1024;
#pragma unroll
        for (int k = 0; k < S_NWPW; ++k)
            s_dma(src, (unsigned)((wave + 4 * k) * 1024 + lane * 16), lbase + k * 4096);
        return S_NWPW;
    };

    // ---------------- coefficient tables -> LDS ----------------
    if (xf) {
        for (int i = tid; i < p.G * 2 * C; i += 256) {
            const int g = i / (2 * C), r = i - g * 2 * C;
            coefl[i] = p.in_coef[(size_t)g * 4 * C + r];
        }
    }
    const bool bnm = p.bn_y != nullptr;
    if (bnm) for (int i = tid; i < p.G * 4 * C; i += 256) bncl[i] = p.bn_coef[i];

    // ---------------- accumulators / statistics ----------------
    f32x4 acc[MT][NT];
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }
    }
    int red_par = 0; bool red_pending = false; int red_grp = 0, red_rep = 0, red_yb = 0;
    auto stats_to_lds = [&](int grp, int rep, int yb) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sa = s_row16_sum(ssum[nt][r]), sb = s_row16_sum(ssq[nt][r]);
                ssum[nt][r] = 0.f; ssq[nt][r] = 0.f;
                if (lx == 0) {
                    // (BatchNorm-backward mode keeps the sums in the TRANSPOSED layout of the epilogue: pair nt >> 1, channel 8 gl + 4 (nt & 1) + r)
                    const int cl = bnm ? ((nt >> 1) * 32 + gl * 8 + (nt & 1) * 4 + r) : (nt * 16 + gl * 4 + r);
                    red[red_par * 512 + (wave * 2 + 0) * 64 + cl] = sa;
                    red[red_par * 512 + (wave * 2 + 1) * 64 + cl] = sb;
                }
            }
        red_pending = true; red_grp = grp; red_rep = rep; red_yb = yb; red_par ^= 1;
    };
    auto stats_flush = [&]() {
        if (tid < 2 * S_CB) {
            const int which = tid / S_CB, c = tid - which * S_CB;
            const float* rd = red + (red_par ^ 1) * 512;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += rd[(w * 2 + which) * 64 + c];
            atomicAdd(p.out_stats + (((size_t)red_rep * p.G + red_grp) * 2 + which) * C + red_yb * S_CB + c, (mfc_stat_t)s);
        }
        red_pending = false;
    };

    // ---------------- fragment addressing ----------------
    int adx[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int px = cx + dx;
        adx[dx] = (wave * MT * MR + ry) * ROWB + px * 64 + ((gl ^ swz(px)) * 16);
    }
    const int abase = p.off_w + (gl * S_CB + lx) * 16;           // A fragment: [granule gl][cout nt*16 + lx]
    const int e_row = W * C * 2;
    const int e_lane = (((wave * MT * MR + ry) * W + cx) * C) * 2 + gl * 16;      // the lane's 8 output channels of m-tile 0, from the unit's origin

    // ---------------- issue cursors: planes run 2 ahead of the compute cursor, weight stages S_WAHEAD ahead ----------------
    UC puc = ucur; int pc = 0, pP = 0;                            // next plane to issue: (unit, chunk), running plane number
    UC wuc = ucur; int wc = 0, wa = 0, wg = 0;                    // next weight stage to issue
    auto plane_issue_next = [&]() {
        const int n = issue_plane(puc, pc, pP % S_NPR);
        ++pP; if (++pc == NPL) { pc = 0; puc = uc_next(puc); }
        return n;
    };
    auto w_issue_next = [&]() {
        const int n = issue_w(wa, wc, wuc.yb, wg % S_NWR);
        ++wg; if (++wa == 3) { wa = 0; if (++wc == NPL) { wc = 0; wuc = uc_next(wuc); } }
        return n;
    };

    // ---------------- prologue: plane 0, W(0), then what else runs ahead; wait for plane 0 + W(0) only ----------------
    // hist[] = vector-memory operations this wave issued during the last four stages; the prologue's issues take the places of the
    // stages -4 .. -1 in which the loop would have issued them (plane 1 in front of W(1))
    int hist[4] = {0, 0, 0, 0};
    plane_issue_next();
    w_issue_next();
    if (NPtot > 1) hist[0] += plane_issue_next();
    for (int k = 1; k < S_WAHEAD && k < NG; ++k) hist[k - 1] += w_issue_next();
    s_barrier_lds();                                             // coefficient tables visible
    s_vm_wait(hist[0] + hist[1] + hist[2] + hist[3]);            // plane 0 and W(0) have landed
    fixup_plane(ucur, 0, 0);
    s_barrier_lds();

    UC fuc = ucur; int fc = 0;                                    // (unit, chunk) of compute plane P + 1: the plane to fix up before stage 3 (P + 1)
    if (++fc == NPL) { fc = 0; fuc = uc_next(fuc); }
    int P = 0, a = 0, c = 0, ui = 0;
    for (int g = 0; g < NG; ++g) {
        const bool last_of_unit = (c == NPL - 1) && (a == 2);
        const int i0 = ucur.ty * TH, j0 = ucur.tx * TWc;
        const size_t toff = (((size_t)ucur.n * H + i0) * W + j0) * C * 2 + (size_t)(ucur.yb * S_CB) * 2;
        const bool full = (i0 + TH <= H) && (j0 + TWc <= W);
        const bool vx = (full || (j0 + cx < W)) && !(p.ablate & 2);
        const bool fuse = p.accumulate || bnm;
        // ---- data-gradient fusions: the epilogue walks its 2 MT (m-tile, cout pair) steps with the global reads of step s + 1 (running sum,
        //      pre-BN tensor, mask bits) in flight while step s is processed; step 0's reads go out here, in front of the unit's last stage.
        //      (All steps up front would be 72 more live registers: beyond 256, hipcc parks the accumulators in AGPRs and pays 440
        //      v_accvgpr moves per stage -- 1.5 us per stage, measured.)
        uint4 pf_old[2], pf_y[2]; unsigned pf_bits[2];
        const char* ab = (p.acc_src ? p.acc_src : (const char*)p.out) + toff;
        const char* yb_ = p.bn_y + toff;
        const unsigned char* bb = p.bn_bits + (toff >> 4);
        auto epi_prefetch = [&](int st) {
            const int mt = st >> 1, pr = st & 1;
            const bool vpx = vx && (full || (i0 + (wave * MT + mt) * MR + ry < H));
            const unsigned lo = vpx ? (unsigned)(e_lane + mt * MR * e_row + pr * 64) : 0u;
            pf_old[st & 1] = make_uint4(0, 0, 0, 0); pf_y[st & 1] = make_uint4(0, 0, 0, 0); pf_bits[st & 1] = 0xffu;
            if (p.accumulate) pf_old[st & 1] = *(const uint4*)(ab + lo);
            if (bnm) pf_y[st & 1] = *(const uint4*)(yb_ + lo);
            if (bnm && p.bn_mode == 3) pf_bits[st & 1] = bb[lo >> 4];
        };
        if (last_of_unit && fuse) {
            epi_prefetch(0);
            __builtin_amdgcn_sched_barrier(0);                   // (the loads stay in front of the MFMAs that hide them)
        }
        // ---- ring refills (every wave has passed the barrier behind stage g - 1: its weight slot and, at a plane change, plane P - 1 are free).
        //      The plane goes first: everything issued before a weight block is older than it.
        int issued = 0;
        if (a == 0 && pP < NPtot && pP <= P + 2 && !(p.ablate & 32)) issued += plane_issue_next();
        if (wg < NG) issued += w_issue_next();
        hist[0] = hist[1]; hist[1] = hist[2]; hist[2] = hist[3]; hist[3] = issued;

        // ---------------- MFMAs of stage g: 3 taps x (MT x 4) ----------------
        if (!(p.ablate & 1)) {
            const char* pl = smem + (P % S_NPR) * PLANE + a * ROWB;
            const char* wl = smem + abase + (g % S_NWR) * S_STG;
            bf16x8 xa[MT], wfa[NT], xb[MT], wfb[NT];
            auto ldx = [&](bf16x8* x, int b) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) x[mt] = *(const bf16x8*)(pl + adx[b] + mt * MR * ROWB);
            };
            auto ldw = [&](bf16x8* w, int b) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w[nt] = *(const bf16x8*)(wl + b * (4 * S_CB * 16) + nt * 256);
            };
            auto mm = [&](const bf16x8* w, const bf16x8* x) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma16<T>(w[nt], x[mt], acc[mt][nt]);
            };
            ldx(xa, 0); ldw(wfa, 0);
            ldx(xb, 1); ldw(wfb, 1);
            mm(wfa, xa);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MT + NT), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            ldx(xa, 2); ldw(wfa, 2);
            mm(wfb, xb);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            mm(wfa, xa);
        }

        // ---------------- unit epilogue ----------------
        if (last_of_unit) {
            char* tbase = p.out + toff;
            const int n = ucur.n;
            const float* cf0 = bncl + (n / p.ipg) * 4 * C + ucur.yb * S_CB + gl * 8;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool vpx = vx && (full || (i0 + (wave * MT + mt) * MR + ry < H));
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int st = mt * 2 + pr;
                    if (fuse) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (st + 1 < 2 * MT) epi_prefetch(st + 1);
                    }
                    float v[2][4];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[q][r] = acc[mt][2 * pr + q][r]; acc[mt][2 * pr + q][r] = 0.f; }
                    const unsigned off = (unsigned)(e_lane + mt * MR * e_row + pr * 64);
                    if (fuse) {
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
                            for (int q = 0; q < 2; ++q)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (vpx) { ssum[2 * pr + q][r] += v[q][r]; ssq[2 * pr + q][r] += v[q][r] * v[q][r]; }
                        }
                        float o8[8];
                        Gran<T>::unpack(pf_old[st & 1], o8);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { w[r] = __uint_as_float(t0[r]) + o8[r]; w[4 + r] = __uint_as_float(t1[r]) + o8[4 + r]; }
                        if (bnm) {
                            const float* cf = cf0 + pr * 32;
                            float yv[8];
                            Gran<T>::unpack(pf_y[st & 1], yv);
                            unsigned mk = 0xffu;
                            if (p.bn_mode == 2) {
                                mk = 0;
#pragma unroll
                                for (int e = 0; e < 8; ++e) mk |= ((yv[e] * cf[e] + cf[C + e]) > 0.f ? 1u : 0u) << e;
                            } else if (p.bn_mode == 3) {
                                mk = pf_bits[st & 1];
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float gmv = ((mk >> e) & 1u) ? w[e] : 0.f;
                                w[e] = gmv;
                                if (vpx) {
                                    ssum[2 * pr + (e >> 2)][e & 3] += gmv;
                                    ssq[2 * pr + (e >> 2)][e & 3] += gmv * ((yv[e] - cf[2 * C + e]) * cf[3 * C + e]);
                                }
                            }
                        }
                        if (vpx) *(uint4*)(tbase + off) = Gran<T>::pack(w);
                    } else {
                        const unsigned p0 = pack2<T>(v[0][0], v[0][1]), p1 = pack2<T>(v[0][2], v[0][3]);
                        const unsigned q0 = pack2<T>(v[1][0], v[1][1]), q1 = pack2<T>(v[1][2], v[1][3]);
                        auto a32 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                        auto a16 = __builtin_amdgcn_permlane16_swap(a32[0], a32[1], false, false);
                        auto b32 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                        auto b16 = __builtin_amdgcn_permlane16_swap(b32[0], b32[1], false, false);
                        if (p.out_stats) {
#pragma unroll
                            for (int q = 0; q < 2; ++q)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (vpx) { ssum[2 * pr + q][r] += v[q][r]; ssq[2 * pr + q][r] += v[q][r] * v[q][r]; }
                        }
                        if (vpx) *(uint4*)(tbase + off) = make_uint4(a16[0], b16[0], a16[1], b16[1]);
                    }
                }
            }
            if (p.out_stats) {
                bool flush = (ui + 1 >= nun);
                if (!flush) { const UC un = uc_next(ucur); flush = (un.yb != ucur.yb) || (un.n / p.ipg != n / p.ipg); }
                if (flush) stats_to_lds(n / p.ipg, (Lb + ui) % MFC_R, ucur.yb);
            }
        }

        // ---------------- hand over to stage g + 1 ----------------
        if (g + 1 < NG) {
            // this wave's pieces of weight stage g + 1 have landed: it was issued S_WAHEAD - 1 = four stages ago, in front of everything
            // issued since -- and so has every plane issued longer ago than that
            if (!(p.ablate & 16)) s_vm_wait(hist[0] + hist[1] + hist[2] + hist[3]);
            if (a == 2) fixup_plane(fuc, fc, (P + 1) % S_NPR);     // the next stage opens plane P + 1 (issued six stages ago)
            if (!(p.ablate & 64)) s_barrier_lds();
            if (red_pending) stats_flush();
            if (++a == 3) {
                a = 0; ++P;
                if (++fc == NPL) { fc = 0; fuc = uc_next(fuc); }
                if (++c == NPL) { c = 0; ucur = uc_next(ucur); ++ui; }
            }
        }
    }
    if (red_pending) { s_barrier_lds(); stats_flush(); }
}

// ------------------------------------------------------------------------------------------
static bool stream_fits(const mfc_conv_desc* d);
bool stream_eligible(const mfc_conv_desc* d) {
    if (d && d->in_fin) return false;                            // (the folded finalize is built into conv_igemm / conv3x3_ring only)
    if (!g_conv_stream || !d || !mfc_is16(d->dtype)) return false;
    if (d->TA != 3 || d->TB != 3 || d->dh0 != -1 || d->dw0 != -1 || d->in_stride != 1) return false;
    if (d->out_sh != 1 || d->out_sw != 1 || d->out_oh != 0 || d->out_ow != 0) return false;
    if (d->Hin != d->Hout || d->Win != d->Wout || d->Hl != d->Hout || d->Wl != d->Wout) return false;
    if (d->Cin != d->Cout || d->Cin_p != d->Cin || d->Cout_p != d->Cout || (d->Cin != 128 && d->Cin != 256)) return false;
    if (d->bias || d->TH > 0 || d->TW > 0) return false;
    if (d->N <= 0 || d->images_per_group <= 0 || d->N % d->images_per_group || d->N / d->images_per_group > 8) return false;
    if (d->Hin < 2 || d->Win < 2) return false;
    if ((double)d->Hin * d->Win * d->Cin * 2.0 >= 2.0e9) return false;
    return stream_fits(d);
}

static int g_stream_form = 0;        // tile form (tuning: mfc_set_flag(36, 10 * MSH + MT with MT in {2, 4}, e.g. 4, 14, 12); 0 = choose)
int mfc_stream_set_mt(int v) { g_stream_form = v; return 0; }

template <int MSH, int MT>
static void stream_geo(const mfc_conv_desc* d, StreamK& k, size_t& lds, int& grid) {
    typedef StreamGeo<MSH, MT> Geo;
    k.tilesY = ceil_div(d->Hout, Geo::TH); k.tilesX = ceil_div(d->Wout, Geo::TWc);
    k.npt = d->N * k.tilesY * k.tilesX;
    k.Yb = d->Cout / S_CB;
    k.nunits = k.npt * k.Yb;
    k.NPL = d->Cin / 32;
    const int C = d->Cin, G = d->N / d->images_per_group;
    k.off_w = Geo::OFF_W;
    k.off_coef = k.off_w + S_NWR * S_STG;
    k.off_bnc = k.off_coef + G * 2 * C * 4;
    k.off_red = k.off_bnc + G * 4 * C * 4;
    lds = (size_t)k.off_red + 2 * 512 * 4;
    grid = 256;
    if (grid > k.nunits) grid = k.nunits;
    k.per_block = ceil_div(k.nunits, grid);
    grid = ceil_div(k.nunits, k.per_block);
}

static const int S_FORMS[4][2] = {{0, 4}, {1, 4}, {0, 2}, {1, 2}};      // (MSH, MT): tiles of 16x16, 32x8, 8x16, 16x8 pixels
static void stream_geo_form(int f, const mfc_conv_desc* d, StreamK& k, size_t& lds, int& grid) {
    switch (f) {
        case 0: stream_geo<0, 4>(d, k, lds, grid); break;
        case 1: stream_geo<1, 4>(d, k, lds, grid); break;
        case 2: stream_geo<0, 2>(d, k, lds, grid); break;
        default: stream_geo<1, 2>(d, k, lds, grid); break;
    }
}

static int stream_setup(const mfc_conv_desc* d, StreamK& k, size_t& lds, int& grid, int& form) {
    if (!d->in || !d->wp || !d->out) return MFC_ERR_INVALID_ARG;
    k.in = (const char*)d->in; k.wp = (const char*)d->wp; k.out = (char*)d->out;
    k.in_coef = d->in_coef; k.out_stats = d->out_stats;
    k.acc_src = (const char*)d->acc_src; k.bn_y = (const char*)d->bn_y; k.bn_coef = d->bn_coef; k.bn_bits = (const unsigned char*)d->bn_bits;
    k.N = d->N; k.H = d->Hout; k.W = d->Wout; k.C = d->Cin;
    k.in_relu = d->in_relu; k.ipg = d->images_per_group; k.G = d->N / d->images_per_group; k.accumulate = d->accumulate; k.bn_mode = d->bn_mask_mode;
    k.ablate = g_stream_ablate;
    // tile form: fewest (rounds of 256 workgroups) x (pixels per tile), the 64-pixel-per-wave forms (MT = 4: half the LDS traffic per MFMA) preferred
    form = -1;
    if (g_stream_form) {
        for (int f = 0; f < 4; ++f) if (S_FORMS[f][0] * 10 + S_FORMS[f][1] == g_stream_form) form = f;
    }
    if (form < 0) {
        double best = 1e30;
        for (int f = 0; f < 4; ++f) {
            StreamK t = k; size_t l; int g;
            stream_geo_form(f, d, t, l, g);
            if (l > (size_t)160 * 1024) continue;
            const double cost = (double)t.per_block * S_FORMS[f][1] * (S_FORMS[f][1] == 2 ? 1.25 : 1.0);
            if (cost < best - 1e-9) { best = cost; form = f; }
        }
        if (form < 0) return MFC_ERR_UNSUPPORTED;
    }
    stream_geo_form(form, d, k, lds, grid);
    if (lds > (size_t)160 * 1024) return MFC_ERR_UNSUPPORTED;
    return MFC_OK;
}

static bool stream_fits(const mfc_conv_desc* d) {          // the coefficient tables of G statistic groups next to the rings: within 160 KiB?
    StreamK k; size_t lds; int grid, form;
    mfc_conv_desc t = *d;
    t.in = (const void*)16; t.wp = (const void*)16; t.out = (void*)16;
    return stream_setup(&t, k, lds, grid, form) == MFC_OK;
}

int stream_layout(const mfc_conv_desc* d, mfc_conv_layout* out) {
    StreamK k; size_t lds; int grid, form;
    mfc_conv_desc t = *d;
    if (!t.in) t.in = (const void*)16;
    if (!t.wp) t.wp = (const void*)16;
    if (!t.out) t.out = (void*)16;
    const int rc = stream_setup(&t, k, lds, grid, form);
    if (rc < 0) return rc;
    const int C = d->Cin, MSH = S_FORMS[form][0], MT = S_FORMS[form][1];
    out->KG = 4; out->nchunks = C / 32; out->NT16 = S_CB; out->Yblocks = C / S_CB; out->nslots = 12;
    out->TA = 3; out->TB = 3; out->TAS = 1; out->lds_bytes = (int32_t)lds;
    out->bytes = (int64_t)9 * (C / 8) * C * 16;
    out->MT = MT; out->TH = 4 * MT * (MSH ? 2 : 1); out->TW = MSH ? 8 : 16; out->grid = grid; out->per_block = k.per_block; out->NW = 4;
    out->fa = 1;
    return MFC_OK;
}

template <typename T, int MSH, int MT>
static int stream_launch_t(const StreamK& k, size_t lds, int grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_stream_kernel<T, MSH, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (g_mfc_prof_on == 1) {
        MFC_PROF_NAME(pname, "conv3x3_stream_kernel<%s, %d, %d>", mfc_tname<T>(), MSH, MT);
        const double px = (double)k.N * k.H * k.W;
        mfc_prof_before(st, pname, 2.0 * px * 9.0 * k.C * k.C, px * 2.0 * k.C * 2.0);
    }
    hipLaunchKernelGGL((conv3x3_stream_kernel<T, MSH, MT>), dim3(grid), dim3(256), lds, st, k);
    if (g_mfc_prof_on == 1) mfc_prof_after(st);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

int stream_launch(const mfc_conv_desc* d, hipStream_t st) {
    StreamK k; size_t lds; int grid, form;
    const int rc = stream_setup(d, k, lds, grid, form);
    if (rc < 0) return rc;
    if (d->acc_src && !d->accumulate) return MFC_ERR_INVALID_ARG;
    if (d->bn_y && (!d->bn_coef || !d->out_stats || (d->bn_mask_mode != 0 && d->bn_mask_mode != 2 && d->bn_mask_mode != 3) ||
                    (d->bn_mask_mode == 3 && !d->bn_bits))) return MFC_ERR_INVALID_ARG;
    int r = MFC_ERR_UNSUPPORTED;
    switch (form) {
        case 0: MFC_TYPED16(d->dtype, T_, r = (stream_launch_t<T_, 0, 4>(k, lds, grid, st))); break;
        case 1: MFC_TYPED16(d->dtype, T_, r = (stream_launch_t<T_, 1, 4>(k, lds, grid, st))); break;
        case 2: MFC_TYPED16(d->dtype, T_, r = (stream_launch_t<T_, 0, 2>(k, lds, grid, st))); break;
        default: MFC_TYPED16(d->dtype, T_, r = (stream_launch_t<T_, 1, 2>(k, lds, grid, st))); break;
    }
    return r;
}
