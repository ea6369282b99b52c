// Weight packer / gradient unpacker (multi-job: one launch covers every convolution of the net).
// Reference layout on both ends is nn.Conv2d.weight [Cout][Cin][KH][KW] fp32 (state_dict of
// models/hrnet.py / models/multiframe_model.py).
#include "common.h"

template <typename J>
__device__ inline int find_job(const J* jobs, int njobs, int bid) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= bid) lo = mid; else hi = mid - 1;
    }
    return lo;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const mfc_pack_job* jobs, int njobs) {
    constexpr int E = Gran<T>::E;
    const mfc_pack_job j = jobs[find_job(jobs, njobs, blockIdx.x)];
    const long total = (long)(j.TA / j.TAS) * j.nchunks * j.Yblocks * j.nslots * j.NT16;
    long idx = (long)(blockIdx.x - j.block0) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int nn = (int)(idx % j.NT16); long r = idx / j.NT16;
    const int slot = (int)(r % j.nslots); r /= j.nslots;
    const int yb = (int)(r % j.Yblocks); r /= j.Yblocks;
    const int c = (int)(r % j.nchunks); const int ag = (int)(r / j.nchunks);
    bool sv = slot < j.TAS * j.TB * j.KG;
    const int al = slot / (j.TB * j.KG); const int rs = slot - al * j.TB * j.KG;
    const int b = rs / j.KG, gi = rs - b * j.KG;
    const int kh = j.kh0 + (ag * j.TAS + al) * j.kh_step, kw = j.kw0 + b * j.kw_step;
    sv = sv && kh >= 0 && kh < j.KH && kw >= 0 && kw < j.KW;      // (a tap outside the filter: zero -- the parity classes of a merged stride-2 data gradient are padded to 2x2 taps)
    const int n = yb * j.NT16 + nn;
    const float* src = (const float*)j.src;
    float f[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = (c * j.KG + gi) * E + e;
        int co, ci;
        if (j.mode == 0) { co = n; ci = k; } else { co = k; ci = n; }
        f[e] = (sv && co < j.Cout && ci < j.Cin) ? src[(((size_t)co * j.Cin + ci) * j.KH + kh) * j.KW + kw] : 0.f;
    }
    ((uint4*)j.dst)[idx] = Gran<T>::pack(f);
}

extern "C" int mfc_pack_weights(const mfc_pack_job* jobs_dev, int32_t njobs, int32_t total_blocks, int32_t dtype, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_blocks <= 0 || !mfc_dtype_ok(dtype)) return MFC_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    MFC_TYPED(dtype, T_, hipLaunchKernelGGL(pack_weights_kernel<T_>, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs));
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// one thread per packed cell (tap, co, ci) [ci fastest: the slice reads are coalesced]: sums the partial-sum slices in a
// fixed order (deterministic) and scatters the total into the reference layout
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const mfc_unpack_job* jobs, int njobs) {
    const mfc_unpack_job j = jobs[find_job(jobs, njobs, blockIdx.x)];
    const int slice = j.KH * j.KW * j.Co16 * j.Ci16;
    const int idx = (blockIdx.x - j.block0) * 256 + threadIdx.x;
    if (idx >= slice) return;
    const int ci = idx % j.Ci16; int r = idx / j.Ci16;
    const int co = r % j.Co16; const int tap = r / j.Co16;
    if (co >= j.Cout || ci >= j.Cin) return;
    const float* src = (const float*)j.src + idx;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    int q = 0;
    for (; q + 8 <= j.nparts; q += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += src[(size_t)(q + k) * slice];
    }
    for (; q < j.nparts; ++q) s[0] += src[(size_t)q * slice];
    const float tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    ((float*)j.dst)[((size_t)co * j.Cin + ci) * (j.KH * j.KW) + tap] = tot;
}

extern "C" int mfc_unpack_wgrad(const mfc_unpack_job* jobs_dev, int32_t njobs, int32_t total_blocks, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_blocks <= 0) return MFC_ERR_INVALID_ARG;
    if (g_mfc_prof_on == 1) mfc_prof_before((hipStream_t)stream, "unpack_wgrad_kernel", 0.0, 0.0);      // (bytes live in the device job table)
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    MFC_PROF_END((hipStream_t)stream);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}
