// Loss, Adam and the program interpreter of libmfcnet_hip (gfx950).
//   mfc_loss_fwd/bwd   F.log_softmax + class-weighted NLL + soft-Jaccard (src/engine.py:65-66, src/loss.py:31-63)
//   mfc_adam_step      torch.optim.Adam over flat fp32 arenas (scripts/train_multiframe_detection.py:128-151)
//   mfc_program_run    one host call enqueues a whole forward or backward pass
#include "common.h"
#include <math.h>

// acc layout: [0] sum w_t*(-logp_t)  [1] sum w_t  [2+c] I_c  [10+c] P_c = sum p_c  [18+c] T_c = sum [t==c]
//             [26] nll  [27] soft-jaccard  [28] total  [29] number of targets outside [0, nc) (they are ignored; nn.NLLLoss raises on them)
// The 26 sums are order-independent: lanes -> wave (shuffles) -> workgroup (fixed order over the 4 waves) in fp32, workgroups -> total
// with fp64 atomics into the scratch half of the acc block; loss_sums_kernel then rounds the totals to acc[0..25].
__global__ __launch_bounds__(256) void loss_fwd_kernel(mfc_loss_desc d, long total) {
    __shared__ float wred[4][26];
    __shared__ unsigned bad_s;
    if (threadIdx.x == 0) bad_s = 0u;
    __syncthreads();
    const long HW = (long)d.H * d.W;
    unsigned bad = 0;
    float a0 = 0.f, a1 = 0.f, I[8], P[8], Tc[8];
    for (int c = 0; c < 8; ++c) { I[c] = 0.f; P[c] = 0.f; Tc[c] = 0.f; }
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long b = idx / HW, r = idx - b * HW;
        float x[8], mx = -3.4e38f;
        for (int c = 0; c < d.nc; ++c) { x[c] = d.logits[(b * d.nc + c) * HW + r]; mx = fmaxf(mx, x[c]); }
        float se = 0.f;
        for (int c = 0; c < d.nc; ++c) se += expf(x[c] - mx);
        const float lse = mx + logf(se);
        const long tl = (long)d.target[idx];
        const bool tok = tl >= 0 && tl < (long)d.nc;        // a label outside the class range (255 "ignore", a corrupt mask) must not index class_w / x
        const int t = tok ? (int)tl : -1;
        bad += tok ? 0u : 1u;
        if (tok) {
            float xt = x[0];
#pragma unroll
            for (int c = 1; c < 8; ++c) xt = (c == t) ? x[c] : xt;
            const float wt = d.class_w ? d.class_w[t] : 1.f;
            a0 += wt * (lse - xt); a1 += wt;
        }
        for (int c = 1; c < d.nc; ++c) {
            const float pc = expf(x[c] - lse);
            P[c] += pc;
            if (t == c) { I[c] += pc; Tc[c] += 1.f; }
        }
    }
    // wave reduce then LDS atomics
    for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off);
        for (int c = 1; c < d.nc; ++c) { I[c] += __shfl_down(I[c], off); P[c] += __shfl_down(P[c], off); Tc[c] += __shfl_down(Tc[c], off); }
    }
    if ((threadIdx.x & 63) == 0) {
        float* r = wred[threadIdx.x >> 6];
        for (int i = 0; i < 26; ++i) r[i] = 0.f;
        r[0] = a0; r[1] = a1;
        for (int c = 1; c < d.nc; ++c) { r[2 + c] = I[c]; r[10 + c] = P[c]; r[18 + c] = Tc[c]; }
    }
    if (bad) atomicAdd(&bad_s, bad);
    __syncthreads();
    if (threadIdx.x < 26) {
        const float v = ((wred[0][threadIdx.x] + wred[1][threadIdx.x]) + wred[2][threadIdx.x]) + wred[3][threadIdx.x];
        if (v != 0.f) atomicAdd((double*)(d.acc + 32) + threadIdx.x, (double)v);
    }
    if (threadIdx.x == 0 && bad_s) atomicAdd(d.acc + 29, (float)bad_s);       // (a count: exact in fp32 whatever the order)
}
__global__ void loss_sums_kernel(mfc_loss_desc d) {
    if (threadIdx.x < 26) d.acc[threadIdx.x] = (float)((const double*)(d.acc + 32))[threadIdx.x];
}

__global__ void loss_finalize_kernel(mfc_loss_desc d) {
    if (threadIdx.x != 0) return;
    const float eps = 1e-15f;
    const float nll = d.acc[0] / d.acc[1];
    float jac = 0.f;
    for (int c = 1; c < d.nc; ++c) {
        const float I = d.acc[2 + c], U = d.acc[10 + c] + d.acc[18 + c] - I;
        jac += -logf((I + eps) / (U + eps));
    }
    jac /= (float)d.nc;
    d.acc[26] = nll; d.acc[27] = jac; d.acc[28] = d.w_nll * nll + d.w_jac * jac;
}

static int loss_check(const mfc_loss_desc* d) {
    if (!d || !d->logits || !d->target || !d->acc || d->nc < 2 || d->nc > 8 || d->B <= 0) return MFC_ERR_INVALID_ARG;
    return MFC_OK;
}

extern "C" int mfc_loss_partial(const mfc_loss_desc* d, void* stream) {
    int rc = loss_check(d); if (rc < 0) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (((uintptr_t)d->acc) & 7) return MFC_ERR_INVALID_ARG;
    if (hipMemsetAsync(d->acc, 0, MFC_LOSS_ACC_FLOATS * sizeof(float), st) != hipSuccess) return MFC_ERR_LAUNCH;
    const long total = (long)d->B * d->H * d->W;
    long blocks = (total + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(loss_fwd_kernel, dim3((int)blocks), dim3(256), 0, st, *d, total);
    hipLaunchKernelGGL(loss_sums_kernel, dim3(1), dim3(64), 0, st, *d);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

extern "C" int mfc_loss_finalize(const mfc_loss_desc* d, void* stream) {
    if (!d || !d->acc || d->nc < 2 || d->nc > 8) return MFC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *d);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

extern "C" int mfc_loss_fwd(const mfc_loss_desc* d, void* stream) {
    const int rc = mfc_loss_partial(d, stream);
    return rc < 0 ? rc : mfc_loss_finalize(d, stream);
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(mfc_loss_desc d, long total) {
    const float eps = 1e-15f;
    const long HW = (long)d.H * d.W;
    const float Wsum = d.acc[1];
    const float gsc = d.grad_scale * (d.grad_scale_dev ? d.grad_scale_dev[0] : 1.f);
    float aI[8], aU[8];
    for (int c = 0; c < 8; ++c) { aI[c] = 0.f; aU[c] = 0.f; }
    for (int c = 1; c < d.nc; ++c) {
        const float I = d.acc[2 + c], U = d.acc[10 + c] + d.acc[18 + c] - I;
        aI[c] = -1.f / ((I + eps) * (float)d.nc);     // d/dp_c where t == c  (dU/dp_c = 0 there)
        aU[c] = 1.f / ((U + eps) * (float)d.nc);      // d/dp_c where t != c
    }
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long b = idx / HW, r = idx - b * HW;
        float x[8], mx = -3.4e38f;
        for (int c = 0; c < d.nc; ++c) { x[c] = d.logits[(b * d.nc + c) * HW + r]; mx = fmaxf(mx, x[c]); }
        float se = 0.f;
        for (int c = 0; c < d.nc; ++c) se += expf(x[c] - mx);
        const float lse = mx + logf(se);
        const long tl = (long)d.target[idx];
        const bool tok = tl >= 0 && tl < (long)d.nc;
        const int t = tok ? (int)tl : -1;                    // out-of-range target: no NLL term, "no class" for the Jaccard term (as in the forward)
        const float wt = tok ? (d.class_w ? d.class_w[t] : 1.f) / Wsum : 0.f;
        float p[8], a[8], dot = 0.f;
        for (int c = 0; c < d.nc; ++c) {
            p[c] = expf(x[c] - lse);
            a[c] = c == 0 ? 0.f : (t == c ? aI[c] : aU[c]);
            dot += a[c] * p[c];
        }
        for (int c = 0; c < d.nc; ++c) {
            const float gn = wt * (p[c] - (c == t ? 1.f : 0.f));
            const float gj = p[c] * (a[c] - dot);
            d.dlogits[(b * d.nc + c) * HW + r] = gsc * (d.w_nll * gn + d.w_jac * gj);
        }
    }
}

extern "C" int mfc_loss_bwd(const mfc_loss_desc* d, void* stream) {
    int rc = loss_check(d); if (rc < 0) return rc;
    if (!d->dlogits) return MFC_ERR_INVALID_ARG;
    const long total = (long)d->B * d->H * d->W;
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, *d, total);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// ------------------------------------------------------------------ validation metrics
// grid (blocks per sample, B): LDS histogram of (truth, argmax) pairs, then one 64-bit atomic per non-empty bin
__global__ __launch_bounds__(256) void confusion_kernel(const float* out, const int64_t* target, unsigned long long* conf,
                                                        int nc, long HW) {
    __shared__ unsigned hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.y;
    const float* o = out + (size_t)b * nc * HW;
    const int64_t* t = target + (size_t)b * HW;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < HW; r += (long)gridDim.x * 256) {
        float best = o[r]; int arg = 0;
        for (int c = 1; c < nc; ++c) {
            const float v = o[(size_t)c * HW + r];
            if (v > best) { best = v; arg = c; }          // strict: the first maximum wins (numpy.argmax)
        }
        const int tt = (int)t[r];
        if (tt >= 0 && tt < nc) atomicAdd(&hist[tt * nc + arg], 1u);
    }
    __syncthreads();
    if (threadIdx.x < nc * nc && hist[threadIdx.x]) atomicAdd(conf + (size_t)b * nc * nc + threadIdx.x, (unsigned long long)hist[threadIdx.x]);
}

extern "C" int mfc_confusion_counts(const float* outputs, const int64_t* target, int64_t* conf, int32_t B, int32_t nc, int32_t H,
                                    int32_t W, void* stream) {
    if (!outputs || !target || !conf || B <= 0 || nc < 2 || nc > 8 || H <= 0 || W <= 0) return MFC_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(conf, 0, (size_t)B * nc * nc * sizeof(int64_t), st) != hipSuccess) return MFC_ERR_LAUNCH;
    const long HW = (long)H * W;
    long bx = (HW + 255) / 256; if (bx > 256) bx = 256;
    hipLaunchKernelGGL(confusion_kernel, dim3((int)bx, B), dim3(256), 0, st, outputs, target, (unsigned long long*)conf, nc, HW);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// ------------------------------------------------------------------ Adam
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n, float step_size,
                                                   float beta1, float beta2, float eps, float inv_bc2_sqrt, float gscale, const int* skip,
                                                   float lr, int step) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (skip) {
        if (skip[0]) return;                 // (guarded step: the gradients of this step are not finite -- leave p, m, v alone)
        // bias correction from the steps actually APPLIED (torch.cuda.amp.GradScaler + Adam: a skipped step does not advance the
        // optimizer's step count); skip[1] = number of skipped steps so far.  Without skipped steps the host's double-precision
        // corrections are used unchanged.
        const int nskip = skip[1];
        if (nskip > 0) {
            const float eff = (float)max(step - nskip, 1);
            const float bc1 = 1.f - exp2f(eff * log2f(beta1)), bc2 = 1.f - exp2f(eff * log2f(beta2));
            step_size = lr / bc1;
            inv_bc2_sqrt = rsqrtf(bc2);
        }
    }
    if (i + 4 <= n) {
        float4 P = *(float4*)(p + i), G = *(const float4*)(g + i), M = *(float4*)(m + i), V = *(float4*)(v + i);
        float* pp = (float*)&P; float* gg = (float*)&G; float* mm = (float*)&M; float* vv = (float*)&V;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * gscale;
            mm[k] = mm[k] + (gk - mm[k]) * (1.f - beta1);
            vv[k] = vv[k] * beta2 + gk * gk * (1.f - beta2);
            pp[k] -= step_size * mm[k] / (sqrtf(vv[k]) * inv_bc2_sqrt + eps);
        }
        *(float4*)(p + i) = P; *(float4*)(m + i) = M; *(float4*)(v + i) = V;
    } else {
        for (long k = i; k < n; ++k) {
            const float gk = g[k] * gscale;
            m[k] = m[k] + (gk - m[k]) * (1.f - beta1);
            v[k] = v[k] * beta2 + gk * gk * (1.f - beta2);
            p[k] -= step_size * m[k] / (sqrtf(v[k]) * inv_bc2_sqrt + eps);
        }
    }
}

// any element of g not finite -> flag[0] = 1, flag[1] += 1 (count of such checks); flag[0] is cleared first, on the stream
__global__ __launch_bounds__(256) void grad_check_kernel(const float* g, long n, int* flag) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const float4 v = *(const float4*)(g + i);
            mine |= (int)!(fabsf(v.x) <= 3.4e38f) | (int)!(fabsf(v.y) <= 3.4e38f) | (int)!(fabsf(v.z) <= 3.4e38f) | (int)!(fabsf(v.w) <= 3.4e38f);
        } else {
            for (long k = i; k < n; ++k) mine |= (int)!(fabsf(g[k]) <= 3.4e38f);
        }
    }
    if (mine) bad = 1;
    __syncthreads();
    if (threadIdx.x == 0 && bad) { if (atomicExch(flag, 1) == 0) atomicAdd(flag + 1, 1); }
}

extern "C" int mfc_grad_check(const float* g, int64_t n, int32_t* flag, void* stream) {
    if (!g || !flag || n <= 0 || ((uintptr_t)g & 15)) return MFC_ERR_INVALID_ARG;
    if (hipMemsetAsync(flag, 0, 4, (hipStream_t)stream) != hipSuccess) return MFC_ERR_LAUNCH;
    long blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (g_mfc_prof_on == 1) mfc_prof_before((hipStream_t)stream, "grad_check_kernel", 0.0, 4.0 * (double)n);
    hipLaunchKernelGGL(grad_check_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, g, (long)n, (int*)flag);
    MFC_PROF_END((hipStream_t)stream);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

static int adam_step_impl(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                          float eps, int32_t step, float grad_scale, const int32_t* skip_flag, void* stream);
extern "C" int mfc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, int32_t step, float grad_scale, void* stream) {
    return adam_step_impl(p, g, m, v, n, lr, beta1, beta2, eps, step, grad_scale, nullptr, stream);
}
extern "C" int mfc_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                     float eps, int32_t step, float grad_scale, const int32_t* skip_flag, void* stream) {
    if (!skip_flag) return MFC_ERR_INVALID_ARG;
    return adam_step_impl(p, g, m, v, n, lr, beta1, beta2, eps, step, grad_scale, skip_flag, stream);
}
static int adam_step_impl(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                          float eps, int32_t step, float grad_scale, const int32_t* skip_flag, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return MFC_ERR_INVALID_ARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return MFC_ERR_INVALID_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const long blocks = ((n + 3) / 4 + 255) / 256;
    if (g_mfc_prof_on == 1) mfc_prof_before((hipStream_t)stream, "adam_kernel", 0.0, 28.0 * (double)n);
    hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, step_size, beta1, beta2, eps, inv_bc2_sqrt, grad_scale, (const int*)skip_flag, lr, (int)step);
    MFC_PROF_END((hipStream_t)stream);
    MFC_CHECK_LAUNCH();
    return MFC_OK;
}

// ------------------------------------------------------------------ descriptor hardening (common.h)
int g_mfc_validate_ptrs = 0;
int g_mfc_wt_min_mb = 12;             // write-through output stores from this many megabytes of output (common.h, mfc_st16); mfc_set_flag(54, MB)
bool mfc_ptrs_ok_impl(const void* const* p, int n) {
    for (int i = 0; i < n; ++i) {
        if (!p[i]) continue;
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p[i]) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeManaged) return false;
    }
    return true;
}

// ------------------------------------------------------------------ event profiler
// Rows are keyed by kernel NAME (the string rocprofv3 prints for that kernel, demangled), so bench.py's `roofline` object and the
// committed rocprofv3 kernel-stats table talk about the same rows.
#include <vector>
#include <new>
#include <string.h>
int g_mfc_prof_on = 0;
namespace {
struct ProfRec { hipEvent_t a, b; const char* name; double flops, bytes; hipStream_t st; bool shared_a = false; };
std::vector<ProfRec> g_log;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}
void mfc_prof_before(hipStream_t st, const char* name, double flops, double bytes) {
    ProfRec r; r.a = get_event(); r.b = get_event(); r.name = name; r.flops = flops; r.bytes = bytes; r.st = st;
    (void)hipEventRecord(r.a, st);
    g_log.push_back(r);
}
void mfc_prof_after(hipStream_t st) { (void)hipEventRecord(g_log.back().b, st); }
// section spans (mfc_prof_enable(2)): a timed event now on `st` ...
hipEvent_t mfc_prof_mark(hipStream_t st) { hipEvent_t e = get_event(); (void)hipEventRecord(e, st); return e; }
// ... and a row from an earlier mark to now-on-`st`; `tag` travels in the flops column
void mfc_prof_span(hipEvent_t from, hipStream_t st, const char* name, double tag) {
    ProfRec r; r.a = from; r.b = get_event(); r.name = name; r.flops = tag; r.bytes = 0; r.st = st; r.shared_a = true;
    (void)hipEventRecord(r.b, st);
    g_log.push_back(r);
}
extern "C" int mfc_prof_enable(int on) { g_mfc_prof_on = on; return MFC_OK; }
// Tuning aid: the recorded launches as a timeline (CSV: name, stream, start_us, end_us relative to the first recorded launch; event
// timestamps are device-wide, so launches on different streams are comparable).  Clears the log like mfc_prof_collect.
extern "C" int mfc_prof_dump(const char* path) {
    if (!path) return MFC_ERR_INVALID_ARG;
    FILE* f = fopen(path, "w");
    if (!f) return MFC_ERR_INVALID_ARG;
    fprintf(f, "name,stream,start_us,end_us\n");
    std::vector<hipStream_t> streams;
    for (auto& r : g_log) {
        if (hipEventSynchronize(r.b) != hipSuccess) { fclose(f); return MFC_ERR_LAUNCH; }
        float t0 = 0.f, t1 = 0.f;
        (void)hipEventElapsedTime(&t0, g_log.front().a, r.a);
        (void)hipEventElapsedTime(&t1, g_log.front().a, r.b);
        int si = -1;
        for (size_t i = 0; i < streams.size() && si < 0; ++i) if (streams[i] == r.st) si = (int)i;
        if (si < 0) { si = (int)streams.size(); streams.push_back(r.st); }
        if (r.shared_a) fprintf(f, "\"%s@%d\",%d,%.2f,%.2f\n", r.name, (int)r.flops, si, t0 * 1e3, t1 * 1e3);
        else fprintf(f, "\"%s\",%d,%.2f,%.2f\n", r.name, si, t0 * 1e3, t1 * 1e3);
    }
    fclose(f);
    for (auto& r : g_log) { if (!r.shared_a) g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_log.clear();
    return MFC_OK;
}
extern "C" int mfc_prof_collect(mfc_prof_entry* out, int32_t cap) {
    if (!out || cap <= 0) return MFC_ERR_INVALID_ARG;
    int n = 0;
    for (auto& r : g_log) {
        if (hipEventSynchronize(r.b) != hipSuccess) return MFC_ERR_LAUNCH;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        int row = -1;
        for (int i = 0; i < n && row < 0; ++i) if (!strcmp(out[i].name, r.name)) row = i;
        if (row < 0 && n < cap) {
            row = n++;
            memset(&out[row], 0, sizeof(out[row]));
            strncpy(out[row].name, r.name, sizeof(out[row].name) - 1);
        }
        if (row >= 0) { out[row].ms += ms; out[row].flops += r.flops; out[row].bytes += r.bytes; out[row].launches += 1; }
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_log.clear();
    return n;
}

hipEvent_t mfc_prof_mark(hipStream_t st);
void mfc_prof_span(hipEvent_t from, hipStream_t st, const char* name, double tag);
// ------------------------------------------------------------------ program interpreter
extern "C" int mfc_op_size(void) { return (int)sizeof(mfc_op); }
extern "C" const char* mfc_version(void) { return "mfcnet_hip 0.1 (gfx950)"; }

static int run_one(const mfc_op& o, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (o.kind) {
        case MFC_OP_CONV: return mfc_conv2d_fwd(&o.u.conv, stream);
        case MFC_OP_WGRAD: return mfc_conv2d_wgrad(&o.u.wgrad, stream);
        case MFC_OP_WGRAD_BATCH: return mfc_conv2d_wgrad_batch((const mfc_wgrad_desc*)o.u.raw.a, o.u.raw.i[0], stream);
        case MFC_OP_BNFIN: return mfc_bn_finalize(&o.u.bnfin, stream);
        case MFC_OP_BNFIN_BATCH: return mfc_bn_finalize_batch((const mfc_bnfin_desc*)o.u.raw.a, o.u.raw.i[0], o.u.raw.i[1], stream);
        case MFC_OP_COMBINE: return mfc_combine_fwd(&o.u.combine, stream);
        case MFC_OP_BNBWD_REDUCE: return mfc_bnbwd_reduce(&o.u.bnbwd, stream);
        case MFC_OP_BNBWD_FIN: return mfc_bnbwd_finalize(&o.u.bnbwdfin, stream);
        case MFC_OP_BNBWD_APPLY: return mfc_bnbwd_apply(&o.u.bnbwd, stream);
        case MFC_OP_MASK_ADD: return mfc_mask_add(&o.u.maskadd, stream);
        case MFC_OP_HEAD_FWD: return mfc_head_gather_fwd(&o.u.head, stream);
        case MFC_OP_HEAD_BWD: return mfc_head_gather_bwd(&o.u.headbwd.d, (void*)o.u.headbwd.dlogits, stream);
        case MFC_OP_BIAS_GRAD:   // a = dy, b = db (or, with i[3] = nparts > 0, the slices), n = npix, i[0]=dtype, i[1]=Cp, i[2]=C
            if (o.u.raw.i[3] > 0)
                return mfc_bias_grad_slices((const void*)o.u.raw.a, (float*)o.u.raw.b, o.u.raw.i[0], o.u.raw.n, o.u.raw.i[1], o.u.raw.i[2], o.u.raw.i[3], stream);
            return mfc_bias_grad((const void*)o.u.raw.a, (float*)o.u.raw.b, o.u.raw.i[0], o.u.raw.n, o.u.raw.i[1], o.u.raw.i[2], stream);
        case MFC_OP_MEMSET:      // a = ptr, n = bytes
            return hipMemsetAsync((void*)o.u.raw.a, 0, (size_t)o.u.raw.n, st) == hipSuccess ? MFC_OK : MFC_ERR_LAUNCH;
        case MFC_OP_PACK:        // a = jobs, i[0]=njobs, i[1]=total_blocks, i[2]=dtype
            return mfc_pack_weights((const mfc_pack_job*)o.u.raw.a, o.u.raw.i[0], o.u.raw.i[1], o.u.raw.i[2], stream);
        case MFC_OP_UNPACK:      // a = jobs, i[0]=njobs, i[1]=total_blocks
            return mfc_unpack_wgrad((const mfc_unpack_job*)o.u.raw.a, o.u.raw.i[0], o.u.raw.i[1], stream);
        case MFC_OP_NCHW2NHWC:   // a = src, b = dst, i = dtype,N,C,H,W,Cp,c_off
            return mfc_nchw_to_nhwc((const float*)o.u.raw.a, (void*)o.u.raw.b, o.u.raw.i[0], o.u.raw.i[1], o.u.raw.i[2], o.u.raw.i[3],
                                    o.u.raw.i[4], o.u.raw.i[5], o.u.raw.i[6], 1, stream);
        case MFC_OP_NHWC2NCHW:   // a = src, b = dst, i = dtype,N,C,H,W,Cp
            return mfc_nhwc_to_nchw((const void*)o.u.raw.a, (float*)o.u.raw.b, o.u.raw.i[0], o.u.raw.i[1], o.u.raw.i[2], o.u.raw.i[3],
                                    o.u.raw.i[4], o.u.raw.i[5], stream);
        case MFC_OP_WSNORM: {    // a = w, b = w_out, i[0] = Cout, i[1] = per_out, i[2] = eps (float bits)
            float eps; memcpy(&eps, &o.u.raw.i[2], 4);
            return mfc_ws_normalize((const float*)o.u.raw.a, (float*)o.u.raw.b, o.u.raw.i[0], o.u.raw.i[1], eps, stream);
        }
        case MFC_OP_GNFIN: return mfc_gn_finalize(&o.u.gnfin, stream);
        case MFC_OP_UPNEAR:      // a = src, b = dst, i = dtype, N, H, W, Cp
            return mfc_upsample_nearest2x((const void*)o.u.raw.a, (void*)o.u.raw.b, o.u.raw.i[0], o.u.raw.i[1], o.u.raw.i[2], o.u.raw.i[3], o.u.raw.i[4], stream);
        case MFC_OP_GNBWD_FIN: return mfc_gnbwd_finalize(&o.u.gnbwdfin, stream);
        case MFC_OP_WSBWD: {     // a = w, b = dws, c = dw, i[0] = Cout, i[1] = per_out, i[2] = eps (float bits)
            float eps; memcpy(&eps, &o.u.raw.i[2], 4);
            return mfc_ws_backward((const float*)o.u.raw.a, (const float*)o.u.raw.b, (float*)o.u.raw.c, o.u.raw.i[0], o.u.raw.i[1], eps, stream);
        }
        case MFC_OP_UPNEAR_BWD:  // a = dsrc, b = ddst, i = dtype, N, H, W, Cp, accumulate
            return mfc_upsample_nearest2x_bwd((const void*)o.u.raw.a, (void*)o.u.raw.b, o.u.raw.i[0], o.u.raw.i[1], o.u.raw.i[2], o.u.raw.i[3], o.u.raw.i[4],
                                              o.u.raw.i[5], stream);
        case MFC_OP_JOIN: return MFC_OK;          // (the interpreter joins the side lanes at any lane-0 record: nothing to launch)
        default: return MFC_ERR_INVALID_ARG;
    }
}

// Lanes: records tagged lane >= 1 belong to a PARALLEL SECTION (a maximal run of such records, e.g. the independent
// branches of one HighResolutionModule, hrnet.py:242-243): lane 1 stays on the caller's stream, lanes 2.. run on side
// streams forked from it by an event and joined back at the next lane-0 record.  Records of different lanes in one section
// must not touch each other's outputs (the plan guarantees it: nothing in the arenas is reused, and a branch only writes
// its own tensors).  Under-filled launches of the low-resolution branches then share the GPU with the other branches.
#define MFC_MAX_LANES 8
#define MFC_ASYNC_STREAMS 4
#define MFC_ASYNC_EVENTS 16
// Streams of one device.  m = the interpreter's own main stream, s[2..] the side lanes, as[] the detached streams: created
// back to back so HIP's round-robin stream -> hardware-queue mapping (4 queues by default) gives the four streams in use
// (m, s[2], s[3], as[0]) distinct queues whatever stream the caller runs on.  (Measured: on a caller stream that happened to
// share a queue with a lane the lanes gained nothing; with >8 hardware queues the firmware time-slices and everything is slower.)
struct LaneSet {
    hipStream_t m, s[MFC_MAX_LANES + 1]; hipEvent_t enter, leave, fork, join[MFC_MAX_LANES + 1];
    hipStream_t as[MFC_ASYNC_STREAMS]; hipEvent_t aev[MFC_ASYNC_EVENTS], ajoin[MFC_ASYNC_STREAMS];
    hipEvent_t seg;          // recorded on the detached stream at the end of a program whose final join was deferred (mfc_wait_detached)
    bool pending;            // detached work of an earlier program has not been joined yet
    bool lane_b;             // a second side lane with a hardware queue of its own was found by the probe (s[3])
    bool lane_c;             // ... and a third one (s[4]; needs GPU_MAX_HW_QUEUES >= 5)
    hipEvent_t fork_t;       // (timed) event of the current section's fork, mfc_prof_enable(2)
    bool ready;
};
static LaneSet g_lanes[16];
static int g_lanes_on = 3;       // bit 0: parallel-section lanes, bit 1: detached (async) records
static int g_lane_streams = 3;   // streams the section lanes are folded onto (1 = main only; measured best: 3 + the detached stream, more co-running persistent kernels thrash); tuning: mfc_set_flag(12, n)
static int g_lane_map[MFC_MAX_LANES + 1] = {0, 1, 2, -3, 1, 0, 0, 0, 0};  // lane -> stream (0 = fold by modulo; -3 = stream 3 if the probe found a second side lane with a
                                                                          // hardware queue of its own, else stream 2).  The 120x160 and 15x20 branches on the main stream, the two middle
                                                                          // ones on a side stream each: with both on ONE side stream the main stream idled ~0.55 ms per 3-branch module
                                                                          // forward and ~0.84 ms backward (tools/lane_timeline.py).  A FOURTH concurrent chain (the 15x20 branch on a
                                                                          // stream / hardware queue of its own, GPU_MAX_HW_QUEUES >= 5, map 1234 or -4 here) costs 45 %
                                                                          // of the step (40.2 -> 58 ms, tools/sweep_queues.sh): the persistent convolution launches of
                                                                          // four chains queue behind each other on every CU.
int mfc_set_lane_streams(int n) {
    if (n >= 1000) {          // decimal digits = streams of lanes 1..4, e.g. 1223
        g_lane_map[1] = n / 1000 % 10; g_lane_map[2] = n / 100 % 10; g_lane_map[3] = n / 10 % 10; g_lane_map[4] = n % 10;
        return 0;
    }
    for (int i = 0; i <= MFC_MAX_LANES; ++i) g_lane_map[i] = 0;
    g_lane_streams = n < 1 ? 1 : (n > MFC_MAX_LANES ? MFC_MAX_LANES : n); return 0;
}
static int g_async_n = 1;        // async streams in use (1..MFC_ASYNC_STREAMS; measured: 1 is best, concurrent wgrads fight each other); tuning: mfc_set_flag(10, n)
int mfc_set_lanes(int on) { g_lanes_on = on; return 0; }
static int g_async_prio = 0;     // priority of the detached stream: 1 lowest, 0 default, -1 highest (read when the streams are created).  Measured:
                                 // either non-default priority costs 35 % of the step (526 -> 330 frames/s) -- keep 0; mfc_set_flag(16, v)
int mfc_set_async_prio(int v) { g_async_prio = v; return 0; }
static int g_lane4_fwd = 1;        // 1: in programs WITHOUT detached records (the forward pass) lane 4 -- the 15x20 branch of a four-branch module, otherwise folded onto the
                                   // main stream -- runs on the detached stream's otherwise idle hardware queue: step 36.42 -> 35.73 ms, eval forward 9.74 -> 9.60 ms; mfc_set_flag(56, v)
int mfc_set_lane4_fwd(int v) { g_lane4_fwd = v; return 0; }
static int g_defer_join = 0;     // 1: a program does not join the detached stream at its end (the next program of the step will); mfc_set_flag(28, v)
int mfc_set_defer_join(int v) { g_defer_join = v; return 0; }
static int g_skip_kinds = 0;     // tuning only: bit k set -> records of kind k are skipped (what-if timing); mfc_set_flag(15, mask)
int mfc_set_skip_kinds(int m) { g_skip_kinds = m; return 0; }
static int g_async_on_lane = 0;  // 0: detached records on their own stream; k >= 2: on side lane k's stream; tuning: mfc_set_flag(13, k)
static int g_own_main = 0;       // 1: the program's main stream is the interpreter's own (immune to a caller stream that shares a hardware queue with a
                                 // side stream, 1.5 % slower); 0: the caller's; tuning: mfc_set_flag(14, v)
int mfc_set_async_on_lane(int k) { g_async_on_lane = k; return 0; }
int mfc_set_own_main(int v) { g_own_main = v; return 0; }
int mfc_set_async_streams(int n) { g_async_n = n < 1 ? 1 : (n > MFC_ASYNC_STREAMS ? MFC_ASYNC_STREAMS : n); return 0; }

// ---- which of our streams really run next to each other? ----
// HIP folds streams onto GPU_MAX_HW_QUEUES (4) hardware queues in creation order, counting every stream the process ever made
// (torch's pool, RCCL's).  Two streams that land on one hardware queue execute in order, so a detached stream that shares
// the queue of the main chain (or of the side lane) silently loses the overlap: measured 43.5 -> 48.7 / 52.2 ms per step,
// depending only on how many streams existed before ours.  So the interpreter does not trust creation order: it makes a few
// candidates and MEASURES which ones overlap with the caller's stream (a 150 us spin on one, an empty kernel on the other).
__global__ void mfc_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    for (int guard = 0; guard < 8192 && wall_clock64() - t0 < ticks; ++guard) __builtin_amdgcn_s_sleep(16);
}
__global__ void mfc_empty_kernel() {}
static int g_probe_streams = 1;      // mfc_set_flag(22, v)
int mfc_set_probe_streams(int v) { g_probe_streams = v; return 0; }
static bool streams_overlap(hipStream_t a, hipStream_t b) {
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) { if (ea) (void)hipEventDestroy(ea); return true; }
    hipLaunchKernelGGL(mfc_spin_kernel, dim3(1), dim3(64), 0, a, (long long)15000);      // wall_clock64 ticks at 100 MHz
    (void)hipEventRecord(ea, a);
    hipLaunchKernelGGL(mfc_empty_kernel, dim3(1), dim3(64), 0, b);
    (void)hipEventRecord(eb, b);
    bool over = true;
    float ms = 0.f;
    if (hipEventSynchronize(ea) == hipSuccess && hipEventSynchronize(eb) == hipSuccess && hipEventElapsedTime(&ms, eb, ea) == hipSuccess)
        over = ms > 0.03f;           // the empty kernel finished well before the spin did
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    (void)hipGetLastError();
    return over;
}

// Interpreter state = one LaneSet.  The classic entry points (mfc_program_run / _ex, mfc_wait_detached, mfc_graph_capture) use the
// per-DEVICE default set g_lanes[dev]; mfc_ctx_create makes a set of its own that mfc_program_run_ctx / mfc_wait_detached_ctx take as an
// argument, so that two models (or two host threads, each with its device current) never share streams, events or the pending-join state.
struct mfc_ctx_s { int device; LaneSet L; };
static LaneSet* lanes_get(LaneSet* L, hipStream_t caller, bool may_probe);
static LaneSet* lanes_for_device(hipStream_t caller = nullptr, bool may_probe = false) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    return lanes_get(&g_lanes[dev], caller, may_probe);
}
static LaneSet* lanes_get(LaneSet* L, hipStream_t caller, bool may_probe) {
    if (!L->ready) {
        if (hipStreamCreateWithFlags(&L->m, hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (int i = 2; i <= 3; ++i)
            if (hipStreamCreateWithFlags(&L->s[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
        {   // (a low-priority detached stream looked attractive -- fill only what the critical chain leaves idle -- but see g_async_prio)
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            const int prio = g_async_prio == 0 ? 0 : (g_async_prio > 0 ? lo : hi);
            if (hipStreamCreateWithPriority(&L->as[0], hipStreamNonBlocking, prio) != hipSuccess) return nullptr;
        }
        for (int i = 4; i <= MFC_MAX_LANES; ++i)
            if (hipStreamCreateWithFlags(&L->s[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (int i = 1; i < MFC_ASYNC_STREAMS; ++i)
            if (hipStreamCreateWithFlags(&L->as[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&L->enter, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&L->leave, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&L->fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        for (int i = 2; i <= MFC_MAX_LANES; ++i)
            if (hipEventCreateWithFlags(&L->join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        for (int i = 0; i < MFC_ASYNC_STREAMS; ++i)
            if (hipEventCreateWithFlags(&L->ajoin[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&L->seg, hipEventDisableTiming) != hipSuccess) return nullptr;
        L->pending = false; L->lane_b = false; L->lane_c = false;
        for (int i = 0; i < MFC_ASYNC_EVENTS; ++i)
            if (hipEventCreateWithFlags(&L->aev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        if (may_probe && g_probe_streams && g_async_prio == 0) {
            // candidates: every side stream made above.  side lane A (s[2]) <- the first one that overlaps with the caller's stream;
            // detached (as[0]) <- the first other one that overlaps with the caller's stream AND with lane A; side lane B (s[3]) <- the
            // first other one that overlaps with all three (with the default 4 hardware queues: main, two side lanes, detached).
            std::vector<hipStream_t> h;
            for (int i = 2; i <= MFC_MAX_LANES; ++i) h.push_back(L->s[i]);
            for (int i = 0; i < MFC_ASYNC_STREAMS; ++i) h.push_back(L->as[i]);
            static const bool dbg = getenv("MFC_DEBUG") != nullptr;
            const int nc = (int)h.size();
            int pick_l = -1, pick_a = -1, pick_3 = -1, pick_4 = -1;
            for (int i = 0; i < nc && pick_l < 0; ++i)
                if (streams_overlap(caller, h[i])) pick_l = i;
            for (int i = 0; i < nc && pick_l >= 0 && pick_a < 0; ++i)
                if (i != pick_l && streams_overlap(caller, h[i]) && streams_overlap(h[pick_l], h[i])) pick_a = i;
            for (int i = 0; i < nc && pick_a >= 0 && pick_3 < 0; ++i)
                if (i != pick_l && i != pick_a && streams_overlap(caller, h[i]) && streams_overlap(h[pick_l], h[i]) && streams_overlap(h[pick_a], h[i])) pick_3 = i;
            // side lane C (s[4]): only with more hardware queues than HIP's default four (GPU_MAX_HW_QUEUES >= 5)
            for (int i = 0; i < nc && pick_3 >= 0 && pick_4 < 0; ++i)
                if (i != pick_l && i != pick_a && i != pick_3 && streams_overlap(caller, h[i]) && streams_overlap(h[pick_l], h[i]) &&
                    streams_overlap(h[pick_a], h[i]) && streams_overlap(h[pick_3], h[i])) pick_4 = i;
            if (dbg) fprintf(stderr, "[mfc lanes] stream probe: side lane A <- candidate %d, detached <- candidate %d, side lane B <- candidate %d, side lane C <- candidate %d (of %d)\n", pick_l, pick_a, pick_3, pick_4, nc);
            // hand the picked streams to their roles, the rest fill the remaining slots in order
            std::vector<hipStream_t> rest;
            for (int i = 0; i < nc; ++i) if (i != pick_l && i != pick_a && i != pick_3 && i != pick_4) rest.push_back(h[i]);
            size_t ri = 0;
            auto take = [&](int pick) { return pick >= 0 ? h[pick] : rest[ri++]; };
            L->s[2] = take(pick_l);
            L->as[0] = take(pick_a);
            L->s[3] = take(pick_3);
            L->s[4] = take(pick_4);
            for (int i = 5; i <= MFC_MAX_LANES; ++i) L->s[i] = rest[ri++];
            for (int i = 1; i < MFC_ASYNC_STREAMS; ++i) L->as[i] = rest[ri++];
            L->lane_b = pick_3 >= 0; L->lane_c = pick_4 >= 0;
        }
        L->ready = true;
    }
    return L;
}

static bool g_capturing = false;      // (a captured program stays on the capturing stream: see mfc_graph_capture)

// Detached records (lane bit MFC_LANE_ASYNC): work whose result nothing in the program reads before the next
// MFC_OP_UNPACK / the program end -- the weight gradients.  Such a record waits (event) for everything issued so far on
// the stream it would have run on, then runs on an extra stream, so the MFMA-bound wgrad launches overlap the HBM-bound
// BatchNorm-backward sweeps and the data-gradient chain instead of sitting in it.
static int program_run(const mfc_op* ops, int32_t n, void* stream, bool defer_join, LaneSet* own = nullptr);
extern "C" int mfc_program_run(const mfc_op* ops, int32_t n, void* stream) { return program_run(ops, n, stream, g_defer_join != 0); }
// the same with per-call options instead of process-wide switches: MFC_RUN_DEFER_JOIN = do not join the detached stream at the end (the
// next program of the step continues on it; see mfc_wait_detached)
extern "C" int mfc_program_run_ex(const mfc_op* ops, int32_t n, void* stream, uint32_t run_flags) {
    return program_run(ops, n, stream, (run_flags & MFC_RUN_DEFER_JOIN) != 0);
}
// ---- context handles: interpreter state as an argument instead of a per-device global (include/mfcnet_hip.h) ----
extern "C" int mfc_ctx_create(int32_t device, void** ctx_out) {
    if (!ctx_out || device < 0) return MFC_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
    if (ndev > 0 && device >= ndev) return MFC_ERR_INVALID_ARG;
    mfc_ctx_s* c = new (std::nothrow) mfc_ctx_s();
    if (!c) return MFC_ERR_LAUNCH;
    c->device = device;
    memset(&c->L, 0, sizeof(c->L));           // streams / events are made on first use, by the thread that has `device` current
    *ctx_out = (void*)c;
    return MFC_OK;
}
extern "C" int mfc_ctx_destroy(void* ctx) {
    mfc_ctx_s* c = (mfc_ctx_s*)ctx;
    if (!c) return MFC_ERR_INVALID_ARG;
    LaneSet& L = c->L;
    if (L.ready) {
        (void)hipStreamDestroy(L.m);
        for (int i = 2; i <= MFC_MAX_LANES; ++i) (void)hipStreamDestroy(L.s[i]);
        for (int i = 0; i < MFC_ASYNC_STREAMS; ++i) { (void)hipStreamDestroy(L.as[i]); (void)hipEventDestroy(L.ajoin[i]); }
        (void)hipEventDestroy(L.enter); (void)hipEventDestroy(L.leave); (void)hipEventDestroy(L.fork); (void)hipEventDestroy(L.seg);
        for (int i = 2; i <= MFC_MAX_LANES; ++i) (void)hipEventDestroy(L.join[i]);
        for (int i = 0; i < MFC_ASYNC_EVENTS; ++i) (void)hipEventDestroy(L.aev[i]);
        (void)hipGetLastError();
    }
    delete c;
    return MFC_OK;
}
static int ctx_check(mfc_ctx_s* c) {
    if (!c) return MFC_ERR_INVALID_ARG;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return MFC_ERR_LAUNCH;
    return dev == c->device ? MFC_OK : MFC_ERR_INVALID_ARG;      // the calling thread must have the context's device current
}
extern "C" int mfc_program_run_ctx(void* ctx, const mfc_op* ops, int32_t n, void* stream, uint32_t run_flags) {
    mfc_ctx_s* c = (mfc_ctx_s*)ctx;
    const int rc = ctx_check(c);
    if (rc != MFC_OK) return rc;
    return program_run(ops, n, stream, (run_flags & MFC_RUN_DEFER_JOIN) != 0, &c->L);
}
extern "C" int mfc_wait_detached_ctx(void* ctx, void* stream) {
    mfc_ctx_s* c = (mfc_ctx_s*)ctx;
    const int rc = ctx_check(c);
    if (rc != MFC_OK) return rc;
    if (!c->L.ready || !c->L.pending) return MFC_OK;
    return hipStreamWaitEvent((hipStream_t)stream, c->L.seg, 0) == hipSuccess ? MFC_OK : MFC_ERR_LAUNCH;
}

static int program_run(const mfc_op* ops, int32_t n, void* stream, bool defer_join, LaneSet* own) {
    if (!ops || n < 0) return MFC_ERR_INVALID_ARG;
    hipStream_t caller = (hipStream_t)stream;
    LaneSet* L = nullptr;
    bool multi = false;
    if (g_lanes_on)
        for (int i = 0; i < n && !multi; ++i) multi = ops[i].lane != 0;
    if (multi && !(L = (own ? lanes_get(own, caller, !g_capturing) : lanes_for_device(caller, !g_capturing)))) return MFC_ERR_LAUNCH;
    // with lanes the whole program runs on the interpreter's own streams, ordered after / before the caller's stream by events
    hipStream_t mainst = (multi && !g_capturing && g_own_main) ? L->m : caller;
    if (mainst != caller) { (void)hipEventRecord(L->enter, caller); (void)hipStreamWaitEvent(mainst, L->enter, 0); }
    // A program WITHOUT detached records (the forward pass: no weight gradients) leaves the detached stream -- the fourth hardware queue -- idle:
    // the fourth branch of a module (lane 4, otherwise folded onto the main stream) may run there (mfc_set_flag(56, 1)).
    bool has_async = false;
    for (int i = 0; i < n && !has_async && (g_lanes_on & 2); ++i) has_async = (ops[i].lane & MFC_LANE_ASYNC) != 0;
    const bool lane4_detached = multi && g_lane4_fwd && !has_async && !(L && L->pending) && !g_capturing;
    auto side = [&](int l) -> hipStream_t { return (lane4_detached && l == 4) ? L->as[0] : L->s[l]; };
    bool in_par = false; unsigned used = 0;
    unsigned aused = (L && L->pending) ? 1u : 0u;     // detached work left over from a program that deferred its join: same stream, in order
    int anext = 0, aevn = 0;
    // mfc_prof_enable(2): no per-launch events; instead one timeline row per parallel section and lane ("section <first record> lane <l>":
    // from the fork to the last record of that lane) -- a handful of events per module, so the step is not perturbed (tools/lane_sections.py)
    int sec_first = 0;
    auto join = [&]() {
        if (g_mfc_prof_on == 2) {
            static char names[MFC_MAX_LANES + 1][32];
            for (int l = 1; l <= MFC_MAX_LANES; ++l)
                if (l == 1 || (used & (1u << l))) {
                    snprintf(names[l], sizeof(names[l]), "lane%d", l);
                    mfc_prof_span(L->fork_t, l == 1 ? mainst : side(l), names[l], (double)sec_first);
                }
        }
        for (int l = 2; l <= MFC_MAX_LANES; ++l)
            if (used & (1u << l)) { (void)hipEventRecord(L->join[l], side(l)); (void)hipStreamWaitEvent(mainst, L->join[l], 0); }
        in_par = false; used = 0;
    };
    auto join_async = [&]() {
        for (int a = 0; a < MFC_ASYNC_STREAMS; ++a)
            if (aused & (1u << a)) {
                hipStream_t src = (g_async_on_lane >= 2 && g_async_on_lane <= MFC_MAX_LANES) ? L->s[g_async_on_lane] : L->as[a];
                (void)hipEventRecord(L->ajoin[a], src); (void)hipStreamWaitEvent(mainst, L->ajoin[a], 0);
            }
        aused = 0;
    };
    auto leave = [&]() {
        if (mainst != caller) { (void)hipEventRecord(L->leave, mainst); (void)hipStreamWaitEvent(caller, L->leave, 0); }
    };
    for (int i = 0; i < n; ++i) {
        int lane = (multi && (g_lanes_on & 1)) ? (ops[i].lane & 0xff) : 0;
        if (lane == 4 && lane4_detached) lane = 4;
        else if (lane >= 1 && lane <= MFC_MAX_LANES && g_lane_map[lane]) lane = g_lane_map[lane] == -3 ? ((L && L->lane_b) ? 3 : 2) : g_lane_map[lane] == -4 ? ((L && L->lane_c) ? 4 : 1) : g_lane_map[lane];
        else if (lane > g_lane_streams) lane = (lane - 1) % g_lane_streams + 1;      // fold the lanes onto the streams in use
        bool detached = multi && (g_lanes_on & 2) && (ops[i].lane & MFC_LANE_ASYNC);
        // a detached UNPACK sums the slices of weight gradients launched before it on the detached stream: in order only if that is ONE stream
        if (detached && ops[i].kind == MFC_OP_UNPACK && g_async_n > 1 && !(g_async_on_lane >= 2 && g_async_on_lane <= MFC_MAX_LANES)) detached = false;
        if (ops[i].kind == MFC_OP_UNPACK && !detached) lane = 0;      // (then it is ordinary serial work: joins the side lanes and the detached stream)
        hipStream_t st = mainst;
        if (lane >= 2 && lane <= MFC_MAX_LANES) {
            if (!in_par) { (void)hipEventRecord(L->fork, mainst); in_par = true; used = 0; sec_first = i; if (g_mfc_prof_on == 2) L->fork_t = mfc_prof_mark(mainst); }
            if (!(used & (1u << lane))) { (void)hipStreamWaitEvent(side(lane), L->fork, 0); used |= 1u << lane; }
            st = side(lane);
        } else if (lane == 1) {
            if (!in_par) { (void)hipEventRecord(L->fork, mainst); in_par = true; used = 0; sec_first = i; if (g_mfc_prof_on == 2) L->fork_t = mfc_prof_mark(mainst); }
        } else if (in_par) {
            join();
        }
        if (ops[i].kind == MFC_OP_UNPACK && aused && !detached) join_async();       // (lane 0: the side lanes were joined just above)
        if (detached) {
            hipEvent_t ev = L->aev[aevn]; aevn = (aevn + 1) % MFC_ASYNC_EVENTS;
            (void)hipEventRecord(ev, st);
            hipStream_t dst = (g_async_on_lane >= 2 && g_async_on_lane <= MFC_MAX_LANES) ? L->s[g_async_on_lane] : L->as[anext];
            if (dst != st) (void)hipStreamWaitEvent(dst, ev, 0);
            st = dst;
            aused |= 1u << anext; anext = (anext + 1) % g_async_n;
        }
        const int rc = (g_skip_kinds >> ops[i].kind) & 1 ? MFC_OK : run_one(ops[i], (void*)st);
        if (rc != MFC_OK) { if (in_par) join(); if (aused) join_async(); leave(); return -(1000 * (i + 1)) + rc; }
    }
    if (in_par) join();
    const bool can_defer = defer_join && !g_capturing && g_async_n == 1 && !(g_async_on_lane >= 2 && g_async_on_lane <= MFC_MAX_LANES);
    if (aused && can_defer) {
        // the caller continues with another program of the same step (backward segments): the detached stream is NOT joined here; an
        // event marks this point on it, for whoever needs the detached results so far (mfc_wait_detached), and the next program's
        // final join covers everything (one stream, in order)
        (void)hipEventRecord(L->seg, L->as[0]);
        L->pending = true;
    } else if (aused) {
        join_async();
        if (L) L->pending = false;
    }
    leave();
    return MFC_OK;
}

// Make `stream` wait for the detached records issued so far by programs that deferred their final join (mfc_set_flag(28, 1)).
extern "C" int mfc_wait_detached(void* stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return MFC_ERR_LAUNCH;
    LaneSet* L = &g_lanes[dev];
    if (!L->ready || !L->pending) return MFC_OK;       // nothing outstanding
    return hipStreamWaitEvent((hipStream_t)stream, L->seg, 0) == hipSuccess ? MFC_OK : MFC_ERR_LAUNCH;
}

// hipGraph capture of a whole program (lanes become parallel graph branches).  `stream` must not be the null stream.
// The caller must have run the program once normally (first launches set function attributes, which is not a stream operation).
extern "C" int mfc_graph_capture(const mfc_op* ops, int32_t n, void* stream, void** exec_out) {
    if (!ops || n <= 0 || !stream || !exec_out) return MFC_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (g_mfc_prof_on) return MFC_ERR_UNSUPPORTED;
    if (!lanes_for_device(st, true)) return MFC_ERR_LAUNCH;          // create (and probe) side streams / events outside the capture
    if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) return MFC_ERR_LAUNCH;
    g_capturing = true;
    const int rc = mfc_program_run(ops, n, stream);
    g_capturing = false;
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != MFC_OK || e != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); return rc != MFC_OK ? rc : MFC_ERR_LAUNCH; }
    hipGraphExec_t ex = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e2 != hipSuccess || !ex) return MFC_ERR_LAUNCH;
    *exec_out = (void*)ex;
    return MFC_OK;
}
extern "C" int mfc_graph_launch(void* exec, void* stream) {
    if (!exec) return MFC_ERR_INVALID_ARG;
    return hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream) == hipSuccess ? MFC_OK : MFC_ERR_LAUNCH;
}
extern "C" int mfc_graph_destroy(void* exec) {
    if (!exec) return MFC_ERR_INVALID_ARG;
    return hipGraphExecDestroy((hipGraphExec_t)exec) == hipSuccess ? MFC_OK : MFC_ERR_LAUNCH;
}

// Tuning aid: run the program with a HIP event between consecutive records (each record `reps` times back to back) and
// return the stream time of every record in ms_out[n] (per repetition).  Synchronises the stream.
extern "C" int mfc_program_profile(const mfc_op* ops, int32_t n, int32_t reps, float* ms_out, void* stream) {
    if (!ops || n <= 0 || !ms_out || reps < 1) return MFC_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t* ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * (size_t)(n + 1));
    if (!ev) return MFC_ERR_INVALID_ARG;
    for (int i = 0; i <= n; ++i) (void)hipEventCreate(&ev[i]);
    int rc = MFC_OK;
    for (int i = 0; i < n && rc == MFC_OK; ++i) {
        (void)hipEventRecord(ev[i], st);
        for (int r = 0; r < reps && rc == MFC_OK; ++r) {
            rc = run_one(ops[i], stream);
            if (rc != MFC_OK) rc = -(1000 * (i + 1)) + rc;
        }
    }
    (void)hipEventRecord(ev[n], st);
    (void)hipStreamSynchronize(st);
    if (rc == MFC_OK)
        for (int i = 0; i < n; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            ms_out[i] = ms / (float)reps;
        }
    for (int i = 0; i <= n; ++i) (void)hipEventDestroy(ev[i]);
    free(ev);
    return rc;
}
