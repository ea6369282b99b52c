// Shared device helpers for libmfcnet_hip (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "mfcnet_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MFC_R MFC_STAT_REPLICAS

// A granule = 16 bytes of one pixel's channels: 4 floats or 8 bf16.
template <typename T> struct Gran;
template <> struct Gran<float> {
    static constexpr int E = 4;
    typedef float4 Quad;                                       // 4 consecutive channels
    __device__ static inline void unquad(const float4& v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    __device__ static inline float4 quad(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
    __device__ static inline void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
__device__ inline unsigned bf16_bits(float f) {
    bf16_t b = (bf16_t)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
// ReLU that keeps NaN (torch.relu(nan) = nan; v_max_f32 would return the other operand and silently swallow a diverged value,
// which the reference's `math.isnan(loss.item())` check, src/engine.py:67, relies on seeing)
__device__ inline float relu_nan(float t, float floor = 0.f) { return t < floor ? floor : t; }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ inline unsigned pack_bf16x2(float lo, float hi) {      // one v_cvt_pk_bf16_f32
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <> struct Gran<bf16_t> {
    static constexpr int E = 8;
    typedef uint2 Quad;                                        // 4 consecutive channels
    __device__ static inline void unquad(const uint2& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    }
    __device__ static inline uint2 quad(const float* f) { return make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3])); }
    __device__ static inline void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
};

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ inline unsigned pack_f16x2(float lo, float hi) {       // RNE (v_cvt_pk_f16_f32 / two v_cvt_f16_f32), NaN and Inf kept
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
__device__ inline void unpack_f16x2(unsigned u, float& lo, float& hi) {
    const f16x2_t h = __builtin_bit_cast(f16x2_t, u);
    lo = (float)h[0]; hi = (float)h[1];
}
// fp16 storage (BASELINE configs[4]): same granule shape as bf16; a training step in it needs the loss scale
template <> struct Gran<f16_t> {
    static constexpr int E = 8;
    typedef uint2 Quad;
    __device__ static inline void unquad(const uint2& v, float* f) { unpack_f16x2(v.x, f[0], f[1]); unpack_f16x2(v.y, f[2], f[3]); }
    __device__ static inline uint2 quad(const float* f) { return make_uint2(pack_f16x2(f[0], f[1]), pack_f16x2(f[2], f[3])); }
    __device__ static inline void unpack(const uint4& v, float* f) {
        unpack_f16x2(v.x, f[0], f[1]); unpack_f16x2(v.y, f[2], f[3]); unpack_f16x2(v.z, f[4], f[5]); unpack_f16x2(v.w, f[6], f[7]);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(pack_f16x2(f[0], f[1]), pack_f16x2(f[2], f[3]), pack_f16x2(f[4], f[5]), pack_f16x2(f[6], f[7]));
    }
};
// two floats -> one dword of T (the 16-bit types)
template <typename T> __device__ inline unsigned pack2(float lo, float hi);
template <> __device__ inline unsigned pack2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ inline unsigned pack2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }
template <typename T> __device__ inline void unpack2(unsigned u, float& lo, float& hi);
template <> __device__ inline void unpack2<bf16_t>(unsigned u, float& lo, float& hi) { lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xffff0000u); }
template <> __device__ inline void unpack2<f16_t>(unsigned u, float& lo, float& hi) { unpack_f16x2(u, lo, hi); }
// MFMA on 16-byte fragments held as raw bf16x8 registers (LDS reads are type-blind): the element type picks the instruction
template <typename T> __device__ inline f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c);
template <> __device__ inline f32x4 mfma16<bf16_t>(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
template <> __device__ inline f32x4 mfma16<f16_t>(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <typename T> __device__ inline f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c);
template <> __device__ inline f32x16 mfma32<bf16_t>(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
template <> __device__ inline f32x16 mfma32<f16_t>(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <typename T> __device__ inline float ld_elem(const T* p);
template <> __device__ inline float ld_elem<float>(const float* p) { return *p; }
template <> __device__ inline float ld_elem<bf16_t>(const bf16_t* p) { return (float)*p; }
template <> __device__ inline float ld_elem<f16_t>(const f16_t* p) { return (float)*p; }
template <typename T> __device__ inline void st_elem(T* p, float v);
template <> __device__ inline void st_elem<float>(float* p, float v) { *p = v; }
template <> __device__ inline void st_elem<bf16_t>(bf16_t* p, float v) { *p = (bf16_t)v; }
template <> __device__ inline void st_elem<f16_t>(f16_t* p, float v) { *p = (f16_t)v; }

// 16-byte store of an OUTPUT tensor.  A plain store leaves its line dirty in the XCD's L2 until it is evicted or the end-of-kernel release writes
// it back, and a kernel boundary costs + (dirty bytes) / 6 TB/s on top of its fixed part (MI355X_MICROARCH.md, "boundary"): a launch that
// writes a 15-30 MB tensor ends with most of the XCDs' 32 MB of L2 dirty.  `sc1` makes the store WRITE-THROUGH: the bytes leave for memory while
// the kernel is still running and nothing is left to flush (the line is dropped from L2, which the next launch -- another kernel, whose reads
// miss the non-coherent L2s anyway -- does not care about).  Measured on the ring convolution, 20 dependent launches (tools/probe_store.py,
// profiles/r04_store_flavours.txt): 32 -> 32 at 120x160 20.5 -> 18.4 us, 64 -> 64 at 60x80 17.4 -> 16.1 us; `nt` stores change nothing.
// Only 16-byte stores: narrower sc1 stores are one fabric write each (2.7x / 6x the time per byte for 8 / 4 bytes, same source).
// Inline asm because hipcc has no builtin for the sc1 bit on a global store; the trailing s_nop keeps the data registers intact until the
// store has read them (cdna_hip_programming.md 5.7).  -DMFC_PLAIN_STORES builds the write-back form for A/B runs.
typedef __attribute__((ext_vector_type(4))) unsigned mfc_u32x4;
__device__ inline void mfc_st16(void* addr, uint4 v) {
#ifdef MFC_PLAIN_STORES
    *(uint4*)addr = v;
#else
    const mfc_u32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(addr), "v"(d) : "memory");
#endif
}

// The write-through form pays only where the output is large: a small tensor (the 30x40 / 15x20 branches, 4-7 MB) stays resident in the XCDs' L2s
// across the kernel boundary and the consumer -- same XCD-contiguous tile order -- hits it there, which sc1 (the line is dropped) gives up.
// Measured per kernel family on the serial step (profiles/r04_store_flavours.txt): ring convolutions 5.34 -> 5.02 ms, bnbwd_apply_fin 4.38 ->
// 4.14, combine_same 2.09 -> 1.98 with sc1; conv_igemm (mostly small tensors) 15.77 -> 16.03 and the two-pass up-sampling adjoint (its fp32
// scratch is re-read at once) 1.11 -> 1.26 the other way.  So the launchers choose per launch: write-through from g_mfc_wt_min_mb megabytes of
// output (mfc_set_flag(54, MB); 0 = never), and kernels whose output is re-read immediately keep plain stores.
extern int g_mfc_wt_min_mb;
static inline int mfc_wt_for(double out_bytes) { return g_mfc_wt_min_mb > 0 && out_bytes >= (double)g_mfc_wt_min_mb * 1048576.0; }
__device__ inline void mfc_st16_if(void* addr, uint4 v, int wt) {        // wt is wave-uniform (a kernel argument)
    if (wt) mfc_st16(addr, v);
    else *(uint4*)addr = v;
}

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous range of
// logical ids so neighbouring tiles (shared halos / shared input patch across cout blocks)
// hit the same L2.  Bijective for any nwg (guide T1).  Speed only, never correctness.
__device__ inline int xcd_remap(int bid, int nwg) {
    int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ inline float wave16_sum(float v) {      // sum over the 16 lanes sharing lane>>4
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// bilinear source index / weight, align_corners=False, explicit output size
// (ATen area_pixel_compute_source_index: scale = in/out, src = scale*(dst+0.5)-0.5 clamped at 0)
__device__ inline void bilin_src(int dst, int in_size, int out_size, int& i0, int& i1, float& lam) {
    float scale = (float)in_size / (float)out_size;
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    lam = s - (float)i0;
}

// event-based kernel timing (runtime.hip)
extern int g_mfc_prof_on;
// `name` must be a string with static storage: the kernel name as rocprofv3 prints it (demangled form); one profile row per name
void mfc_prof_before(hipStream_t st, const char* name, double flops, double bytes);
void mfc_prof_after(hipStream_t st);
#include <cstdio>
// name of a template instantiation, built once (function-static buffer)
#define MFC_PROF_NAME(buf, ...) static char buf[120]; if (!buf[0]) snprintf(buf, sizeof(buf), __VA_ARGS__)
// bracket an element-wise launch: MFC_PROF_EW(st, "kernel<%s>", tname, bytes) ... launch ... MFC_PROF_END(st)
#define MFC_PROF_END(st) do { if (g_mfc_prof_on == 1) mfc_prof_after(st); } while (0)
template <typename T> inline const char* mfc_tname();
template <> inline const char* mfc_tname<float>() { return "float"; }
template <> inline const char* mfc_tname<bf16_t>() { return "__bf16"; }
template <> inline const char* mfc_tname<f16_t>() { return "_Float16"; }
static inline const char* mfc_dtname(int dtype) { return dtype == MFC_BF16 ? "__bf16" : (dtype == MFC_F16 ? "_Float16" : "float"); }
static inline bool mfc_is16(int dtype) { return dtype == MFC_BF16 || dtype == MFC_F16; }
static inline bool mfc_dtype_ok(int dtype) { return dtype == MFC_F32 || mfc_is16(dtype); }
// run a statement with T bound to the element type of `dtype` (which the caller has validated)
#define MFC_TYPED(dtype, T, ...) do { \
    if ((dtype) == MFC_BF16) { typedef bf16_t T; __VA_ARGS__; } \
    else if ((dtype) == MFC_F16) { typedef f16_t T; __VA_ARGS__; } \
    else { typedef float T; __VA_ARGS__; } } while (0)
// the same for kernels that exist for the 16-bit types only
#define MFC_TYPED16(dtype, T, ...) do { \
    if ((dtype) == MFC_F16) { typedef f16_t T; __VA_ARGS__; } else { typedef bf16_t T; __VA_ARGS__; } } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// Optional descriptor hardening (mfc_set_flag(53, 1); the -m gpu test session switches it on): every non-null pointer of a descriptor must
// be DEVICE memory known to the HIP runtime (hipPointerGetAttributes), otherwise the entry point returns MFC_ERR_INVALID_ARG instead of
// launching a kernel that would fault the GPU.  Round 3 lost two test runs to exactly that: descriptors cloned from a dry plan (placeholder
// addresses) reached mfc_conv2d_fwd / mfc_combine_fwd with the placeholder of the folded BatchNorm finalize still in `in_fin` / `fin`.
extern int g_mfc_validate_ptrs;
bool mfc_ptrs_ok_impl(const void* const* p, int n);
template <typename... P> static inline bool mfc_ptrs_ok(P... ptrs) {
    if (!g_mfc_validate_ptrs) return true;
    const void* a[] = {(const void*)(uintptr_t)ptrs...};
    return mfc_ptrs_ok_impl(a, (int)(sizeof(a) / sizeof(a[0])));
}
#define MFC_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return MFC_ERR_LAUNCH; } while (0)
