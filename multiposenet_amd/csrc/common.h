// Common device/host helpers for libmpn_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/mpn.h"

// ---------------------------------------------------------------- errors
void mpn_set_error(const char* fmt, ...);

#define MPN_FAIL(code, ...)            \
    do {                               \
        mpn_set_error(__VA_ARGS__);    \
        return (code);                 \
    } while (0)

#define MPN_REQUIRE(cond, code, ...)   \
    do {                               \
        if (!(cond)) MPN_FAIL(code, __VA_ARGS__); \
    } while (0)

#define MPN_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess)                                                     \
            MPN_FAIL(MPN_ERR_HIP, "%s:%d launch failed: %s", __FILE__, __LINE__, \
                     hipGetErrorString(e_));                                      \
    } while (0)

#define MPN_HIP(call)                                                             \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess)                                                     \
            MPN_FAIL(MPN_ERR_HIP, "%s:%d %s: %s", __FILE__, __LINE__, #call,      \
                     hipGetErrorString(e_));                                      \
    } while (0)

// Dynamic-LDS limit of a kernel above the 64 KB default: HIP function attributes are PER DEVICE, so the "already set"
// record is a bit per device ordinal (one process may drive several GPUs). `mask` is a static of the calling launcher,
// atomic: two host threads may race on a kernel's first launch - both then set the (idempotent) attribute, neither
// launches before it is set, and no bit of another device is lost.
typedef std::atomic<unsigned long long> mpn_attr_mask_t;
static inline hipError_t mpn_ensure_dynamic_lds(const void* func, int bytes, mpn_attr_mask_t* mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (mask->load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) mask->fetch_or(bit, std::memory_order_release);
    return e;
}

static inline bool mpn_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---------------------------------------------------------------- storage types
// Activations are stored either as f32 (parity build) or bf16 (throughput build);
// all arithmetic accumulates in f32.
typedef __bf16 bf16_t;
typedef _Float16 half_t;   // MPN_F16: fp16 storage (BASELINE config 5 runs the pose residual network in fp16)

template <typename T> struct StoreTraits;
template <> struct StoreTraits<float> {
    static constexpr int kDtype = MPN_F32;
    static constexpr int kVec = 4;  // elements per 16-byte vector
};
template <> struct StoreTraits<bf16_t> {
    static constexpr int kDtype = MPN_BF16;
    static constexpr int kVec = 8;
};
template <> struct StoreTraits<half_t> {
    static constexpr int kDtype = MPN_F16;
    static constexpr int kVec = 8;
};

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
__device__ __forceinline__ float to_f32(half_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }
template <> __device__ __forceinline__ half_t from_f32<half_t>(float x) { return (half_t)x; }   // v_cvt_f16_f32: round to nearest even

// Two f32 -> one dword of two bf16 (lo in bits 0..15): ONE v_cvt_pk_bf16_f32 (RNE, a NaN stays a NaN). Written as two scalar
// casts + shift + or, hipcc emits two converts, a shift and a v_or_b32_sdwa - four instructions per pair in every staging and
// epilogue path (found in the 3x3 kernel's commit: 48 converts + 24 merges for 24 pairs).
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_t));
}

// 16-byte vector of T, unpacked to / packed from f32 registers.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float4 raw;
    __device__ __forceinline__ void load(const float* p) { raw = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void load_nt(const float* p) {
        typedef float f32x4n_t __attribute__((ext_vector_type(4)));
        const f32x4n_t q = __builtin_nontemporal_load(reinterpret_cast<const f32x4n_t*>(p));
        raw = make_float4(q.x, q.y, q.z, q.w);
    }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = raw; }
    __device__ __forceinline__ void zero() { raw = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void unpack(float (&f)[4]) const { f[0] = raw.x; f[1] = raw.y; f[2] = raw.z; f[3] = raw.w; }
    __device__ __forceinline__ void pack(const float (&f)[4]) { raw = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    uint4 raw;
    __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void load_nt(const bf16_t* p) {   // streamed once: do not keep the line
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        raw = make_uint4(q.x, q.y, q.z, q.w);
    }
    __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<uint4*>(p) = raw; }
    __device__ __forceinline__ void zero() { raw = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ void unpack(float (&f)[8]) const {
        const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(u[i] << 16);
            f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void pack(const float (&f)[8]) {
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
        }
        raw = make_uint4(u[0], u[1], u[2], u[3]);
    }
};

template <> struct Vec16<half_t> {
    static constexpr int N = 8;
    uint4 raw;
    __device__ __forceinline__ void load(const half_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void load_nt(const half_t* p) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        raw = make_uint4(q.x, q.y, q.z, q.w);
    }
    __device__ __forceinline__ void store(half_t* p) const { *reinterpret_cast<uint4*>(p) = raw; }
    __device__ __forceinline__ void zero() { raw = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ void unpack(float (&f)[8]) const {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const h2_t h = __builtin_bit_cast(h2_t, u[i]);
            f[2 * i] = (float)h[0];
            f[2 * i + 1] = (float)h[1];
        }
    }
    __device__ __forceinline__ void pack(const float (&f)[8]) {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const h2_t h = {(_Float16)f[2 * i], (_Float16)f[2 * i + 1]};
            u[i] = __builtin_bit_cast(unsigned, h);
        }
        raw = make_uint4(u[0], u[1], u[2], u[3]);
    }
};

// The 16-bit storage types on the MFMA path: fragment vectors, v_mfma_f32_16x16x32_{bf16,f16}, transposing LDS read.
template <typename T> struct H16;
template <> struct H16<bf16_t> {
    typedef __bf16 x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 x4 __attribute__((ext_vector_type(4)));
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(const x8& a, const x8& b, const acc_t& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ x4 tr_read(const unsigned char* p) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((x4 __attribute__((address_space(3)))*)(p));
    }
};
template <> struct H16<half_t> {
    typedef _Float16 x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 x4 __attribute__((ext_vector_type(4)));
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(const x8& a, const x8& b, const acc_t& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ x4 tr_read(const unsigned char* p) {
        typedef __fp16 fp4_t __attribute__((ext_vector_type(4)));   // (the builtin's element type; same bits as _Float16)
        return __builtin_bit_cast(x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((fp4_t __attribute__((address_space(3)))*)(p)));
    }
};

// ---------------------------------------------------------------- wave helpers (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int mpn_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// the dense-conv / PRN entry points also take fp16 storage
#define MPN_DISPATCH_DTYPE3(dtype, ...)                                 \
    do {                                                                \
        if ((dtype) == MPN_F32) { using T = float; __VA_ARGS__; }       \
        else if ((dtype) == MPN_BF16) { using T = bf16_t; __VA_ARGS__; }\
        else if ((dtype) == MPN_F16) { using T = half_t; __VA_ARGS__; } \
        else MPN_FAIL(MPN_ERR_BAD_DTYPE, "unsupported dtype %d", (int)(dtype)); \
    } while (0)

#define MPN_DISPATCH_DTYPE(dtype, ...)                                  \
    do {                                                                \
        if ((dtype) == MPN_F32) { using T = float; __VA_ARGS__; }       \
        else if ((dtype) == MPN_BF16) { using T = bf16_t; __VA_ARGS__; }\
        else MPN_FAIL(MPN_ERR_BAD_DTYPE, "unsupported dtype %d", (int)(dtype)); \
    } while (0)
