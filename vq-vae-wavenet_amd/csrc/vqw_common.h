// Shared host/device helpers of libvqwave (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vqwave.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// 4 floats with only 4-byte alignment: lets the compiler emit one global_load_dwordx4 for
// the time-shifted (dilated) activation windows, whose start is not 16-byte aligned.
struct __attribute__((packed, aligned(4))) F4U {
    float x, y, z, w;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__
// Raw buffer access: a 16-byte load from an address that is only 4-byte aligned (the dilated
// window of a conv tap starts at t0 - j*d) is ONE buffer_load_dwordx4 this way; through a plain
// pointer hipcc splits it into four global_load_dword (alignment 4), i.e. 4x the TA work.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vqw_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 vqw_buf_load4(__amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0);
    return __builtin_bit_cast(f32x4, v);
}
// LDS-DMA: 16 bytes per lane global -> LDS with no VGPR in between (buffer_load_dwordx4 ... lds).  The 64 lanes
// of a wave fill 1 KiB starting at the wave-uniform `lds_dst` (lane i lands at +16 i); the SOURCE address is per
// lane.  Counts on vmcnt.  (The builtin only exists in the device pass; called directly from a template kernel it
// makes the host pass drop the kernel's launch stub without a diagnostic, hence this wrapper and the guard.)
__device__ __forceinline__ void vqw_buf_load_lds16(__amdgpu_buffer_rsrc_t r, float* lds_dst, int voff_bytes, int soff_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff_bytes, soff_bytes, 0, 0);
#endif
}
#endif

int vqw_set_error(const char* fmt, ...);
int vqw_device_cus();                     // CU count of the calling thread's current device
int vqw_env_enabled(const char* name);    // 0 only if the environment variable starts with '0'

#define VQW_CHECK(cond, ...)                           \
    do {                                               \
        if (!(cond)) return vqw_set_error(__VA_ARGS__); \
    } while (0)

#define VQW_LAUNCH_CHECK(name)                                                     \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess)                                                      \
            return vqw_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline int vqw_cdiv(int a, int b) { return (a + b - 1) / b; }

// XCD-aware block remap (bijective for any grid size): consecutive logical ids land on the
// same XCD so that neighbouring tiles (which share an operand panel) share one L2.
__device__ __forceinline__ int vqw_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
