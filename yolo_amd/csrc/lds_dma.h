// LDS-DMA helpers shared by the large-tile convolution kernels.
#pragma once
#include "common.h"

namespace ymx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

// 128-bit raw buffer descriptor (base, stride 0, byte extent) held in SGPRs.
__device__ __forceinline__ i32x4_t make_rsrc(const void *p, unsigned bytes) {
    const unsigned long long addr = (unsigned long long)p;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(addr & 0xffffffffull));
    r[1] = __builtin_amdgcn_readfirstlane((int)((addr >> 32) & 0xffffull));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = __builtin_amdgcn_readfirstlane(0x00020000);
    return r;
}

// One LDS-DMA wave-instruction: lane l's 16 bytes at buffer offset `voff` land at LDS byte
// address lds_addr + 16*l (zeros if voff is out of range).  Written as inline asm on
// purpose: the compiler neither counts it in vmcnt nor orders LDS reads behind it, so
// the transfer overlaps the MFMAs that follow; the caller waits (vmcnt) and then
// barriers before any wave reads the bytes.  M0 carries the LDS base and is restored.
__device__ __forceinline__ void lds_dma16(const i32x4_t rsrc, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 4\n\t"
                 "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(rsrc)
                 : "memory");
}

// The same inside a steady-state loop: only the one wait state between the scalar write of M0
// and the LDS-DMA that reads it is inserted.  The five wait states of lds_dma16 cover a
// descriptor or LDS address that a VECTOR instruction (v_readfirstlane) has just written;
// callers of this form pass scalar-computed, long-lived operands.  (An immediate `offset:` on
// the instruction is no way to save the offset arithmetic: the hardware adds it to the LDS
// address as well as to the buffer offset.)
__device__ __forceinline__ void lds_dma16_steady(const i32x4_t rsrc, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(rsrc)
                 : "memory");
}

}  // namespace ymx

// ---- in-kernel time stamps (diagnostic build only: make stamps) -----------------------
// With YMX_STAMPS defined, thread 0 of each of the first YMX_STAMP_BLOCKS workgroups
// records the shader clock at up to 8 points into a buffer nothing else reads;
// ymx_debug_stamps() copies it out.  Slot 7 holds the 100 MHz wall clock at slot 0's time
// and slot 6 at the last stamp, which gives the shader clock rate.  The product build
// defines none of this.
#ifdef YMX_STAMPS
#define YMX_STAMP_BLOCKS 16384
namespace ymx {
static __device__ unsigned long long ymx_stamp_buf[YMX_STAMP_BLOCKS * 8];
}
#define YMX_STAMP(k)                                                                        \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < YMX_STAMP_BLOCKS) {                            \
            ymx::ymx_stamp_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();        \
            if ((k) == 0) ymx::ymx_stamp_buf[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                   \
    } while (0)
#define YMX_STAMP_END(k)                                                                    \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < YMX_STAMP_BLOCKS) {                            \
            ymx::ymx_stamp_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();        \
            ymx::ymx_stamp_buf[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();      \
        }                                                                                   \
    } while (0)
#define YMX_STAMP_EXPORT(fn)                                                                \
    extern "C" int fn(void *host, size_t bytes) {                                           \
        if (bytes > sizeof(unsigned long long) * YMX_STAMP_BLOCKS * 8) return -1;           \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ymx::ymx_stamp_buf), bytes);       \
    }
#else
#define YMX_STAMP(k) do { } while (0)
#define YMX_STAMP_END(k) do { } while (0)
#define YMX_STAMP_EXPORT(fn)
#endif

// ---- cycles of ONE region of a kernel's main loop (diagnostic build only: make regions) ---------
// Built with -DYMX_REGION=n, the code between YMX_REGION_BEGIN(n) and YMX_REGION_END(n) is
// bracketed by two reads of the shader clock and the differences are summed per wave over the
// loop; YMX_REGION_STORE leaves the sums of the first 2048 workgroups in a buffer nothing else
// reads (ymx_debug_regions copies it out).  One region per build, so that the two clock reads
// perturb only the region they measure; tools/region_time.py builds and runs all of them.  The
// product build defines none of this.
#ifdef YMX_REGION
namespace ymx {
static __device__ unsigned long long ymx_region_buf[2048 * 8];
}
#define YMX_REGION_DECL unsigned long long ymx_racc_ = 0, ymx_r0_ = 0; (void)ymx_r0_
#define YMX_REGION_BEGIN(n) do { if ((n) == YMX_REGION) ymx_r0_ = __builtin_amdgcn_s_memtime(); } while (0)
#define YMX_REGION_END(n) do { if ((n) == YMX_REGION) ymx_racc_ += __builtin_amdgcn_s_memtime() - ymx_r0_; } while (0)
#define YMX_REGION_STORE(wave, lane) do { if ((lane) == 0 && blockIdx.x < 2048) ymx::ymx_region_buf[blockIdx.x * 8 + (wave)] = ymx_racc_; } while (0)
#define YMX_REGION_EXPORT                                                                       \
    extern "C" int ymx_debug_regions(void *host, size_t bytes) {                                \
        if (bytes > sizeof(unsigned long long) * 2048 * 8) return -1;                           \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ymx::ymx_region_buf), bytes);          \
    }
#else
#define YMX_REGION_DECL do { } while (0)
#define YMX_REGION_BEGIN(n) do { } while (0)
#define YMX_REGION_END(n) do { } while (0)
#define YMX_REGION_STORE(wave, lane) do { } while (0)
#define YMX_REGION_EXPORT
#endif
