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

}  // namespace ymx
