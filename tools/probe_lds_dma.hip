// Hardware probe: what does a raw buffer load with an LDS destination do for lanes whose
// offset is out of range, and where does each lane's 16 bytes land?
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_lds_dma probe_lds_dma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* x, unsigned* out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* s = (unsigned*)smem;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = 0xAAAAAAAAu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // odd lanes out of range; even lanes read 16 B at a permuted source position
    unsigned off = (lane & 1) ? 0x80000000u : (unsigned)((wave * 64 + (lane ^ 2)) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = s[i];
}
int main() {
    const int n = 2048;
    std::vector<unsigned> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = 0x1000 + i;
    unsigned *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 8192, 0, dx, dout, n * 4);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 2; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                unsigned got = o[(w * 64 + l) * 4 + e];
                unsigned want = (l & 1) ? 0u : h[(w * 64 + (l ^ 2)) * 4 + e];
                if (got != want) { if (bad < 8) printf("w%d l%d e%d got %08x want %08x\n", w, l, e, got, want); ++bad; }
            }
    printf("lds-dma probe: %d mismatches (OOB lanes %s)\n", bad, o[4] == 0 ? "write zeros" : (o[4] == 0xAAAAAAAAu ? "leave LDS untouched" : "other"));
    printf("untouched region intact: %s\n", o[1000] == 0xAAAAAAAAu ? "yes" : "no");
    return bad != 0;
}
