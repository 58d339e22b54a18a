#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __amdgpu_buffer_rsrc_t srd_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int nvalid) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 7.0f;
    __syncthreads();
    srd_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nvalid * 16, 0x00020000);
    unsigned off = threadIdx.x * 16;             // lanes >= nvalid are out of range
    if (threadIdx.x & 1) off = 0x80000000u;      // odd lanes: far out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    float *src, *out, h[512], hs[256];
    for (int i = 0; i < 256; ++i) hs[i] = 100.f + i;
    hipMalloc(&src, 1024); hipMalloc(&out, 2048);
    hipMemcpy(src, hs, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 2048, 0, src, out, 48);
    hipMemcpy(h, out, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %g %g %g %g\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    return 0;
}
