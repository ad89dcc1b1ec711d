// Micro-probe (development aid): semantics of ds_read_b64_tr_b16 and of an out-of-range LDS-DMA on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void tr_kernel(int* out) {
    __shared__ __attribute__((aligned(16))) short sm[64 * 64];        // [row][col] 64 x 64 shorts, value = row * 100 + col
    for (int i = threadIdx.x; i < 64 * 64; i += 64) sm[i] = (short)((i / 64) * 100 + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, q = (l >> 2) & 3, p = l & 3;
    // group g: block rows 4g..4g+3 (row q), columns 16g.. : lane supplies row q, columns 4p..4p+3
    const int row = 4 * g + q, col = 16 * g + 4 * p;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(sm + row * 64 + col));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
__global__ void dma_kernel(const float* src, int nbytes, float* out) {
    __shared__ __attribute__((aligned(16))) float sm[256];
    for (int i = threadIdx.x; i < 256; i += 64) sm[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)sm, 16, threadIdx.x * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = sm[i];
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    tr_kernel<<<1, 64>>>(d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    float *s, *o; hipMalloc(&s, 1024); hipMalloc(&o, 1024);
    float hs[256]; for (int i = 0; i < 256; ++i) hs[i] = (float)i;
    hipMemcpy(s, hs, 1024, hipMemcpyHostToDevice);
    dma_kernel<<<1, 64>>>(s, 512, o);          // only the first 512 bytes are in range
    float ho[256]; hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    printf("dma in-range [0]=%g [127]=%g ; out-of-range [128]=%g [255]=%g (-7 = not written, 0 = zero-filled)\n", ho[0], ho[127], ho[128], ho[255]);
    return 0;
}
