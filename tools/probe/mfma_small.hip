// Issue cost of the small MFMA shapes the attention kernels can use, on gfx950: shader cycles per instruction for a chain of
// DEPENDENT MFMAs on one accumulator and for 4 independent accumulators, one wave per SIMD (4 waves per workgroup, one workgroup per CU).
//   v_mfma_f32_16x16x4_f32   (what attention.h used through round 3: exact fp32 products)
//   v_mfma_f32_16x16x16_f16  (legacy k = 16 fp16 form)
//   v_mfma_f32_16x16x32_f16  (gfx950 k = 32 form)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_small mfma_small.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a32 = 1.0f + lane * 1e-3f, b32 = 0.5f - lane * 1e-3f;
    f16x4 a4, b4; f16x8 a8, b8;
    for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)(0.01f * (lane + i)); b4[i] = (_Float16)(0.02f * (lane - i)); }
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (lane + i)); b8[i] = (_Float16)(0.02f * (lane - i)); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, acc[i], 0, 0, 0);
                else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
            }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int NACC> void run(const char* name, float* d, unsigned long long* c) {
    const int iters = 2000;
    k<KIND, NACC><<<256, 256>>>(d, c, iters); hipDeviceSynchronize();
    k<KIND, NACC><<<256, 256>>>(d, c, iters); hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s %d accumulator(s): %.1f memtime ticks per MFMA\n", name, NACC, (double)h / (iters * 8.0 * NACC));
}
int main() {
    float* d; hipMalloc(&d, 4); unsigned long long* c; hipMalloc(&c, 8);
    run<0, 1>("v_mfma_f32_16x16x4_f32", d, c);  run<0, 4>("v_mfma_f32_16x16x4_f32", d, c);
    run<1, 1>("v_mfma_f32_16x16x16_f16", d, c); run<1, 4>("v_mfma_f32_16x16x16_f16", d, c);
    run<2, 1>("v_mfma_f32_16x16x32_f16", d, c); run<2, 4>("v_mfma_f32_16x16x32_f16", d, c);
    return 0;
}
