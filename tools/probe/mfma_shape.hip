// Which fp16 MFMA shape does the chip sustain the higher clock on?  (MI355X_MICROARCH.md, DVFS give-back item 7: on random
// data a 16x16x32 loop delivered ~1.15x the FLOP/s of a 32x32x16 loop at equal cycles per FLOP.)  Same output tile per wave
// (128 x 64), same three partial products per k-step as gemm_planes.h, 8 waves per CU (2 per SIMD) or 4 (1 per SIMD).
//   SHAPE 0: v_mfma_f32_32x32x16_f16, 4 x 2 tiles, 2 k16 steps per "k-tile" = 48 MFMAs of 32 cycles
//   SHAPE 1: v_mfma_f32_16x16x32_f16, 8 x 4 tiles, 1 k32 step per "k-tile" = 96 MFMAs of 16 cycles
//   LDS 0: operands stay in registers; LDS 1: every fragment is re-read from LDS (ds_read_b128) each k-tile, like the GEMM
// Interleaved rounds in one process, random operand bits.   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int LDS, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k(float* out, int iters, unsigned seed) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned h = (tid + blockIdx.x * 977u) * 2654435761u + seed;
    for (int i = tid; i < 65536 / 4; i += NT) { h = h * 1664525u + 1013904223u;
        // two random fp16 in [-1, 1)
        const _Float16 a = (_Float16)(((int)(h >> 16) - 32768) * (1.f / 32768.f));
        const _Float16 b = (_Float16)(((int)(h & 0xffff) - 32768) * (1.f / 32768.f));
        ((unsigned*)smem)[i] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16); }
    __syncthreads();
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int li = lane & 31, lh = lane >> 5, swz = (li >> 1) & 7;
        unsigned fa[2][2], fb[2][2];
        for (int pl = 0; pl < 2; ++pl) for (int st = 0; st < 2; ++st) {
            const int ch = ((4 * pl + 2 * st + lh) ^ swz) << 4;
            fa[pl][st] = li * 128 + ch; fb[pl][st] = 32768 + li * 128 + ch; }
        f16x8 ah[4], al[4], bh[2], bl[2];
        for (int i = 0; i < 4; ++i) { ah[i] = *(const f16x8*)(smem + fa[0][0] + i * 4096); al[i] = *(const f16x8*)(smem + fa[1][0] + i * 4096); }
        for (int j = 0; j < 2; ++j) { bh[j] = *(const f16x8*)(smem + fb[0][0] + j * 4096); bl[j] = *(const f16x8*)(smem + fb[1][0] + j * 4096); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (LDS) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) { bh[j] = *(const f16x8*)(smem + fb[0][st] + j * 4096); bl[j] = *(const f16x8*)(smem + fb[1][st] + j * 4096); }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (LDS) { ah[i] = *(const f16x8*)(smem + fa[0][st] + i * 4096); al[i] = *(const f16x8*)(smem + fa[1][st] + i * 4096); }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
                }
            }
            asm volatile("" ::: "memory");
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else {
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        const int l15 = lane & 15, lq = lane >> 4, swz = (l15 >> 1) & 7;
        unsigned fa[2], fb[2];
        for (int pl = 0; pl < 2; ++pl) { const int ch = ((4 * pl + lq) ^ swz) << 4; fa[pl] = l15 * 128 + ch; fb[pl] = 32768 + l15 * 128 + ch; }
        f16x8 ah[8], al[8], bh[4], bl[4];
        for (int i = 0; i < 8; ++i) { ah[i] = *(const f16x8*)(smem + fa[0] + i * 2048); al[i] = *(const f16x8*)(smem + fa[1] + i * 2048); }
        for (int j = 0; j < 4; ++j) { bh[j] = *(const f16x8*)(smem + fb[0] + j * 2048); bl[j] = *(const f16x8*)(smem + fb[1] + j * 2048); }
        for (int it = 0; it < iters; ++it) {
            if (LDS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { bh[j] = *(const f16x8*)(smem + fb[0] + j * 2048); bl[j] = *(const f16x8*)(smem + fb[1] + j * 2048); }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (LDS) { ah[i] = *(const f16x8*)(smem + fa[0] + i * 2048); al[i] = *(const f16x8*)(smem + fa[1] + i * 2048); }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], c, 0, 0, 0);
                    acc[i][j] = c;
                }
            }
            asm volatile("" ::: "memory");
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    if (s == 12345.678f) out[0] = s;
}

struct Var { const char* name; void (*fn)(float*, int, unsigned); int nt; };
template <int SHAPE, int LDS, int NT> void launch(float* d, int iters, unsigned seed) { k<SHAPE, LDS, NT><<<256, NT>>>(d, iters, seed); }

int main() {
    float* d; hipMalloc(&d, 4);
    Var vars[] = {
        {"32x32x16 regs  8 waves", launch<0, 0, 512>, 512}, {"16x16x32 regs  8 waves", launch<1, 0, 512>, 512},
        {"32x32x16 LDS   8 waves", launch<0, 1, 512>, 512}, {"16x16x32 LDS   8 waves", launch<1, 1, 512>, 512},
        {"32x32x16 LDS   4 waves", launch<0, 1, 256>, 256}, {"16x16x32 LDS   4 waves", launch<1, 1, 256>, 256},
    };
    const int nv = sizeof(vars) / sizeof(vars[0]), iters = 6000, rounds = 7;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<std::vector<float>> ms(nv);
    for (int w = 0; w < 2; ++w) for (int v = 0; v < nv; ++v) vars[v].fn(d, iters, 1u);      // warm up (clocks settle)
    hipDeviceSynchronize();
    for (int r = 0; r < rounds; ++r)
        for (int v = 0; v < nv; ++v) {
            hipEventRecord(e0); vars[v].fn(d, iters, 7u + r); hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); ms[v].push_back(t);
        }
    for (int v = 0; v < nv; ++v) {
        std::sort(ms[v].begin(), ms[v].end());
        const double fl = 256.0 * (vars[v].nt / 64) * iters * 48 * 32768.0;      // 48 MFMAs of 32x32x16 (= 96 of 16x16x32) per iteration
        const double med = ms[v][rounds / 2], mn = ms[v][0];
        printf("%-24s median %.3f ms %.1f TF fp16 (= %.1f TF of fp16x3 products)   best %.1f TF\n", vars[v].name, med, fl / med * 1e-9,
               fl / med * 1e-9 / 3, fl / mn * 1e-9);
    }
    return 0;
}
