// Practical MFMA peak on this part: nothing but v_mfma_f32_32x32x16_bf16 on register operands (8 independent accumulators
// per wave, 2 waves per SIMD, every CU), random bf16 operand bits so that the power draw is representative.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512, 2) void peak(const bf16x8* in, float* out, int iters) {
    bf16x8 a[2], b[4];
    for (int i = 0; i < 2; i++) a[i] = in[(threadIdx.x * 6 + i) % 4096];
    for (int i = 0; i < 4; i++) b[i] = in[(threadIdx.x * 6 + 2 + i) % 4096];
    f32x16 acc[8];
    for (int i = 0; i < 8; i++)
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 9; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i >> 2], b[i & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++)
        for (int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = 4000;
    bf16x8* in; float* out;
    hipMalloc(&in, 4096 * 16); hipMalloc(&out, blocks * 512 * 4);
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; i++) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.01f; unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(peak, dim3(blocks), dim3(512), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 8 * iters * 72 * 32768.0;
        printf("blocks %d: %.3f ms  %.1f TFLOP/s  (%.0f MHz-equivalent at 1024 flop/clk/CU-SIMD... %.3f GHz)\n", blocks, ms, fl / ms / 1e9,
               0.0, fl / ms / 1e6 / (256.0 * 4 * 32768 / 32 * 1.0) / 1e3);
    }
    return 0;
}
