// HBM streaming rates on this part by access mix (16 bytes per lane): pure read, pure write, copy (1R+1W), 2R+1W (gn_silu_add),
// 3R+1W (pw_res), 3R (pw_res with the fused final conv).  Buffers are 268 MB each (one 64-channel bf16 tensor at B = 32, 256x256),
// larger than the 256 MB Infinity Cache together.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NR, int NW> __global__ __launch_bounds__(256) void stream(const f4* a, const f4* b, const f4* c, f4* o, size_t n, float* sink) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4 v = {1, 2, 3, 4};
        if (NR >= 1) v = a[i];
        if (NR >= 2) v += b[i];
        if (NR >= 3) v += c[i];
        if (NW) o[i] = v; else acc += v;
    }
    if (!NW && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
int main() {
    const size_t bytes = 268435456, n = bytes / 16;
    f4 *a, *b, *c, *o; float* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 1, bytes); hipMemset(c, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {2048, 8192, 65536};
    for (int gi = 0; gi < 3; gi++) {
        const int g = grids[gi];
#define RUN(NR, NW, name)                                                                             \
    {                                                                                                 \
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL((stream<NR, NW>), dim3(g), dim3(256), 0, 0, a, b, c, o, n, sink); \
        hipEventRecord(e0);                                                                           \
        for (int r = 0; r < 10; r++) hipLaunchKernelGGL((stream<NR, NW>), dim3(g), dim3(256), 0, 0, a, b, c, o, n, sink); \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                  \
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;                                         \
        printf("grid %6d %-8s %7.1f us  %6.2f TB/s\n", g, name, ms * 1e3, (double)(NR + NW) * bytes / ms / 1e9); \
    }
        RUN(1, 0, "1R") RUN(3, 0, "3R") RUN(0, 1, "1W") RUN(1, 1, "1R+1W") RUN(2, 1, "2R+1W") RUN(3, 1, "3R+1W")
    }
    return 0;
}
