// Read rate of a GEMM-operand-shaped sweep by the size of the contiguous run a wave instruction takes from each row:
// a [M][ROWB bytes] table is read tile by tile (128 rows per block), K step by K step; one wave instruction (64 lanes x 16 B)
// covers 64 / SEGL rows x SEGL * 16 bytes.  SEGL = 4 is gemm_nt_dma's BK = 32 bf16 (64-byte runs), 8 = BK 64, 16 = BK 128,
// ROWB / 16 = whole rows.  Same bytes, same vectors in flight per thread (8), only the order differs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int SEGL, int ROWB> __global__ __launch_bounds__(256, 3) void sweep(const unsigned char* a, int mtiles, float* sink, int reps) {
    constexpr int NV = 128 * ROWB / 16, PER = NV / 256;
    f4 acc = {0, 0, 0, 0};
    for (int rep = 0; rep < reps; rep++)
    for (int mt = blockIdx.x; mt < mtiles; mt += gridDim.x) {
        const unsigned char* base = a + (size_t)mt * 128 * ROWB;
#pragma unroll
        for (int r0 = 0; r0 < PER; r0 += 8) {
            f4 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = threadIdx.x + 256 * (r0 + j);
                const int piece = i % SEGL, row = (i / SEGL) % 128, step = i / (SEGL * 128);
                v[j] = *reinterpret_cast<const f4*>(base + (size_t)row * ROWB + (step * SEGL + piece) * 16);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) acc += v[j];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
static int g_reps = 1;
template <int SEGL, int ROWB> void run(const unsigned char* a, size_t bytes, float* sink, int grid) {
    const int mtiles = (int)(bytes / ((size_t)128 * ROWB));
    const int g = grid ? grid : mtiles;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((sweep<SEGL, ROWB>), dim3(g), dim3(256), 0, 0, a, mtiles, sink, g_reps);
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL((sweep<SEGL, ROWB>), dim3(g), dim3(256), 0, 0, a, mtiles, sink, g_reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("rowbytes %4d run %4d B grid %6d: %7.1f us %6.2f TB/s\n", ROWB, SEGL * 16, g, ms * 1e3, (double)g_reps * mtiles * 128 * ROWB / ms / 1e9);
}
int main(int argc, char** argv) {
    // default 768 MB: three times the Infinity Cache (HBM rate); a 12 MB table is L2-resident per XCD after the warm-up launches
    // (grid-stride over tiles: XCD x keeps re-reading the same eighth) and shows the rate of the L1 front end alone
    if (argc > 2) g_reps = atoi(argv[2]);  // every block re-reads its tiles: L2 hits after the first pass, L1 (32 KB) cannot hold a 96 KB tile
    const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) * 786432 : 805306368;
    unsigned char* a; float* sink;
    hipMalloc(&a, bytes); hipMalloc(&sink, 4); hipMemset(a, 1, bytes);
    for (int grid : {0, 768}) {
        run<4, 768>(a, bytes, sink, grid); run<8, 768>(a, bytes, sink, grid); run<16, 768>(a, bytes, sink, grid); run<48, 768>(a, bytes, sink, grid);
        run<4, 1024>(a, bytes, sink, grid); run<8, 1024>(a, bytes, sink, grid); run<16, 1024>(a, bytes, sink, grid); run<64, 1024>(a, bytes, sink, grid);
        run<4, 512>(a, bytes, sink, grid); run<8, 512>(a, bytes, sink, grid); run<32, 512>(a, bytes, sink, grid);
    }
    return 0;
}
