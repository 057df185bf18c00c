// Developer micro-benchmark: what HBM rate does a pure streaming kernel reach on gfx950 as a function of the NUMBER of read and write streams?
// out_w[i] = sum_r in_r[i] * c_w (float4 per lane, grid-stride), arrays of 64 x 1024^2 floats (268 MB each, >> the 256 MB Infinity Cache in total).
// The residual kernels of this library move 5 + 3 (fd_residual), 8 + 3 (spectral row pass) and 8 + 6 (fused marching row pass) streams at
// 4.9 - 5.1 TB/s; a 1 + 1 copy reaches 6.0 - 6.3.  Build: hipcc -O3 --offload-arch=gfx950 tools/streams_bench.hip -o tools/streams_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Ptrs { const float4* in[8]; float4* out[8]; };

template <int NR, int NW, int UNR>
__global__ __launch_bounds__(256) void streams(Ptrs p, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride * UNR) {
        float4 v[UNR][NR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int r = 0; r < NR; ++r) { const size_t j = i + u * stride; v[u][r] = p.in[r][j < n4 ? j : i]; }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float4 s = v[u][0];
#pragma unroll
            for (int r = 1; r < NR; ++r) { s.x += v[u][r].x; s.y += v[u][r].y; s.z += v[u][r].z; s.w += v[u][r].w; }
            const size_t j = i + u * stride;
            if (j < n4) {
#pragma unroll
                for (int w = 0; w < NW; ++w) p.out[w][j] = make_float4(s.x * (w + 1), s.y * (w + 1), s.z * (w + 1), s.w * (w + 1));
            }
        }
    }
}

template <int NR, int NW, int UNR>
void run(const Ptrs& p, size_t n4, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((streams<NR, NW, UNR>), dim3(blocks), dim3(256), 0, 0, p, n4);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((streams<NR, NW, UNR>), dim3(blocks), dim3(256), 0, 0, p, n4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)(NR + NW) * n4 * 16;
    printf("reads %d writes %d unroll %d blocks %5d : %.3f ms  %.2f TB/s\n", NR, NW, UNR, blocks, ms, bytes / ms / 1e9);
}

int main() {
    const size_t n4 = (size_t)64 * 1024 * 1024 / 4;
    Ptrs p;
    for (int r = 0; r < 8; ++r) { float4* q; CK(hipMalloc(&q, n4 * 16)); CK(hipMemset(q, 0, n4 * 16)); p.in[r] = q; }
    for (int w = 0; w < 8; ++w) { float4* q; CK(hipMalloc(&q, n4 * 16)); p.out[w] = q; }
    for (int blocks : {2048, 8192}) {
        run<1, 1, 4>(p, n4, blocks); run<2, 1, 4>(p, n4, blocks); run<5, 3, 2>(p, n4, blocks); run<5, 3, 4>(p, n4, blocks);
        run<8, 3, 2>(p, n4, blocks); run<8, 6, 2>(p, n4, blocks); run<8, 6, 1>(p, n4, blocks); run<4, 4, 2>(p, n4, blocks); run<8, 1, 2>(p, n4, blocks); run<1, 8, 4>(p, n4, blocks);
    }
    return 0;
}
