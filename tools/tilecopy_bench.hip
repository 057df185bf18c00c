// Developer micro-benchmark (not part of the product): how fast can ONE 512-thread workgroup per CU move
// C-column x 1024-row tiles of three fp32 arrays (in -> registers -> out), as the spectral x-pass does?
//   C  = columns per row piece (8, 16, 32)      W = floats per lane (1, 2, 4)
//   ALU = dependent-FMA iterations between the load burst and the store burst (emulates the transforms)
//   SPREAD = 1: the stores of the previous tile and loads of the next are issued in small groups inside the ALU phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { using T = float __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {   // as nns_common.h
    const unsigned xcd = bid % 8, q = nblocks / 8, r = nblocks % 8;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / 8;
}
__device__ __forceinline__ float burn(float x, int n) {
    for (int i = 0; i < n; ++i) x = fmaf(x, 1.0000001f, 1e-9f);
    return x;
}

template <int C, int W, int SPREAD, int MODE = 0>
__global__ __launch_bounds__(512) void tile_copy(const float* __restrict__ a0, const float* __restrict__ a1, const float* __restrict__ a2,
                                                 float* __restrict__ b0, float* __restrict__ b1, float* __restrict__ b2,
                                                 int ny, int tiles_per_grid, long ntiles, int alu) {
    using V = typename Vec<W>::T;
    extern __shared__ unsigned char smem[];
    constexpr int LPR = C / W;                 // lanes per row piece
    constexpr int RPI = 512 / LPR;             // rows per instruction across the workgroup
    constexpr int NI = 1024 / RPI;             // instructions per field and thread
    const int cc = (threadIdx.x % LPR) * W, cr = threadIdx.x / LPR;
    V r0[NI], r1[NI], r2[NI];
    float acc = threadIdx.x;
    for (long tt = blockIdx.x; tt < ntiles; tt += gridDim.x) {
        const long t = xcd_remap((unsigned)tt, (unsigned)ntiles);
        const int j0 = (int)(t % tiles_per_grid) * C;
        const size_t g = (size_t)(t / tiles_per_grid) * 1024 * ny;
        if (!SPREAD) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const size_t c = g + (size_t)(cr + RPI * i) * ny + j0 + cc;
                if (MODE != 1) { r0[i] = *(const V*)(a0 + c); r1[i] = *(const V*)(a1 + c); r2[i] = *(const V*)(a2 + c); }
                else { r0[i] = (V)(acc + i); r1[i] = r0[i]; r2[i] = r0[i]; }
            }
            acc = burn(acc, alu);
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < NI; ++i) acc += ((const float*)&r0[i])[0] + ((const float*)&r1[i])[0] + ((const float*)&r2[i])[0];
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const size_t c = g + (size_t)(cr + RPI * i) * ny + j0 + cc;
                    *(V*)(b0 + c) = r0[i]; *(V*)(b1 + c) = r1[i]; *(V*)(b2 + c) = r2[i];
                }
            }
        } else {
            // same traffic, but each load group is followed by 1/NI of the ALU work and then its store group
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const size_t c = g + (size_t)(cr + RPI * i) * ny + j0 + cc;
                r0[i] = *(const V*)(a0 + c); r1[i] = *(const V*)(a1 + c); r2[i] = *(const V*)(a2 + c);
                acc = burn(acc, alu / NI);
                if (i > 0) {
                    const size_t cp = g + (size_t)(cr + RPI * (i - 1)) * ny + j0 + cc;
                    *(V*)(b0 + cp) = r0[i - 1]; *(V*)(b1 + cp) = r1[i - 1]; *(V*)(b2 + cp) = r2[i - 1];
                }
            }
            const size_t cp = g + (size_t)(cr + RPI * (NI - 1)) * ny + j0 + cc;
            *(V*)(b0 + cp) = r0[NI - 1]; *(V*)(b1 + cp) = r1[NI - 1]; *(V*)(b2 + cp) = r2[NI - 1];
        }
    }
    if (acc == 12345.678f) smem[0] = 1;
}

template <int C, int W, int SPREAD, int MODE = 0>
void run(const float* a, float* b, int batch, int ny, int alu, int grid) {
    const size_t fld = (size_t)batch * 1024 * ny;
    const int tpg = ny / C; const long ntiles = (long)batch * tpg;
    auto k = tile_copy<C, W, SPREAD, MODE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it)
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 150 * 1024, 0, a, a + fld, a + 2 * fld, b, b + fld, b + 2 * fld, ny, tpg, ntiles, alu);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        if (rep == 1) printf("C=%2d W=%d spread=%d mode=%d alu=%5d grid=%4d : %.3f ms  %.2f TB/s\n", C, W, SPREAD, MODE, alu, grid, ms, (MODE ? 3.0 : 6.0) * fld * 4 / ms * 1e-9);
    }
}

int main() {
    const int batch = 64, ny = 1024;
    const size_t fld = (size_t)batch * 1024 * ny;
    float *a, *b; CK(hipMalloc(&a, 3 * fld * 4)); CK(hipMalloc(&b, 3 * fld * 4));
    CK(hipMemset(a, 0, 3 * fld * 4)); CK(hipMemset(b, 0, 3 * fld * 4));
    for (int grid : {256, 2048}) {
        run<8, 1, 0>(a, b, batch, ny, 0, grid); run<8, 2, 0>(a, b, batch, ny, 0, grid); run<8, 4, 0>(a, b, batch, ny, 0, grid);
        run<16, 1, 0>(a, b, batch, ny, 0, grid); run<16, 4, 0>(a, b, batch, ny, 0, grid);
        run<32, 1, 0>(a, b, batch, ny, 0, grid); run<32, 4, 0>(a, b, batch, ny, 0, grid);
    }
    // store-only and load-only bursts (mode 1 / 2): which lane width suits the 32-byte row pieces
    run<8, 1, 0, 1>(a, b, batch, ny, 0, 2048); run<8, 2, 0, 1>(a, b, batch, ny, 0, 2048); run<8, 4, 0, 1>(a, b, batch, ny, 0, 2048);
    run<8, 1, 0, 2>(a, b, batch, ny, 0, 2048); run<8, 2, 0, 2>(a, b, batch, ny, 0, 2048); run<8, 4, 0, 2>(a, b, batch, ny, 0, 2048);
    run<16, 4, 0, 1>(a, b, batch, ny, 0, 2048); run<16, 4, 0, 2>(a, b, batch, ny, 0, 2048);
    return 0;
    // with an ALU phase of about the transforms' length: burst vs spread
    for (int alu : {4000, 8000, 16000}) {
        run<8, 1, 0>(a, b, batch, ny, alu, 256); run<8, 1, 1>(a, b, batch, ny, alu, 256);
        run<8, 4, 0>(a, b, batch, ny, alu, 256); run<8, 4, 1>(a, b, batch, ny, alu, 256);
        run<32, 4, 0>(a, b, batch, ny, alu, 256); run<32, 4, 1>(a, b, batch, ny, alu, 256);
    }
    return 0;
}
