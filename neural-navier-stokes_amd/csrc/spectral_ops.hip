// Standalone spectral operators on the LDS FFT engine (fft_lds.h), gfx950:
//   nns_spec_rfft2_f32 / nns_spec_irfft2_f32  -- numpy.fft.rfft2 / irfft2 layout and normalisation
//                                                (spec = interleaved complex64 [batch, nx, ny/2+1])
//   nns_spec_derivs_f32                       -- f_x, f_y, lap f of ONE real field (oracle/periodic.py: spectral_derivs)
// The residual kernels (spectral_kernels.hip) never materialise a 2-D spectrum; these entry points exist for
// callers that want the transform or the derivatives themselves (SURVEY.md section 8b lists them in the C ABI).
//
// Every kernel is "lines through LDS": a workgroup of 8 waves owns LINES = 8 * 64/(N/16) lines; lines along the
// contiguous axis are loaded straight into the FFT ownership pattern (lane t holds t + (N/16) m: coalesced);
// lines along the strided axis go through an LDS transpose stage (global row pieces <-> [line][row]).
#include "nns_common.h"
#include "fft_lds.h"
#include <type_traits>

using namespace nns;

namespace {

constexpr int kT = 512;                    // threads per workgroup
constexpr int kW = kT / kWave;

template <int N, typename TF>
struct OpsLds {
    static constexpr int TPF = N / 16, FPW = kWave / TPF, LINES = kW * FPW;
    static constexpr int SLOTS = N + N / 16;
    static constexpr int XB_BYTES = SLOTS * (int)sizeof(C2<TF>);
    static constexpr int STAGE_BYTES = (N + 16) * 8 + 128;                       // one complex64 (or two float) line + skew
    static constexpr int LINE_BYTES = ((XB_BYTES > STAGE_BYTES ? XB_BYTES : STAGE_BYTES) + 127) / 128 * 128;
    static constexpr int TAB_BYTES = (N / 2 + Pass2<N>::ENTRIES) * (int)sizeof(C2<TF>);
    static constexpr int TABI_BYTES = sizeof(TF) == 4 ? 0 : (N / 2 + Pass2<N>::ENTRIES) * (int)sizeof(C2<float>);
    static constexpr int TOTAL = TAB_BYTES + TABI_BYTES + LINES * LINE_BYTES;
    static constexpr int SKEW_MOD = LINES < 32 ? LINES : 32, SKEW_DW = 32 / SKEW_MOD;
};

template <int N, typename TF>
__device__ __forceinline__ void ops_setup(unsigned char* smem, C2<TF>*& tabF, C2<float>*& tabI, unsigned char*& lines) {
    using L = OpsLds<N, TF>;
    tabF = reinterpret_cast<C2<TF>*>(smem);
    fill_twiddles<TF, N>(tabF, threadIdx.x, kT);
    fill_twiddles2<TF, N>(tabF + N / 2, threadIdx.x, kT);
    if constexpr (sizeof(TF) == 4) tabI = reinterpret_cast<C2<float>*>(smem);
    else {
        tabI = reinterpret_cast<C2<float>*>(smem + L::TAB_BYTES);
        fill_twiddles<float, N>(tabI, threadIdx.x, kT);
        fill_twiddles2<float, N>(tabI + N / 2, threadIdx.x, kT);
    }
    lines = smem + L::TAB_BYTES + L::TABI_BYTES;
    __syncthreads();
}

// ---------------------------------------------------------------------------------- contiguous-axis kernels
// MODE 0: r2c   in = real rows [nrows][N]            out = complex [nrows][N/2+1]   (unnormalised forward)
// MODE 1: c2r   in = complex [nrows][N/2+1]          out = real rows [nrows][N] * scale   (Hermitian fill, inverse)
// MODE 2: deriv in = real rows, out0 = f_y (may be null), out1 += f_yy (lap accumulate, may be null)
template <int N, typename TF, int MODE>
__global__ __launch_bounds__(kT) void rows_kernel(const float* __restrict__ in, float* __restrict__ out0, float* __restrict__ out1,
                                                   long nrows, double kscale, float scale) {
    using L = OpsLds<N, TF>;
    constexpr int TPF = L::TPF, NH = N / 2 + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C2<TF>* tabF; C2<float>* tabI; unsigned char* lines;
    ops_setup<N, TF>(smem, tabF, tabI, lines);
    const long niter = (nrows + L::LINES - 1) / L::LINES;
    for (long it = blockIdx.x; it < niter; it += gridDim.x) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave, sub = lane / TPF, tid = lane % TPF;
        const int line = wave * L::FPW + sub;
        unsigned char* xb = lines + (size_t)line * L::LINE_BYTES;
        const long row_raw = it * L::LINES + line;
        const bool valid = row_raw < nrows;
        const long row = valid ? row_raw : nrows - 1;
        if constexpr (MODE == 1) {
            const float* src = in + (size_t)row * NH * 2;
            C2<float> z[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int e = tid + TPF * m;
                const int es = e <= N / 2 ? e : N - e;
                const float2 c = *reinterpret_cast<const float2*>(src + 2 * es);
                z[m].x = c.x * scale; z[m].y = (e <= N / 2 ? c.y : -c.y) * scale;
            }
            fft_line<float, N, true>(z, tabI, tabI + N / 2, reinterpret_cast<C2<float>*>(xb), tid);
            if (valid) {
#pragma unroll
                for (int m = 0; m < 16; ++m) out0[(size_t)row * N + tid + TPF * m] = z[m].x;
            }
        } else {
            C2<TF> z[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { z[m].x = (TF)in[(size_t)row * N + tid + TPF * m]; z[m].y = (TF)0; }
            fft_line<TF, N, false>(z, tabF, tabF + N / 2, reinterpret_cast<C2<TF>*>(xb), tid);
            if constexpr (MODE == 0) {
                if (valid) {
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const int e = tid + TPF * m;
                        if (e < NH) *reinterpret_cast<float2*>(out0 + ((size_t)row * NH + e) * 2) = make_float2((float)z[m].x, (float)z[m].y);
                    }
                }
            } else {
                int te = tid;
                asm volatile("" : "+v"(te), "+v"(z[0].x));
                // two separate inverses (f' and f'' differ by a factor ~k in magnitude: packing them into one complex
                // float32 transform would put f'' rounding noise on f')
                C2<float> c[16], d[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const int e = te + TPF * m;
                    const int kk = e < N / 2 ? e : e - N;
                    const TF k1 = (TF)((e == N / 2 ? 0.0 : (double)kk * kscale) / N);
                    const double kd = (double)kk * kscale;
                    const TF k2 = (TF)(-kd * kd / N);
                    c[m].x = (float)(-k1 * z[m].y); c[m].y = (float)(k1 * z[m].x);      // i k Z
                    d[m].x = (float)(k2 * z[m].x); d[m].y = (float)(k2 * z[m].y);       // -k^2 Z
                }
                fft_line<float, N, true>(c, tabI, tabI + N / 2, reinterpret_cast<C2<float>*>(xb), tid);
                __builtin_amdgcn_sched_barrier(0);
                fft_line<float, N, true>(d, tabI, tabI + N / 2, reinterpret_cast<C2<float>*>(xb), tid);
                if (valid) {
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const size_t g = (size_t)row * N + tid + TPF * m;
                        if (out0) out0[g] = c[m].x;
                        if (out1) out1[g] += d[m].x;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------- strided-axis kernels
// MODE 0: complex c2c in place along axis 0 of [N][ncols] complex64 (INV selects direction, unnormalised)
// MODE 2: deriv of a real field [N][ncols]: out0 = f_x (may be null), out1 = f_xx (lap, overwritten; may be null)
template <int N, typename TF, int MODE, bool INV>
__global__ __launch_bounds__(kT) void cols_kernel(const float* __restrict__ in, float* __restrict__ out0, float* __restrict__ out1,
                                                   int ncols, int tiles_per_grid, long ntiles, double kscale) {
    using L = OpsLds<N, TF>;
    constexpr int TPF = L::TPF, CW = L::LINES, RPI = kT / CW;
    constexpr int EPT = MODE == 0 ? 2 : 1;                      // floats per element in global memory
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C2<TF>* tabF; C2<float>* tabI; unsigned char* lines;
    ops_setup<N, TF>(smem, tabF, tabI, lines);
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int wave = tx / kWave, lane = tx % kWave, sub = lane / TPF, tid = lane % TPF;
        const int line = wave * L::FPW + sub;
        unsigned char* xb = lines + (size_t)line * L::LINE_BYTES;
        float* mine = reinterpret_cast<float*>(xb) + (line % L::SKEW_MOD) * L::SKEW_DW;
        const int cc = tx % CW, cr = tx / CW;
        float* cp = reinterpret_cast<float*>(lines + (size_t)cc * L::LINE_BYTES) + (cc % L::SKEW_MOD) * L::SKEW_DW;
        const long lt = xcd_remap((unsigned)t, (unsigned)ntiles);
        const int j0 = (int)(lt % tiles_per_grid) * CW;
        const size_t g = (size_t)(lt / tiles_per_grid) * N * ncols * EPT;
        const bool ok = j0 + cc < ncols;
        for (int r = cr; r < N; r += RPI) {
            const size_t c = g + ((size_t)r * ncols + j0 + cc) * EPT;
            if constexpr (MODE == 0) { const float2 v = ok ? *reinterpret_cast<const float2*>(in + c) : make_float2(0.f, 0.f); cp[2 * r] = v.x; cp[2 * r + 1] = v.y; }
            else cp[r] = ok ? in[c] : 0.f;
        }
        __syncthreads();
        int tv = tid;
        asm volatile("" : "+v"(tv));
        C2<float> res[16];
        if constexpr (MODE == 0) {
            C2<float> z[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { z[m].x = mine[2 * (tv + TPF * m)]; z[m].y = mine[2 * (tv + TPF * m) + 1]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            fft_line<float, N, INV>(z, tabI, tabI + N / 2, reinterpret_cast<C2<float>*>(xb), tv);
#pragma unroll
            for (int m = 0; m < 16; ++m) res[m] = z[m];
        } else {
            C2<TF> z[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { z[m].x = (TF)mine[tv + TPF * m]; z[m].y = (TF)0; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            fft_line<TF, N, false>(z, tabF, tabF + N / 2, reinterpret_cast<C2<TF>*>(xb), tv);
            int te = tv;
            asm volatile("" : "+v"(te), "+v"(z[0].x));
            C2<float> d[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int e = te + TPF * m;
                const int kk = e < N / 2 ? e : e - N;
                const TF k1 = (TF)((e == N / 2 ? 0.0 : (double)kk * kscale) / N);
                const double kd = (double)kk * kscale;
                const TF k2 = (TF)(-kd * kd / N);
                res[m].x = (float)(-k1 * z[m].y); res[m].y = (float)(k1 * z[m].x);
                d[m].x = (float)(k2 * z[m].x); d[m].y = (float)(k2 * z[m].y);
            }
            fft_line<float, N, true>(res, tabI, tabI + N / 2, reinterpret_cast<C2<float>*>(xb), tv);
            __builtin_amdgcn_sched_barrier(0);
            fft_line<float, N, true>(d, tabI, tabI + N / 2, reinterpret_cast<C2<float>*>(xb), tv);
#pragma unroll
            for (int m = 0; m < 16; ++m) res[m].y = d[m].x;                  // (f', f'') for the staged write-back
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 16; ++m) { mine[2 * (tv + TPF * m)] = res[m].x; mine[2 * (tv + TPF * m) + 1] = res[m].y; }
        __syncthreads();
        if (ok) {
            for (int r = cr; r < N; r += RPI) {
                const size_t c = g + ((size_t)r * ncols + j0 + cc) * EPT;
                if constexpr (MODE == 0) *reinterpret_cast<float2*>(out0 + c) = make_float2(cp[2 * r], cp[2 * r + 1]);
                else { if (out0) out0[c] = cp[2 * r]; if (out1) out1[c] = cp[2 * r + 1]; }
            }
        }
        __syncthreads();
    }
}

template <typename K>
int set_lds(K kern, int bytes, const char* what) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "%s: hipFuncSetAttribute(%d B): %s", what, bytes, hipGetErrorString(e));
    return NNS_OK;
}

inline bool pow2ok(int n) { return n >= 64 && n <= 1024 && (n & (n - 1)) == 0; }

template <typename F>
int dispatch(int n, F&& f) {
    switch (n) {
        case 64: return f(std::integral_constant<int, 64>{});
        case 128: return f(std::integral_constant<int, 128>{});
        case 256: return f(std::integral_constant<int, 256>{});
        case 512: return f(std::integral_constant<int, 512>{});
        case 1024: return f(std::integral_constant<int, 1024>{});
    }
    return fail(NNS_ERR_UNSUPPORTED, "spectral op: axis length %d is not a power of two in [64, 1024]", n);
}

template <int N, typename TF, int MODE>
int launch_rows(const float* in, float* o0, float* o1, long nrows, double kscale, float scale, hipStream_t s) {
    using L = OpsLds<N, TF>;
    auto kern = rows_kernel<N, TF, MODE>;
    if (int rc = set_lds(kern, L::TOTAL, "spectral rows")) return rc;
    const long niter = (nrows + L::LINES - 1) / L::LINES;
    hipLaunchKernelGGL(kern, dim3((unsigned)(niter < 2048 ? niter : 2048)), dim3(kT), L::TOTAL, s, in, o0, o1, nrows, kscale, scale);
    return check_launch("spectral rows");
}

template <int N, typename TF, int MODE, bool INV>
int launch_cols(const float* in, float* o0, float* o1, int batch, int ncols, double kscale, hipStream_t s) {
    using L = OpsLds<N, TF>;
    auto kern = cols_kernel<N, TF, MODE, INV>;
    if (int rc = set_lds(kern, L::TOTAL, "spectral cols")) return rc;
    const int tpg = (ncols + L::LINES - 1) / L::LINES;
    const long ntiles = (long)batch * tpg;
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < 2048 ? ntiles : 2048)), dim3(kT), L::TOTAL, s, in, o0, o1, ncols, tpg, ntiles, kscale);
    return check_launch("spectral cols");
}

}  // namespace

#define S(stream) reinterpret_cast<hipStream_t>(stream)

NNS_API int nns_spec_rfft2_f32(const float* f, float* spec, int batch, int nx, int ny, void* stream) {
    if (!f || !spec || batch < 1) return fail(NNS_ERR_INVALID_ARG, "spec_rfft2: bad args");
    if (!pow2ok(nx) || !pow2ok(ny)) return fail(NNS_ERR_UNSUPPORTED, "spec_rfft2: nx=%d, ny=%d must be powers of two in [64, 1024]", nx, ny);
    const int nh = ny / 2 + 1;
    int rc = dispatch(ny, [&](auto n) { return launch_rows<decltype(n)::value, float, 0>(f, spec, nullptr, (long)batch * nx, 0.0, 1.f, S(stream)); });
    if (rc) return rc;
    return dispatch(nx, [&](auto n) { return launch_cols<decltype(n)::value, float, 0, false>(spec, spec, nullptr, batch, nh, 0.0, S(stream)); });
}

// spec is used as scratch for the column pass (it is overwritten).
NNS_API int nns_spec_irfft2_f32(float* spec, float* f, int batch, int nx, int ny, void* stream) {
    if (!f || !spec || batch < 1) return fail(NNS_ERR_INVALID_ARG, "spec_irfft2: bad args");
    if (!pow2ok(nx) || !pow2ok(ny)) return fail(NNS_ERR_UNSUPPORTED, "spec_irfft2: nx=%d, ny=%d must be powers of two in [64, 1024]", nx, ny);
    const int nh = ny / 2 + 1;
    int rc = dispatch(nx, [&](auto n) { return launch_cols<decltype(n)::value, float, 0, true>(spec, spec, nullptr, batch, nh, 0.0, S(stream)); });
    if (rc) return rc;
    const float scale = (float)(1.0 / ((double)nx * ny));
    return dispatch(ny, [&](auto n) { return launch_rows<decltype(n)::value, float, 1>(spec, f, nullptr, (long)batch * nx, 0.0, scale, S(stream)); });
}

NNS_API int nns_spec_derivs_f32(const float* f, float* f_x, float* f_y, float* f_lap, int batch, int nx, int ny, double Lx, double Ly,
                                int precise, void* stream) {
    if (!f || batch < 1 || Lx == 0 || Ly == 0) return fail(NNS_ERR_INVALID_ARG, "spec_derivs: bad args");
    if (!pow2ok(nx) || !pow2ok(ny)) return fail(NNS_ERR_UNSUPPORTED, "spec_derivs: nx=%d, ny=%d must be powers of two in [64, 1024]", nx, ny);
    if (!f_x && !f_y && !f_lap) return NNS_OK;
    hipStream_t s = S(stream);
    const double kx = 2.0 * M_PI / Lx, ky = 2.0 * M_PI / Ly;
    int rc = NNS_OK;
    if (f_x || f_lap) {
        rc = dispatch(nx, [&](auto n) {
            constexpr int N = decltype(n)::value;
            return precise ? launch_cols<N, double, 2, false>(f, f_x, f_lap, batch, ny, kx, s) : launch_cols<N, float, 2, false>(f, f_x, f_lap, batch, ny, kx, s);
        });
        if (rc) return rc;
    }
    if (f_y || f_lap) {
        rc = dispatch(ny, [&](auto n) {
            constexpr int N = decltype(n)::value;
            return precise ? launch_rows<N, double, 2>(f, f_y, f_lap, (long)batch * nx, ky, 1.f, s) : launch_rows<N, float, 2>(f, f_y, f_lap, (long)batch * nx, ky, 1.f, s);
        });
    }
    return rc;
}
