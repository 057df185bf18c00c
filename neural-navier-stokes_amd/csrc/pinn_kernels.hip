// The loss head of the physics-informed training step (hypothesis:
// src/neural_spectral/derivations/derivation.tex:25-34, "Neural Residual PDEs" -- stated by the reference, never implemented):
//
//     pred  = state + mlp_out                                  (channels u, v, p)
//     data  = mean (pred - target)^2                           over all 3 batch npix values
//     phys  = mean r_u^2 + mean r_v^2 + w_div mean r_div^2     of the Navier-Stokes residual of pred (nns_fd_residual_* / nns_spec_residual_*)
//     total = data + lam phys
//
// Round 3 left everything between the MLP and the residual kernels to tensor ops: ~60 launches of 5-15 us per step, a third of the
// 1.75 ms step at 16 x 512^2 pixels (profiles/r04_pinn_glue.txt).  Here it is three HBM-bound passes:
//   * assemble: pred's channels as the contiguous fields the residual kernels take (+ u_prev, v_prev out of a [B, 3, npix] state),
//     and the data term's sum of squares                                                   -- 36 B read + 12 (20) written per pixel
//   * loss:     the three residual sums of squares, every scalar of the loss on the device   -- 12 B read per pixel
//   * combine:  d total / d mlp_out = upstream (2 (pred - target) / n_data + (2 lam / n) J^T r) from the residual adjoint's fields
//                                                                                            -- 36 B read + 12 written per pixel
// Sums are DETERMINISTIC: a fixed grid, per-block partials in double, the block that finishes last adds them in index order.
#include "nns_common.h"
#include <cstdint>

using namespace nns;

namespace {

constexpr int kBlocks = 1024, kThreads = 256;

struct Ws {                       // the caller's workspace (nns_pinn_workspace_bytes, zeroed ONCE: the counter resets itself)
    double part[4][kBlocks];      // [0] = data term (assemble); [1..3] = r_u, r_v, r_div (loss)
    unsigned count;
};

__device__ __forceinline__ double block_sum(double x, double* sh) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = x;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

template <int V> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };
template <int V> __device__ __forceinline__ float at(const typename Vec<V>::T& x, int k);
template <> __device__ __forceinline__ float at<4>(const float4& x, int k) { return k == 0 ? x.x : k == 1 ? x.y : k == 2 ? x.z : x.w; }
template <> __device__ __forceinline__ float at<1>(const float& x, int) { return x; }
template <int V> __device__ __forceinline__ typename Vec<V>::T mk(const float* a);
template <> __device__ __forceinline__ float4 mk<4>(const float* a) { return make_float4(a[0], a[1], a[2], a[3]); }
template <> __device__ __forceinline__ float mk<1>(const float* a) { return a[0]; }

// out, state, target: [batch][3][npix]; u, v, p, u_prev, v_prev: [batch][npix].  One item = V consecutive pixels of one batch entry.
template <int V>
__global__ __launch_bounds__(kThreads) void pinn_assemble_kernel(const float* __restrict__ out, const float* __restrict__ state,
                                                                 const float* __restrict__ target, float* __restrict__ u, float* __restrict__ v,
                                                                 float* __restrict__ p, float* __restrict__ u_prev, float* __restrict__ v_prev,
                                                                 Ws* __restrict__ ws, long batch, long npv) {
    using T = typename Vec<V>::T;
    __shared__ double sh[4];
    float* const dst[3] = {u, v, p};
    float* const prev[2] = {u_prev, v_prev};
    double acc = 0.0;
    const long items = batch * npv;
    for (long it = (long)blockIdx.x * kThreads + threadIdx.x; it < items; it += (long)kBlocks * kThreads) {
        const long b = it / npv, i = it - b * npv;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const long src = (b * 3 + c) * npv + i;
            const T o = reinterpret_cast<const T*>(out)[src], s = reinterpret_cast<const T*>(state)[src];
            float pr[V];
#pragma unroll
            for (int k = 0; k < V; ++k) pr[k] = at<V>(s, k) + at<V>(o, k);
            reinterpret_cast<T*>(dst[c])[b * npv + i] = mk<V>(pr);
            if (c < 2 && u_prev) reinterpret_cast<T*>(prev[c])[b * npv + i] = s;
            if (target) {
                const T t = reinterpret_cast<const T*>(target)[src];
#pragma unroll
                for (int k = 0; k < V; ++k) { const float d = pr[k] - at<V>(t, k); a = fmaf(d, d, a); }
            }
        }
        acc += (double)a;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) ws->part[0][blockIdx.x] = s;
}

// r_u, r_v, r_div: n values each.  out3 = (total, data, phys).  w_div != 1: r_div is SCALED IN PLACE by w_div after its square went into
// the sum -- the residual adjoint then takes (r_u, r_v, r_div) as they stand (J^T is linear in each field).
template <int V>
__global__ __launch_bounds__(kThreads) void pinn_loss_kernel(const float* __restrict__ r_u, const float* __restrict__ r_v, float* __restrict__ r_d,
                                                             Ws* __restrict__ ws, long nv, double inv_n, double inv_n_data, double lam, double w_div,
                                                             int scale_rd, float* __restrict__ out3) {
    using T = typename Vec<V>::T;
    __shared__ double sh[4];
    __shared__ unsigned last;
    double acc[3] = {0.0, 0.0, 0.0};
    const float wd = (float)w_div;
    for (long it = (long)blockIdx.x * kThreads + threadIdx.x; it < nv; it += (long)kBlocks * kThreads) {
        const T a = reinterpret_cast<const T*>(r_u)[it], b = reinterpret_cast<const T*>(r_v)[it], c = reinterpret_cast<T*>(r_d)[it];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, sc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            s0 = fmaf(at<V>(a, k), at<V>(a, k), s0), s1 = fmaf(at<V>(b, k), at<V>(b, k), s1), s2 = fmaf(at<V>(c, k), at<V>(c, k), s2);
            sc[k] = wd * at<V>(c, k);
        }
        if (scale_rd) reinterpret_cast<T*>(r_d)[it] = mk<V>(sc);
        acc[0] += (double)s0, acc[1] += (double)s1, acc[2] += (double)s2;
    }
    for (int f = 0; f < 3; ++f) {
        const double s = block_sum(acc[f], sh);
        if (threadIdx.x == 0) ws->part[1 + f][blockIdx.x] = s;
    }
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(&ws->count, 1u) == (unsigned)kBlocks - 1u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double tot[4];
    for (int f = 0; f < 4; ++f) {                     // kBlocks partials per sum, always added in the same order
        double x = 0.0;
        for (int j = threadIdx.x; j < kBlocks; j += kThreads) x += __builtin_nontemporal_load(&ws->part[f][j]);
        tot[f] = block_sum(x, sh);
    }
    if (threadIdx.x == 0) {
        const double data = tot[0] * inv_n_data, phys = (tot[1] + tot[2] + w_div * tot[3]) * inv_n;
        out3[0] = (float)(data + lam * phys), out3[1] = (float)data, out3[2] = (float)phys;
        ws->count = 0u;
    }
}

// grad_out[b][c][i] = up_data[0] c_data (pred_c - target_c) + up_phys[0] c_phys g_c,  pred = (u, v, p), g = (g_u, g_v, g_p)
template <int V>
__global__ __launch_bounds__(kThreads) void pinn_combine_kernel(const float* __restrict__ g_u, const float* __restrict__ g_v, const float* __restrict__ g_p,
                                                                const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ p,
                                                                const float* __restrict__ target, const float* __restrict__ up_data,
                                                                const float* __restrict__ up_phys, float c_data, float c_phys,
                                                                float* __restrict__ grad_out, long batch, long npv) {
    using T = typename Vec<V>::T;
    const float* const g[3] = {g_u, g_v, g_p};
    const float* const f[3] = {u, v, p};
    const float cd = target ? up_data[0] * c_data : 0.f, cp = up_phys[0] * c_phys;
    const long items = batch * npv;
    for (long it = (long)blockIdx.x * kThreads + threadIdx.x; it < items; it += (long)gridDim.x * kThreads) {
        const long b = it / npv, i = it - b * npv;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const long dst = (b * 3 + c) * npv + i;
            const T gg = reinterpret_cast<const T*>(g[c])[b * npv + i];
            float o[V];
            if (target) {
                const T pr = reinterpret_cast<const T*>(f[c])[b * npv + i], t = reinterpret_cast<const T*>(target)[dst];
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = fmaf(cp, at<V>(gg, k), cd * (at<V>(pr, k) - at<V>(t, k)));
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = cp * at<V>(gg, k);
            }
            reinterpret_cast<T*>(grad_out)[dst] = mk<V>(o);
        }
    }
}

bool aligned16(std::initializer_list<const void*> ps) {
    for (const void* q : ps)
        if (q && (reinterpret_cast<uintptr_t>(q) & 15)) return false;
    return true;
}

}  // namespace

NNS_API long nns_pinn_workspace_bytes(void) { return (long)sizeof(Ws); }

NNS_API int nns_pinn_assemble_f32(const float* out, const float* state, const float* target, float* u, float* v, float* p, float* u_prev, float* v_prev,
                                  void* ws, int batch, long npix, void* stream) {
    if (!out || !state || !u || !v || !p || !ws) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_assemble_f32: NULL argument");
    if ((u_prev == nullptr) != (v_prev == nullptr)) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_assemble_f32: u_prev and v_prev come together");
    if (batch < 1 || npix < 1) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_assemble_f32: bad sizes (batch=%d npix=%ld)", batch, npix);
    if (reinterpret_cast<uintptr_t>(ws) & 7) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_assemble_f32: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (npix % 4 == 0 && aligned16({out, state, target, u, v, p, u_prev, v_prev}))
        hipLaunchKernelGGL(pinn_assemble_kernel<4>, dim3(kBlocks), dim3(kThreads), 0, s, out, state, target, u, v, p, u_prev, v_prev, (Ws*)ws, (long)batch, npix / 4);
    else
        hipLaunchKernelGGL(pinn_assemble_kernel<1>, dim3(kBlocks), dim3(kThreads), 0, s, out, state, target, u, v, p, u_prev, v_prev, (Ws*)ws, (long)batch, npix);
    return check_launch("nns_pinn_assemble_f32");
}

NNS_API int nns_pinn_loss_f32(const float* r_u, const float* r_v, float* r_div, long n, void* ws, double n_data, double lam, double w_div, float* out3,
                              void* stream) {
    if (!r_u || !r_v || !r_div || !ws || !out3) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_loss_f32: NULL argument");
    if (n < 1 || n_data < 0) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_loss_f32: bad sizes (n=%ld n_data=%g)", n, n_data);
    if (reinterpret_cast<uintptr_t>(ws) & 7) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_loss_f32: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const double inv_n = 1.0 / (double)n, inv_nd = n_data > 0 ? 1.0 / n_data : 0.0;
    const int scale = w_div != 1.0;
    if (n % 4 == 0 && aligned16({r_u, r_v, r_div}))
        hipLaunchKernelGGL(pinn_loss_kernel<4>, dim3(kBlocks), dim3(kThreads), 0, s, r_u, r_v, r_div, (Ws*)ws, n / 4, inv_n, inv_nd, lam, w_div, scale, out3);
    else
        hipLaunchKernelGGL(pinn_loss_kernel<1>, dim3(kBlocks), dim3(kThreads), 0, s, r_u, r_v, r_div, (Ws*)ws, n, inv_n, inv_nd, lam, w_div, scale, out3);
    return check_launch("nns_pinn_loss_f32");
}

NNS_API int nns_pinn_combine_f32(const float* g_u, const float* g_v, const float* g_p, const float* u, const float* v, const float* p, const float* target,
                                 const float* up_data, const float* up_phys, double c_data, double c_phys, float* grad_out, int batch, long npix,
                                 void* stream) {
    if (!g_u || !g_v || !g_p || !up_phys || !grad_out) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_combine_f32: NULL argument");
    if (target && (!u || !v || !p || !up_data)) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_combine_f32: a target needs u, v, p and up_data");
    if (batch < 1 || npix < 1) return fail(NNS_ERR_INVALID_ARG, "nns_pinn_combine_f32: bad sizes (batch=%d npix=%ld)", batch, npix);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = npix % 4 == 0 && aligned16({g_u, g_v, g_p, u, v, p, target, grad_out});
    const long items = (long)batch * (vec ? npix / 4 : npix);
    const unsigned grid = (unsigned)(items / kThreads < 1 ? 1 : (items / kThreads > 8192 ? 8192 : items / kThreads));
    if (vec)
        hipLaunchKernelGGL(pinn_combine_kernel<4>, dim3(grid), dim3(kThreads), 0, s, g_u, g_v, g_p, u, v, p, target, up_data, up_phys, (float)c_data, (float)c_phys,
                           grad_out, (long)batch, npix / 4);
    else
        hipLaunchKernelGGL(pinn_combine_kernel<1>, dim3(grid), dim3(kThreads), 0, s, g_u, g_v, g_p, u, v, p, target, up_data, up_phys, (float)c_data, (float)c_phys,
                           grad_out, (long)batch, npix);
    return check_launch("nns_pinn_combine_f32");
}
