// The optimiser step of the reference's training loops (src/neural_spectral/spectral_ode.py:171,189; spectral_ode2.py:159,171; rnn.py:90,102;
// spectral_rnn.py:131,149: torch.optim.Adam(model.parameters(), lr=1e-3), optimizer.step()) as ONE launch over all parameter tensors.
//
// torch's default (foreach) Adam is seven launches of 5-13 us over the parameter list -- 70 us of BASELINE config 2's 0.8 ms training iteration
// (profiles/r04_c2_kernel_summary.txt).  Arithmetic, per element, in torch's order (torch/optim/adam.py, _multi_tensor_adam, no amsgrad):
//     g    = grad (+ weight_decay * p)            (maximize: g = -grad)
//     m    = m + (1 - beta1) (g - m)              (lerp)
//     v    = beta2 v + (1 - beta2) g g
//     p    = p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// The bias corrections are computed on the host in double, as torch does for a host-side step counter.  HBM-bound: 16 B read + 12 B written per
// parameter; the parameter lists here are small (config 2: 0.5 M floats, config 5: 2 M), so what matters is the launch count.
// A second entry point zeroes a list of buffers in one launch (the six gradient buffers of the ODE-MLP backward were six memsets).
#include "nns_common.h"
#include <cmath>
#include <cstdint>

using namespace nns;

namespace {

constexpr int kMaxTensors = 24;       // per launch (the kernel argument carries the table)
constexpr int kChunk = 2048;          // elements per workgroup: 256 threads x 2 float4

struct AdamTable {
    float* p[kMaxTensors];
    const float* g[kMaxTensors];
    float* m[kMaxTensors];
    float* v[kMaxTensors];
    long n[kMaxTensors];
    int first_chunk[kMaxTensors + 1];   // chunk index at which tensor i starts
    int count;
};

struct AdamK { float lr_over_bc1, inv_sqrt_bc2, beta1m, beta2, beta2m, eps, weight_decay, gsign; };

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamK& k) {
    g *= k.gsign;
    if (k.weight_decay != 0.f) g = fmaf(k.weight_decay, p, g);
    m = fmaf(k.beta1m, g - m, m);
    v = fmaf(k.beta2m * g, g, k.beta2 * v);
    const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
    p = p - k.lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(AdamTable t, AdamK k) {
    int lo = 0, hi = t.count;                       // the tensor this chunk belongs to: first_chunk[lo] <= blockIdx.x < first_chunk[lo + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= t.first_chunk[mid]) lo = mid; else hi = mid;
    }
    const long n = t.n[lo], base = (long)(blockIdx.x - t.first_chunk[lo]) * kChunk;
    float* __restrict__ p = t.p[lo];
    const float* __restrict__ g = t.g[lo];
    float* __restrict__ m = t.m[lo];
    float* __restrict__ v = t.v[lo];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const long e = base + 4 * (threadIdx.x + 256 * q);
        if (e >= n) continue;
        if (vec && e + 4 <= n) {
            float4 pp = *reinterpret_cast<float4*>(p + e), mm = *reinterpret_cast<float4*>(m + e), vv = *reinterpret_cast<float4*>(v + e);
            const float4 gg = *reinterpret_cast<const float4*>(g + e);
            adam_elem(pp.x, gg.x, mm.x, vv.x, k), adam_elem(pp.y, gg.y, mm.y, vv.y, k), adam_elem(pp.z, gg.z, mm.z, vv.z, k), adam_elem(pp.w, gg.w, mm.w, vv.w, k);
            *reinterpret_cast<float4*>(p + e) = pp, *reinterpret_cast<float4*>(m + e) = mm, *reinterpret_cast<float4*>(v + e) = vv;
        } else {
            for (long i = e; i < n && i < e + 4; ++i) adam_elem(p[i], g[i], m[i], v[i], k);
        }
    }
}

struct ZeroTable { void* p[kMaxTensors]; long bytes[kMaxTensors]; int first_chunk[kMaxTensors + 1]; int count; };
constexpr int kZeroChunk = 16384;     // bytes per workgroup

__global__ __launch_bounds__(256) void zero_list_kernel(ZeroTable t) {
    int lo = 0, hi = t.count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= t.first_chunk[mid]) lo = mid; else hi = mid;
    }
    unsigned char* b = static_cast<unsigned char*>(t.p[lo]);
    const long n = t.bytes[lo], base = (long)(blockIdx.x - t.first_chunk[lo]) * kZeroChunk;
    const bool vec = (reinterpret_cast<uintptr_t>(b) & 15) == 0;
#pragma unroll
    for (int q = 0; q < kZeroChunk / (256 * 16); ++q) {
        const long e = base + 16 * (threadIdx.x + 256 * q);
        if (e >= n) continue;
        if (vec && e + 16 <= n) *reinterpret_cast<uint4*>(b + e) = make_uint4(0u, 0u, 0u, 0u);
        else for (long i = e; i < n && i < e + 16; ++i) b[i] = 0;
    }
}

}  // namespace

namespace nns {
// (used by the ODE-MLP backward: csrc/neural_kernels.hip)
int zero_buffers(void* const* bufs, const long* bytes, int count, hipStream_t s) {
    for (int i = 0; i < count;) {
        ZeroTable t{};
        int c = 0, chunks = 0;
        for (; i < count && c < kMaxTensors; ++i) {
            if (bytes[i] <= 0) continue;
            if (!bufs[i]) return fail(NNS_ERR_INVALID_ARG, "zero_buffers: buffer %d is NULL", i);
            t.p[c] = bufs[i], t.bytes[c] = bytes[i], t.first_chunk[c] = chunks;
            chunks += (int)((bytes[i] + kZeroChunk - 1) / kZeroChunk);
            ++c;
        }
        if (!c) continue;
        t.first_chunk[c] = chunks, t.count = c;
        hipLaunchKernelGGL(zero_list_kernel, dim3(chunks), dim3(256), 0, s, t);
    }
    return check_launch("zero_buffers");
}
}  // namespace nns

NNS_API int nns_adam_step_f32(float* const* params_host, const float* const* grads_host, float* const* exp_avg_host, float* const* exp_avg_sq_host,
                              const long* sizes_host, int ntensors, double lr, double beta1, double beta2, double eps, double weight_decay, long step,
                              int maximize, void* stream) {
    if (ntensors < 0 || (ntensors > 0 && (!params_host || !grads_host || !exp_avg_host || !exp_avg_sq_host || !sizes_host)))
        return fail(NNS_ERR_INVALID_ARG, "nns_adam_step_f32: NULL table (ntensors=%d)", ntensors);
    if (step < 1 || !(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(weight_decay >= 0.0))
        return fail(NNS_ERR_INVALID_ARG, "nns_adam_step_f32: bad hyper-parameters (step=%ld lr=%g betas=%g,%g eps=%g weight_decay=%g)", step, lr, beta1, beta2, eps,
                    weight_decay);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamK k;
    k.lr_over_bc1 = (float)(lr / bc1), k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2)), k.beta1m = (float)(1.0 - beta1), k.beta2 = (float)beta2, k.beta2m = (float)(1.0 - beta2);
    k.eps = (float)eps, k.weight_decay = (float)weight_decay, k.gsign = maximize ? -1.f : 1.f;
    for (int i = 0; i < ntensors; ++i) {
        if (sizes_host[i] < 0) return fail(NNS_ERR_INVALID_ARG, "nns_adam_step_f32: tensor %d has size %ld", i, sizes_host[i]);
        if (sizes_host[i] > 0 && (!params_host[i] || !grads_host[i] || !exp_avg_host[i] || !exp_avg_sq_host[i]))
            return fail(NNS_ERR_INVALID_ARG, "nns_adam_step_f32: tensor %d has a NULL buffer", i);
        if (sizes_host[i] > (long)kChunk * 0x3fffffffL) return fail(NNS_ERR_UNSUPPORTED, "nns_adam_step_f32: tensor %d too large (%ld)", i, sizes_host[i]);
    }
    hipStream_t s = (hipStream_t)stream;
    for (int i0 = 0; i0 < ntensors;) {
        AdamTable t{};
        int c = 0;
        long chunks = 0;
        int i = i0;
        for (; i < ntensors && c < kMaxTensors; ++i) {
            if (sizes_host[i] == 0) continue;
            const long nc = (sizes_host[i] + kChunk - 1) / kChunk;
            if (chunks + nc > 0x7fffffffL) break;
            t.p[c] = params_host[i], t.g[c] = grads_host[i], t.m[c] = exp_avg_host[i], t.v[c] = exp_avg_sq_host[i], t.n[c] = sizes_host[i];
            t.first_chunk[c] = (int)chunks;
            chunks += nc;
            ++c;
        }
        if (i == i0) return fail(NNS_ERR_UNSUPPORTED, "nns_adam_step_f32: tensor %d does not fit one launch", i0);
        i0 = i;
        if (!c) continue;
        t.first_chunk[c] = (int)chunks, t.count = c;
        hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)chunks), dim3(256), 0, s, t, k);
    }
    return check_launch("nns_adam_step_f32");
}
