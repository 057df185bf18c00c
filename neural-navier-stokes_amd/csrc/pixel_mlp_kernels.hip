// Per-pixel MLP (1x1-convolution stack) forward on gfx950: the reference's BasisFunc
// (src/neural_spectral/spectral_ode.py:100-119: Conv2d(3,16,1)-ReLU-Conv2d(16,32,1)-ReLU-...-Conv2d(16,3,1)),
// generalised to any depth <= 8 and width <= 64 (BASELINE configs 2 and 3: depth-4 width-32 float32, depth-8
// width-64 bfloat16 weights/activations with float32 accumulation).
//
// GEMM-shaped (M = pixels, N = K = channels) -> matrix cores.  The layers are CHAINED IN REGISTERS:
// a wave owns a tile of 32 pixels and computes the transposed product  Y^T[out x pix] = W[out x in] X^T[in x pix]
// with A = W (from LDS) and B = X^T.  The 32x32 accumulator has the pixel on the lane and the channels in its 16
// registers, which is exactly what the NEXT layer's B operand needs ("an accumulator tile as the next MFMA's
// operand", cdna_hip_programming.md section 3) -- provided the weights' k order is permuted to the accumulator's
// register order.  That permutation is applied once, when the weights are staged into LDS, so no activation
// ever goes through LDS or HBM between layers: HBM traffic is C_in + C_out floats per pixel.
//   float32 : v_mfma_f32_32x32x2_f32   (exact fp32 fma chain; k-step i uses channels {c(i), c(i)+4},
//             c(i) = (i&3) + 8*(i>>2): the two rows register i holds in lane halves 0 / 1)
//   bfloat16: v_mfma_f32_32x32x16_bf16 (k-step s uses registers 8s..8s+7 packed pairwise to bf16; element j of
//             lane half h is channel 16s + 8*(j>>2) + 4h + (j&3))
#include "nns_common.h"

using namespace nns;

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;      // 8 bf16 in 4 VGPRs

constexpr int kMaxLayers = 8;
constexpr int kMaxWidth = 64;

struct PixelMlpDesc {
    int nlayers;
    int cin[kMaxLayers], cout[kMaxLayers];
    int woff[kMaxLayers], boff[kMaxLayers];      // offsets (floats) into the packed weight / bias arrays
    int lds_off[kMaxLayers];                     // offset (bytes) of the layer's pre-permuted fragments in LDS
    int lds_bias[kMaxLayers];                    // offset (bytes) of the layer's padded bias
};

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }     // C/D map, 32x32

__device__ __forceinline__ unsigned short f2bf(float x) {          // round-to-nearest-even; NaN-safe via the plain cast
    return __builtin_bit_cast(unsigned short, (__bf16)x);
}

// LDS image of a layer's weights, one entry per (out tile ot, k block, lane):
//   float32 : [ot][kb = in/32][i = 0..15][lane 64]  float   = W[32 ot + (lane&31)][32 kb + c(i) + 4 (lane>>5)]
//   bfloat16: [ot][s  = in/16][lane 64][8]          bf16    = W[32 ot + (lane&31)][16 s + 8 (j>>2) + 4 (lane>>5) + (j&3)]
template <bool BF16>
__device__ void stage_weights(const PixelMlpDesc& d, const float* __restrict__ W, const float* __restrict__ B, unsigned char* lds, int tid, int nthreads) {
    for (int l = 0; l < d.nlayers; ++l) {
        const int cin = d.cin[l], cout = d.cout[l];
        const int ots = (cout + 31) / 32;
        const float* Wl = W + d.woff[l];
        if constexpr (BF16) {
            const int ss = (cin + 15) / 16;
            unsigned short* dst = reinterpret_cast<unsigned short*>(lds + d.lds_off[l]);
            for (int e = tid; e < ots * ss * 64 * 8; e += nthreads) {
                const int j = e & 7, lane = (e >> 3) & 63, s = (e >> 9) % ss, ot = (e >> 9) / ss;
                const int row = 32 * ot + (lane & 31), k = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
                dst[e] = f2bf((row < cout && k < cin) ? Wl[row * cin + k] : 0.f);
            }
        } else {
            const int kbs = (cin + 31) / 32;
            float* dst = reinterpret_cast<float*>(lds + d.lds_off[l]);
            for (int e = tid; e < ots * kbs * 16 * 64; e += nthreads) {
                const int lane = e & 63, i = (e >> 6) & 15, kb = (e >> 10) % kbs, ot = (e >> 10) / kbs;
                const int row = 32 * ot + (lane & 31), k = 32 * kb + acc_row(i, lane >> 5);
                dst[e] = (row < cout && k < cin) ? Wl[row * cin + k] : 0.f;
            }
        }
        float* bl = reinterpret_cast<float*>(lds + d.lds_bias[l]);
        for (int e = tid; e < ots * 32; e += nthreads) bl[e] = e < cout ? B[d.boff[l] + e] : 0.f;
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void pixel_mlp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ Bv,
                                                             float* __restrict__ y, long npix_total, int P, PixelMlpDesc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    stage_weights<BF16>(d, W, Bv, lds, threadIdx.x, 256);
    __syncthreads();
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, r = lane & 31, h = lane >> 5;
    const int cin0 = d.cin[0], coutL = d.cout[d.nlayers - 1];
    const long ntiles = (npix_total + 31) / 32;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const long gp = tile * 32 + r;
        const bool ok = gp < npix_total;
        const long b = ok ? gp / P : 0, p = ok ? gp % P : 0;
        const float* xb = x + (size_t)b * cin0 * P + p;
        // activations: up to 64 channels x 32 pixels = two 32-row accumulator tiles
        f32x16 act[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int c = 32 * t + acc_row(i, h); act[t][i] = (ok && c < cin0) ? xb[(size_t)c * P] : 0.f; }
        for (int l = 0; l < d.nlayers; ++l) {
            const int cin = d.cin[l], cout = d.cout[l];
            const int ots = (cout + 31) / 32;
            const float* bl = reinterpret_cast<const float*>(lds + d.lds_bias[l]);
            f32x16 out[2];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                if (ot < ots) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) out[ot][i] = bl[32 * ot + acc_row(i, h)];          // bias as the initial accumulator
                    if constexpr (BF16) {
                        const int ss = (cin + 15) / 16;
                        const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds + d.lds_off[l]);
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            if (s < ss) {
                                bf16x8 bfrag;
#pragma unroll
                                for (int j = 0; j < 8; ++j) bfrag[j] = (short)f2bf(act[s >> 1][8 * (s & 1) + j]);
                                out[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[(ot * ss + s) * 64 + lane], bfrag, out[ot], 0, 0, 0);
                            }
                        }
                    } else {
                        const int kbs = (cin + 31) / 32;
                        const float* wl = reinterpret_cast<const float*>(lds + d.lds_off[l]);
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            if (kb < kbs) {
#pragma unroll
                                for (int i = 0; i < 16; ++i)
                                    out[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[((ot * kbs + kb) * 16 + i) * 64 + lane], act[kb][i], out[ot], 0, 0, 0);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) out[ot][i] = 0.f;
                }
            }
            const bool relu = l + 1 < d.nlayers;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) act[t][i] = relu ? fmaxf(out[t][i], 0.f) : out[t][i];
        }
        if (ok) {
            float* yb = y + (size_t)b * coutL * P + p;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) { const int c = 32 * t + acc_row(i, h); if (c < coutL) yb[(size_t)c * P] = act[t][i]; }
        }
    }
}

}  // namespace

// x [mb, C_in, P], y [mb, C_out, P] (NCHW with P = nx*ny, as the reference's Conv2d stack); weights packed layer after
// layer in torch layout [C_out_l][C_in_l] (a 1x1 Conv2d weight squeezed), biases packed likewise; widths[0..nlayers]
// = C_in, hidden..., C_out.  ReLU between layers, none after the last (spectral_ode.py:106-116).
// bf16 != 0: weights and inter-layer activations rounded to bfloat16, float32 accumulation (config 3).
NNS_API int nns_pixel_mlp_fwd_f32(const float* x, const float* weights, const float* biases, float* y, int mb, int P,
                                  const int* widths_host, int nlayers, int bf16, void* stream) {
    if (!x || !weights || !biases || !y || !widths_host || mb < 1 || P < 1) return fail(NNS_ERR_INVALID_ARG, "pixel_mlp_fwd: bad args");
    if (nlayers < 1 || nlayers > kMaxLayers) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_fwd: %d layers (1..%d supported)", nlayers, kMaxLayers);
    PixelMlpDesc d;
    d.nlayers = nlayers;
    int woff = 0, boff = 0, lds = 0;
    for (int l = 0; l < kMaxLayers; ++l) { d.cin[l] = d.cout[l] = 1; d.woff[l] = d.boff[l] = d.lds_off[l] = d.lds_bias[l] = 0; }
    for (int l = 0; l < nlayers; ++l) {
        const int cin = widths_host[l], cout = widths_host[l + 1];
        if (cin < 1 || cout < 1 || cin > kMaxWidth || cout > kMaxWidth) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_fwd: layer %d is %d -> %d (widths 1..%d supported)", l, cin, cout, kMaxWidth);
        d.cin[l] = cin; d.cout[l] = cout; d.woff[l] = woff; d.boff[l] = boff;
        woff += cin * cout; boff += cout;
        const int ots = (cout + 31) / 32;
        d.lds_off[l] = lds;
        lds += bf16 ? ots * ((cin + 15) / 16) * 64 * 16 : ots * ((cin + 31) / 32) * 16 * 64 * 4;
        d.lds_bias[l] = lds;
        lds += ots * 32 * 4;
    }
    if (lds > 160 * 1024) return fail(NNS_ERR_UNSUPPORTED, "pixel_mlp_fwd: weights need %d B of LDS (> 160 KiB)", lds);
    const long npix = (long)mb * P;
    const long ntiles = (npix + 31) / 32;
    long blocks = (ntiles + 3) / 4; if (blocks > 1024) blocks = 1024;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e;
    if (bf16) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(pixel_mlp_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "pixel_mlp_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(pixel_mlp_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, s, x, weights, biases, y, npix, P, d);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(pixel_mlp_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "pixel_mlp_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(pixel_mlp_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, s, x, weights, biases, y, npix, P, d);
    }
    return check_launch("pixel_mlp_fwd");
}
