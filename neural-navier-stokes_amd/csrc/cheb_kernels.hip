// chorin_spectral (Chebyshev collocation) operators on gfx950: src/chorin_spectral/simulate.py:232-383.
// The per-step work of the reference is ~30 dense (N-2)x(N-2) float64 matmuls (derivatives as D @ U and
// U @ D^T, the diagonalised Helmholtz / Uzawa solves) plus a few elementwise assemblies.  Here:
//   * nns_cheb_gemm_f64: C = alpha * op(A) op(B) + beta * C on the matrix cores (v_mfma_f64_16x16x4_f64,
//     one wave per 16x16 output tile; operands stay L2-resident, they are a few tens of KB);
//   * small fused kernels for the RHS assembly (:277-282), the eigenvalue-sum division (:287-288,:372-373) and
//     the embedding of interior + boundary rows (:322-334).
// f64 MFMA maps (cdna_hip_programming.md section 3): A: lane l holds A[row = l&15][k = l>>4]; B: lane l holds
// B[k = l>>4][col = l&15]; D: lane l holds D[row = (l>>4) + 4*r][col = l&15], r = 0..3.
#include "nns_common.h"

using namespace nns;

namespace {

using f64x4 = __attribute__((ext_vector_type(4))) double;

__global__ __launch_bounds__(64) void cheb_gemm_kernel(const double* __restrict__ A, int lda, int ta, const double* __restrict__ B, int ldb, int tb,
                                                        double* __restrict__ C, int ldc, int M, int N, int K, double alpha, double beta) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const size_t boff = (size_t)blockIdx.z;
    const double* Ab = A + boff * (size_t)(ta ? K : M) * lda;
    const double* Bb = B + boff * (size_t)(tb ? N : K) * ldb;
    double* Cb = C + boff * (size_t)M * ldc;
    f64x4 acc = {0., 0., 0., 0.};
    const int row = m0 + r, col = n0 + r;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + q;
        const double a = (row < M && k < K) ? (ta ? Ab[(size_t)k * lda + row] : Ab[(size_t)row * lda + k]) : 0.;
        const double b = (col < N && k < K) ? (tb ? Bb[(size_t)col * ldb + k] : Bb[(size_t)k * ldb + col]) : 0.;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = m0 + q + 4 * i, cc = n0 + r;
        if (rr < M && cc < N) {
            double* p = Cb + (size_t)rr * ldc + cc;
            *p = beta == 0. ? alpha * acc[i] : alpha * acc[i] + beta * *p;
        }
    }
}

// F = 2 f - 3 dt (un fx + vn fy) + dt (un1 f1x + vn1 f1y) + dt (fxx + fyy)     (:277-282)
__global__ __launch_bounds__(256) void cheb_rhs_kernel(const double* f, const double* un, const double* vn, const double* un1, const double* vn1,
                                                        const double* fx, const double* fy, const double* f1x, const double* f1y,
                                                        const double* fxx, const double* fyy, double* F, int n, double dt) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    F[e] = 2 * f[e] - 3 * dt * (un[e] * fx[e] + vn[e] * fy[e]) + dt * (un1[e] * f1x[e] + vn1[e] * f1y[e]) + dt * (fxx[e] + fyy[e]);
}

// out[i][j] = Hm[i][j] / (c0 + cx * lx[i] + cy * ly[j])                      (:287-288, :372-373)
__global__ __launch_bounds__(256) void cheb_diag_div_kernel(const double* Hm, const double* lx, const double* ly, double* out, int ni, int nj,
                                                             double c0, double cx, double cy) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ni * nj) return;
    const int i = e / nj, j = e % nj;
    out[e] = Hm[e] / (c0 + cx * lx[i] + cy * ly[j]);
}

// full[Nx][Ny]: interior = sol, rows 0 / Nx-1 = x0 / xN, columns 0 / Ny-1 = y0 / yN, corners 0   (:322-334)
__global__ __launch_bounds__(256) void cheb_embed_kernel(const double* sol, const double* x0, const double* xN, const double* y0, const double* yN,
                                                          double* full, int Nx, int Ny) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Nx * Ny) return;
    const int i = e / Ny, j = e % Ny;
    const bool ei = (i == 0 || i == Nx - 1), ej = (j == 0 || j == Ny - 1);
    double v;
    if (ei && ej) v = 0.;
    else if (i == 0) v = x0[j - 1];
    else if (i == Nx - 1) v = xN[j - 1];
    else if (j == 0) v = y0[i - 1];
    else if (j == Ny - 1) v = yN[i - 1];
    else v = sol[(size_t)(i - 1) * (Ny - 2) + (j - 1)];
    full[e] = v;
}

}  // namespace

#define S(stream) reinterpret_cast<hipStream_t>(stream)

NNS_API int nns_cheb_gemm_f64(const double* A, int lda, int transA, const double* B, int ldb, int transB, double* C, int ldc,
                              int M, int N, int K, double alpha, double beta, int batch, void* stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch < 1 || lda < 1 || ldb < 1 || ldc < N)
        return fail(NNS_ERR_INVALID_ARG, "cheb_gemm: bad args (M=%d N=%d K=%d lda=%d ldb=%d ldc=%d)", M, N, K, lda, ldb, ldc);
    if (batch > 65535) return fail(NNS_ERR_UNSUPPORTED, "cheb_gemm: batch > 65535");
    hipLaunchKernelGGL(cheb_gemm_kernel, dim3((N + 15) / 16, (M + 15) / 16, batch), dim3(64), 0, S(stream), A, lda, transA, B, ldb, transB, C, ldc, M, N, K, alpha, beta);
    return check_launch("cheb_gemm");
}

NNS_API int nns_cheb_helmholtz_rhs_f64(const double* f, const double* un, const double* vn, const double* un1, const double* vn1,
                                       const double* fx, const double* fy, const double* f1x, const double* f1y, const double* fxx,
                                       const double* fyy, double* F, int n, double dt, void* stream) {
    if (!f || !un || !vn || !un1 || !vn1 || !fx || !fy || !f1x || !f1y || !fxx || !fyy || !F || n < 1) return fail(NNS_ERR_INVALID_ARG, "cheb_helmholtz_rhs: bad args");
    hipLaunchKernelGGL(cheb_rhs_kernel, dim3((n + 255) / 256), dim3(256), 0, S(stream), f, un, vn, un1, vn1, fx, fy, f1x, f1y, fxx, fyy, F, n, dt);
    return check_launch("cheb_helmholtz_rhs");
}

NNS_API int nns_cheb_diag_div_f64(const double* Hm, const double* lam_x, const double* lam_y, double* out, int ni, int nj,
                                  double c0, double cx, double cy, void* stream) {
    if (!Hm || !lam_x || !lam_y || !out || ni < 1 || nj < 1) return fail(NNS_ERR_INVALID_ARG, "cheb_diag_div: bad args");
    hipLaunchKernelGGL(cheb_diag_div_kernel, dim3((ni * nj + 255) / 256), dim3(256), 0, S(stream), Hm, lam_x, lam_y, out, ni, nj, c0, cx, cy);
    return check_launch("cheb_diag_div");
}

NNS_API int nns_cheb_embed_f64(const double* sol, const double* x0, const double* xN, const double* y0, const double* yN, double* full,
                               int Nx, int Ny, void* stream) {
    if (!sol || !x0 || !xN || !y0 || !yN || !full || Nx < 3 || Ny < 3) return fail(NNS_ERR_INVALID_ARG, "cheb_embed: bad args");
    hipLaunchKernelGGL(cheb_embed_kernel, dim3((Nx * Ny + 255) / 256), dim3(256), 0, S(stream), sol, x0, xN, y0, yN, full, Nx, Ny);
    return check_launch("cheb_embed");
}
