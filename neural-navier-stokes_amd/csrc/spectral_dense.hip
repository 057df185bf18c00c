// Fourier-spectral residual on axes whose length the LDS FFT engine does not serve (anything but a power of two in [64, 1024]):
// the reference drivers' own grids are 51 x 51 and 50 x 50 (/root/reference src/chorin_fd/simulate.py:280-281,
// src/direct_fd/simulate.py:153-154), so the physics-informed spectral loss has to accept them.
// Operator definition: oracle/periodic.py (spectral_derivs / spectral_residual / spectral_residual_vjp); no reference symbol
// (SURVEY.md section 8 row a17).
//
// A Fourier derivative on a periodic axis of n points is a CIRCULANT matrix: (D f)_i = sum_m d[m] f[(i - m) mod n], with
//     d1[m] = -(2/n) sum_{k=1..K} kappa_k sin(2 pi k m / n)                   (i kappa; the Nyquist mode of an even axis dropped)
//     d2[m] = -(1/n) [ 2 sum_{k=1..K} kappa_k^2 cos(2 pi k m / n) + (n even) kappa_{n/2}^2 cos(pi m) ]     (-kappa^2)
// kappa_k = 2 pi k / L, K = (n - 1) / 2.  The vectors are built on the host in float64 (O(n^2) once per (n, L), cached on the
// device) and applied by one thread per grid point in float64: O(n) work per point and axis -- a fallback for small or odd
// sizes, not a fast path (51^2: 5 k multiply-adds per point; the FFT path stays the product for the BASELINE sizes).
// The two directions are separate launches with the same partial-field convention as the FFT passes (x-pass leaves
// P_u, P_v, P_d in the outputs, the y-pass finishes them), so an axis pair may mix the two engines (e.g. 96 x 256).
#include "spectral_common.h"
#include <map>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

using namespace nns;

namespace {

constexpr int kDenseMax = nns::spec::kDenseMaxLen;

struct Circ { const double* d1; const double* d2; };

// device-resident circulant vectors of (device, n, L): built once per device, never freed (a handful of sizes per process).  The FIRST call
// for a size builds the table on the host (O(n^2)), allocates and copies synchronously -- none of which may happen while the stream is being
// captured into a HIP graph: a capture that meets a missing table is refused with a clear message (build it first: any eager call of the
// same size, or nns_spec_dense_warmup).
int circulant(int n, double L, hipStream_t s, Circ& out) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, double>, double*> cache;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "spectral (dense): hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({dev, n, L});
    if (it == cache.end()) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(NNS_ERR_UNSUPPORTED, "spectral (dense): the circulant table of n=%d, L=%g is not built yet on device %d and cannot be built during a stream capture "
                        "(allocation + synchronous copy): call nns_spec_dense_warmup(n, L) or run the step once eagerly before capturing", n, L, dev);
        std::vector<double> h(2 * (size_t)n);
        const int K = (n - 1) / 2;
        const double ks = 2.0 * M_PI / L;
        for (int m = 0; m < n; ++m) {
            double a1 = 0, a2 = 0;
            for (int k = 1; k <= K; ++k) {
                const long r = ((long)k * m) % n;                     // exact argument reduction
                const double th = 2.0 * M_PI * (double)r / (double)n, kap = ks * k;
                a1 += kap * std::sin(th);
                a2 += kap * kap * std::cos(th);
            }
            double nyq = 0;
            if (n % 2 == 0) { const double kap = ks * (n / 2); nyq = kap * kap * ((m & 1) ? -1.0 : 1.0); }
            h[m] = -2.0 * a1 / n;
            h[n + m] = -(2.0 * a2 + nyq) / n;
        }
        double* d = nullptr;
        e = hipMalloc(&d, h.size() * sizeof(double));
        if (e != hipSuccess) return fail(NNS_ERR_LAUNCH, "spectral (dense): hipMalloc: %s", hipGetErrorString(e));
        // synchronous copy: the host vector dies at the end of this scope, and the table must be complete before any stream uses it
        e = hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return fail(NNS_ERR_LAUNCH, "spectral (dense): hipMemcpy: %s", hipGetErrorString(e)); }
        it = cache.emplace(std::make_tuple(dev, n, L), d).first;
    }
    out.d1 = it->second;
    out.d2 = it->second + n;
    return NNS_OK;
}

constexpr int kBx = 64, kBy = 4;

// axis 0 (columns): lines are strided by ny; consecutive threads take consecutive columns, so every load of the m-loop is coalesced
__global__ __launch_bounds__(kBx * kBy) void dense_fwd_x(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ p,
                                                          float* __restrict__ ru, float* __restrict__ rv, float* __restrict__ rd,
                                                          int nx, int ny, Circ c, double inv_rho, double nu) {
    const int j = blockIdx.x * kBx + threadIdx.x, i = blockIdx.y * kBy + threadIdx.y;
    if (j >= ny || i >= nx) return;
    const size_t g = (size_t)blockIdx.z * nx * ny;
    double ux = 0, vx = 0, px = 0, uxx = 0, vxx = 0;
    int r = i;                                                        // (i - m) mod nx, walked down
    for (int m = 0; m < nx; ++m) {
        const size_t q = g + (size_t)r * ny + j;
        const double a = c.d1[m], b = c.d2[m], uu = u[q], vv = v[q];
        ux += a * uu; vx += a * vv; px += a * (double)p[q]; uxx += b * uu; vxx += b * vv;
        r = r == 0 ? nx - 1 : r - 1;
    }
    const size_t q = g + (size_t)i * ny + j;
    const double uc = u[q];
    ru[q] = (float)(uc * ux + px * inv_rho - nu * uxx);              // P_u = u u_x + p_x/rho - nu u_xx
    rv[q] = (float)(uc * vx - nu * vxx);                              // P_v = u v_x - nu v_xx
    rd[q] = (float)ux;                                                // P_d = u_x
}

// axis 1 (rows): lane j reads element (j - m) mod ny of its row -- a rotating contiguous window
__global__ __launch_bounds__(kBx * kBy) void dense_fwd_y(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ p,
                                                          const float* __restrict__ up, const float* __restrict__ vp,
                                                          float* __restrict__ ru, float* __restrict__ rv, float* __restrict__ rd,
                                                          int nx, int ny, Circ c, double inv_rho, double nu, double inv_dt) {
    const int j = blockIdx.x * kBx + threadIdx.x, i = blockIdx.y * kBy + threadIdx.y;
    if (j >= ny || i >= nx) return;
    const size_t row = (size_t)blockIdx.z * nx * ny + (size_t)i * ny;
    double uy = 0, vy = 0, py = 0, uyy = 0, vyy = 0;
    int r = j;
    for (int m = 0; m < ny; ++m) {
        const double a = c.d1[m], b = c.d2[m], uu = u[row + r], vv = v[row + r];
        uy += a * uu; vy += a * vv; py += a * (double)p[row + r]; uyy += b * uu; vyy += b * vv;
        r = r == 0 ? ny - 1 : r - 1;
    }
    const size_t q = row + j;
    const double uc = u[q], vc = v[q];
    ru[q] = (float)((uc - (double)up[q]) * inv_dt + (double)ru[q] + vc * uy - nu * uyy);
    rv[q] = (float)((vc - (double)vp[q]) * inv_dt + (double)rv[q] + vc * vy + py * inv_rho - nu * vyy);
    rd[q] = (float)((double)rd[q] + vy);
}

// backward, axis 0:  GU = a u_x + b v_x - [D_x(a u + d) + nu a_xx],  GV = -[D_x(b u) + nu b_xx],  GP = -(D_x a)/rho
__global__ __launch_bounds__(kBx * kBy) void dense_bwd_x(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ ga,
                                                          const float* __restrict__ gb, const float* __restrict__ gd,
                                                          float* __restrict__ gu, float* __restrict__ gv, float* __restrict__ gp,
                                                          int nx, int ny, Circ c, double inv_rho, double nu) {
    const int j = blockIdx.x * kBx + threadIdx.x, i = blockIdx.y * kBy + threadIdx.y;
    if (j >= ny || i >= nx) return;
    const size_t g = (size_t)blockIdx.z * nx * ny;
    double ux = 0, vx = 0, s1 = 0, s2 = 0, ax = 0, axx = 0, bxx = 0;
    int r = i;
    for (int m = 0; m < nx; ++m) {
        const size_t q = g + (size_t)r * ny + j;
        const double d1 = c.d1[m], d2 = c.d2[m], uu = u[q], vv = v[q], a = ga[q], b = gb[q];
        ux += d1 * uu; vx += d1 * vv; s1 += d1 * (a * uu + (double)gd[q]); s2 += d1 * (b * uu); ax += d1 * a; axx += d2 * a; bxx += d2 * b;
        r = r == 0 ? nx - 1 : r - 1;
    }
    const size_t q = g + (size_t)i * ny + j;
    gu[q] = (float)((double)ga[q] * ux + (double)gb[q] * vx - (s1 + nu * axx));
    gv[q] = (float)(-(s2 + nu * bxx));
    gp[q] = (float)(-ax * inv_rho);
}

// backward, axis 1:  grad_u = GU + a/dt - [D_y(a v) + nu a_yy],  grad_v = GV + b/dt + a u_y + b v_y - [D_y(b v + d) + nu b_yy],
//                    grad_p = GP - (D_y b)/rho,  grad_u_prev = -a/dt,  grad_v_prev = -b/dt (optional)
__global__ __launch_bounds__(kBx * kBy) void dense_bwd_y(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ ga,
                                                          const float* __restrict__ gb, const float* __restrict__ gd,
                                                          float* __restrict__ gu, float* __restrict__ gv, float* __restrict__ gp,
                                                          float* __restrict__ gup, float* __restrict__ gvp,
                                                          int nx, int ny, Circ c, double inv_rho, double nu, double inv_dt) {
    const int j = blockIdx.x * kBx + threadIdx.x, i = blockIdx.y * kBy + threadIdx.y;
    if (j >= ny || i >= nx) return;
    const size_t row = (size_t)blockIdx.z * nx * ny + (size_t)i * ny;
    double uy = 0, vy = 0, s1 = 0, s2 = 0, by = 0, ayy = 0, byy = 0;
    int r = j;
    for (int m = 0; m < ny; ++m) {
        const double d1 = c.d1[m], d2 = c.d2[m], uu = u[row + r], vv = v[row + r], a = ga[row + r], b = gb[row + r];
        uy += d1 * uu; vy += d1 * vv; s1 += d1 * (a * vv); s2 += d1 * (b * vv + (double)gd[row + r]); by += d1 * b; ayy += d2 * a; byy += d2 * b;
        r = r == 0 ? ny - 1 : r - 1;
    }
    const size_t q = row + j;
    const double a = ga[q], b = gb[q];
    gu[q] = (float)((double)gu[q] + a * inv_dt - (s1 + nu * ayy));
    gv[q] = (float)((double)gv[q] + b * inv_dt + a * uy + b * vy - (s2 + nu * byy));
    gp[q] = (float)((double)gp[q] - by * inv_rho);
    if (gup) gup[q] = (float)(-a * inv_dt);
    if (gvp) gvp[q] = (float)(-b * inv_dt);
}

dim3 dense_grid(int batch, int nx, int ny) { return dim3((unsigned)((ny + kBx - 1) / kBx), (unsigned)((nx + kBy - 1) / kBy), (unsigned)batch); }

int dense_check(const char* what, int n, int batch, int other) {
    if (n < 3 || n > kDenseMax)
        return fail(NNS_ERR_UNSUPPORTED, "%s: axis length %d: the FFT engine takes powers of two in [64, 1024], the dense (circulant) fallback 3 .. %d", what, n, kDenseMax);
    if (batch > 65535 || (other + kBy - 1) / kBy > 65535) return fail(NNS_ERR_UNSUPPORTED, "%s (dense fallback): batch %d / cross axis %d too large for one launch", what, batch, other);
    return NNS_OK;
}

}  // namespace

namespace nns {
namespace spec {

int dense_xpass(const float* u, const float* v, const float* p, float* ru, float* rv, float* rd, int batch, int nx, int ny,
                double Lx, double rho, double nu, hipStream_t s) {
    if (int rc = dense_check("spec_residual_xpass", nx, batch, nx)) return rc;
    Circ c;
    if (int rc = circulant(nx, Lx, s, c)) return rc;
    hipLaunchKernelGGL(dense_fwd_x, dense_grid(batch, nx, ny), dim3(kBx, kBy), 0, s, u, v, p, ru, rv, rd, nx, ny, c, 1.0 / rho, nu);
    return check_launch("spec_residual_xpass (dense)");
}

int dense_ypass(const float* u, const float* v, const float* p, const float* up, const float* vp, float* ru, float* rv, float* rd,
                int batch, int nx, int ny, double dt, double Ly, double rho, double nu, hipStream_t s) {
    if (int rc = dense_check("spec_residual_ypass", ny, batch, nx)) return rc;
    Circ c;
    if (int rc = circulant(ny, Ly, s, c)) return rc;
    hipLaunchKernelGGL(dense_fwd_y, dense_grid(batch, nx, ny), dim3(kBx, kBy), 0, s, u, v, p, up, vp, ru, rv, rd, nx, ny, c, 1.0 / rho, nu, 1.0 / dt);
    return check_launch("spec_residual_ypass (dense)");
}

int dense_bwd_xpass(const float* u, const float* v, const float* ga, const float* gb, const float* gd, float* gu, float* gv, float* gp,
                    int batch, int nx, int ny, double Lx, double rho, double nu, hipStream_t s) {
    if (int rc = dense_check("spec_residual_bwd (x)", nx, batch, nx)) return rc;
    Circ c;
    if (int rc = circulant(nx, Lx, s, c)) return rc;
    hipLaunchKernelGGL(dense_bwd_x, dense_grid(batch, nx, ny), dim3(kBx, kBy), 0, s, u, v, ga, gb, gd, gu, gv, gp, nx, ny, c, 1.0 / rho, nu);
    return check_launch("spec_residual_bwd xpass (dense)");
}

int dense_bwd_ypass(const float* u, const float* v, const float* ga, const float* gb, const float* gd, float* gu, float* gv, float* gp,
                    float* gup, float* gvp, int batch, int nx, int ny, double dt, double Ly, double rho, double nu, hipStream_t s) {
    if (int rc = dense_check("spec_residual_bwd (y)", ny, batch, nx)) return rc;
    Circ c;
    if (int rc = circulant(ny, Ly, s, c)) return rc;
    hipLaunchKernelGGL(dense_bwd_y, dense_grid(batch, nx, ny), dim3(kBx, kBy), 0, s, u, v, ga, gb, gd, gu, gv, gp, gup, gvp, nx, ny, c, 1.0 / rho, nu, 1.0 / dt);
    return check_launch("spec_residual_bwd ypass (dense)");
}

}  // namespace spec
}  // namespace nns

// Builds (or finds) the circulant table of one axis length on the CURRENT device ahead of time: the only part of the dense path that allocates
// and synchronises.  Call it once per (n, L) before capturing a stream that evaluates the residual on such an axis.
NNS_API int nns_spec_dense_warmup(int n, double L) {
    if (n < 3 || n > nns::spec::kDenseMaxLen || L == 0) return nns::fail(NNS_ERR_INVALID_ARG, "spec_dense_warmup: n=%d must be in [3, %d] and L non-zero", n, nns::spec::kDenseMaxLen);
    Circ c;
    return circulant(n, L, nullptr, c);
}
