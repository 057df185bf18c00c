// Fourier-spectral back-end of the periodic Navier-Stokes residual, gfx950: the whole-grid entry points (kernels: spectral_fwd.h).
// Operator definition: oracle/periodic.py (spectral_residual); no reference symbol exists.
#include "spectral_fwd.h"

using namespace nns;
using namespace nns::spec;

namespace {

int xpass(const float* u, const float* v, const float* p, float* ru, float* rv, float* rd, int batch, int nx, int ny,
          double Lx, double rho, double nu, int precise, hipStream_t s) {
    if (!u || !v || !p || !ru || !rv || !rd || batch < 1 || ny < 1) return fail(NNS_ERR_INVALID_ARG, "spec_residual_xpass: bad args");
    if (Lx == 0 || rho == 0) return fail(NNS_ERR_INVALID_ARG, "spec_residual_xpass: Lx, rho must be non-zero");
    if (!pow2_in_range(nx)) return dense_xpass(u, v, p, ru, rv, rd, batch, nx, ny, Lx, rho, nu, s);          // any other length 3 .. 2048: circulant matrices, float64
    const double ks = 2.0 * M_PI / Lx;
    SpecK k{ks / nx, ks / (rho * nx), nu * ks * ks / nx, 0.f};
    const bool f64 = !spec_f32_mode(precise, nu, nx, Lx);
    return dispatch_n(nx, [&](auto n) {
        constexpr int N = decltype(n)::value;
        return f64 ? launch_xpass<N, double>(u, v, p, ru, rv, rd, batch, ny, k, s)
                   : launch_xpass<N, float>(u, v, p, ru, rv, rd, batch, ny, k, s);
    });
}

int ypass(const float* u, const float* v, const float* p, const float* up, const float* vp, float* ru, float* rv, float* rd,
          int batch, int nx, int ny, double dt, double Ly, double rho, double nu, int precise, hipStream_t s) {
    if (!u || !v || !p || !up || !vp || !ru || !rv || !rd || batch < 1 || nx < 1) return fail(NNS_ERR_INVALID_ARG, "spec_residual_ypass: bad args");
    if (Ly == 0 || rho == 0 || dt == 0) return fail(NNS_ERR_INVALID_ARG, "spec_residual_ypass: Ly, rho, dt must be non-zero");
    if (!pow2_in_range(ny)) return dense_ypass(u, v, p, up, vp, ru, rv, rd, batch, nx, ny, dt, Ly, rho, nu, s);
    const double ks = 2.0 * M_PI / Ly;
    SpecK k{ks / ny, ks / (rho * ny), nu * ks * ks / ny, (float)(1.0 / dt)};
    const long nrows = (long)batch * nx;
    const bool f64 = !spec_f32_mode(precise, nu, ny, Ly);
    return dispatch_n(ny, [&](auto n) {
        constexpr int N = decltype(n)::value;
        return f64 ? launch_ypass<N, double>(u, v, p, up, vp, ru, rv, rd, nrows, k, s)
                   : launch_ypass<N, float>(u, v, p, up, vp, ru, rv, rd, nrows, k, s);
    });
}

// FD 5-point + spectral residual of the same inputs: spectral x-pass, then the row pass with the stencil fused in.
int residual_both(const float* u, const float* v, const float* p, const float* up, const float* vp, float* fu, float* fv, float* fd,
                  float* ru, float* rv, float* rd, int batch, int nx, int ny, double dt, double Lx, double Ly, double rho, double nu,
                  int precise, hipStream_t s, bool with_xpass, const float* halo_top = nullptr, const float* halo_bot = nullptr, long halo_fstride = 0) {
    if (!u || !v || !p || !up || !vp || !fu || !fv || !fd || !ru || !rv || !rd || batch < 1 || nx < 3)
        return fail(NNS_ERR_INVALID_ARG, "residual_both: bad args");
    const bool slab = halo_top || halo_bot;               // a row slab: nx is the LOCAL row count (any value >= 3), only ny is transformed here
    if (slab && (!halo_top || !halo_bot || with_xpass)) return fail(NNS_ERR_INVALID_ARG, "residual_both: a row slab needs both halo messages and the row pass only");
    if (Ly == 0 || rho == 0 || dt == 0 || Lx == 0) return fail(NNS_ERR_INVALID_ARG, "residual_both: Lx, Ly, rho, dt must be non-zero");
    if (with_xpass) precise = spec_resolve_precise(precise, nu, nx, Lx, ny, Ly);       // a whole evaluation: one arithmetic for both passes
    if (!slab && (!spec_len_ok(nx) || !spec_len_ok(ny)))
        return fail(NNS_ERR_UNSUPPORTED, "residual_both: nx=%d, ny=%d: powers of two in [64, 1024] (FFT engine) or any length 3 .. %d (dense fallback)", nx, ny, kDenseMaxLen);
    if (!pow2_in_range(ny) && !slab) {
        // rows the FFT engine does not serve: no fused row pass -- the standalone stencil kernel, then the spectral passes (dense where needed)
        if (int rc = nns_fd_residual_f32(u, v, p, up, vp, fu, fv, fd, batch, nx, ny, dt, Lx / nx, Ly / ny, rho, nu, 5, s)) return rc;
        if (with_xpass) { if (int rc = xpass(u, v, p, ru, rv, rd, batch, nx, ny, Lx, rho, nu, precise, s)) return rc; }
        return ypass(u, v, p, up, vp, ru, rv, rd, batch, nx, ny, dt, Ly, rho, nu, precise, s);
    }
    if (!pow2_in_range(ny)) return fail(NNS_ERR_UNSUPPORTED, "residual_both (row slab): ny=%d must be a power of two in [64, 1024]", ny);
    if (with_xpass) {
        // NNS_BOTH_GROUP=G (a measurement switch, default off): the two launches per GROUP of G grids instead of per batch, so that a group's row pass
        // follows its column pass while the group's partials and inputs may still be in the 256 MB Infinity Cache.
        // Measured at 1024^2 x 64 (profiles/r04_grouped_launches.txt): slower at every G -- see DESIGN.md section 7.1.
        static const int group = [] { const char* e = getenv("NNS_BOTH_GROUP"); return e ? atoi(e) : 0; }();
        if (group > 0 && group < batch) {
            for (int g0 = 0; g0 < batch; g0 += group) {
                const int gb = batch - g0 < group ? batch - g0 : group;
                const size_t o = (size_t)g0 * nx * ny;
                if (int rc = residual_both(u + o, v + o, p + o, up + o, vp + o, fu + o, fv + o, fd + o, ru + o, rv + o, rd + o, gb, nx, ny, dt, Lx, Ly, rho, nu, precise, s, true)) return rc;
            }
            return NNS_OK;
        }
        if (int rc = xpass(u, v, p, ru, rv, rd, batch, nx, ny, Lx, rho, nu, precise, s)) return rc;
    }
    const double ks = 2.0 * M_PI / Ly, dx = slab ? Lx : Lx / nx, dy = Ly / ny;          // a row slab passes the grid spacing itself in Lx
    const SpecK k{ks / ny, ks / (rho * ny), nu * ks * ks / ny, (float)(1.0 / dt)};
    const FdK fk{(float)(1.0 / (2 * dx)), (float)(1.0 / (2 * dy)), (float)(1.0 / rho), (float)nu, 1.0 / (dx * dx), 1.0 / (dy * dy),
                 (float)(1.0 / (dx * dx)), (float)(1.0 / (dy * dy))};
    const HaloK hk{halo_top, halo_bot, halo_fstride > 0 ? halo_fstride : (long)batch * ny};
    return dispatch_n(ny, [&](auto n) {
        constexpr int N = decltype(n)::value;
        const long nrows = (long)batch * nx;
        if (!spec_f32_mode(precise, nu, ny, Ly)) return launch_ypass<N, double, true>(u, v, p, up, vp, ru, rv, rd, nrows, k, s, fu, fv, fd, nx, fk, hk);
        if constexpr (NNS_ROWMARCH) return launch_rowmarch<N>(u, v, p, up, vp, ru, rv, rd, fu, fv, fd, batch, nx, march_chunk_rows<N>(nrows, nx), k, fk, hk, s);
        else return launch_ypass<N, float, true>(u, v, p, up, vp, ru, rv, rd, nrows, k, s, fu, fv, fd, nx, fk, hk);
    });
}

}  // namespace

#define S(stream) reinterpret_cast<hipStream_t>(stream)

NNS_API int nns_residual_both_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                  float* fd_r_u, float* fd_r_v, float* fd_r_div, float* sp_r_u, float* sp_r_v, float* sp_r_div,
                                  int batch, int nx, int ny, double dt, double Lx, double Ly, double rho, double nu, int precise, void* stream) {
    return residual_both(u, v, p, u_prev, v_prev, fd_r_u, fd_r_v, fd_r_div, sp_r_u, sp_r_v, sp_r_div, batch, nx, ny, dt, Lx, Ly, rho, nu, precise, S(stream), true);
}
NNS_API int nns_residual_both_rowpass_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                          float* fd_r_u, float* fd_r_v, float* fd_r_div, float* sp_r_u, float* sp_r_v, float* sp_r_div,
                                          int batch, int nx, int ny, double dt, double Lx, double Ly, double rho, double nu, int precise, void* stream) {
    return residual_both(u, v, p, u_prev, v_prev, fd_r_u, fd_r_v, fd_r_div, sp_r_u, sp_r_v, sp_r_div, batch, nx, ny, dt, Lx, Ly, rho, nu, precise, S(stream), false);
}

NNS_API int nns_residual_both_rowpass_halo_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                               const float* halo_top, const float* halo_bot,
                                               float* fd_r_u, float* fd_r_v, float* fd_r_div, float* sp_r_u, float* sp_r_v, float* sp_r_div,
                                               int batch, int nx_local, int ny, long halo_field_stride, double dt, double dx, double Ly, double rho, double nu,
                                               int precise, void* stream) {
    if (!halo_top || !halo_bot) return fail(NNS_ERR_INVALID_ARG, "residual_both_rowpass_halo: halo_top and halo_bot are required");
    if (halo_field_stride != 0 && halo_field_stride < (long)batch * ny)
        return fail(NNS_ERR_INVALID_ARG, "residual_both_rowpass_halo: halo_field_stride=%ld must be 0 (= batch * ny) or >= batch * ny = %ld", halo_field_stride, (long)batch * ny);
    return residual_both(u, v, p, u_prev, v_prev, fd_r_u, fd_r_v, fd_r_div, sp_r_u, sp_r_v, sp_r_div, batch, nx_local, ny, dt, dx, Ly, rho, nu, precise,
                         S(stream), false, halo_top, halo_bot, halo_field_stride);
}
NNS_API int nns_spec_residual_xpass_f32(const float* u, const float* v, const float* p, float* r_u, float* r_v, float* r_div,
                                        int batch, int nx, int ny, double Lx, double rho, double nu, int precise, void* stream) {
    return xpass(u, v, p, r_u, r_v, r_div, batch, nx, ny, Lx, rho, nu, precise, S(stream));
}
NNS_API int nns_spec_residual_ypass_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                        float* r_u, float* r_v, float* r_div, int batch, int nx, int ny, double dt, double Ly,
                                        double rho, double nu, int precise, void* stream) {
    return ypass(u, v, p, u_prev, v_prev, r_u, r_v, r_div, batch, nx, ny, dt, Ly, rho, nu, precise, S(stream));
}
NNS_API int nns_spec_residual_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                  float* r_u, float* r_v, float* r_div, int batch, int nx, int ny, double dt, double Lx, double Ly,
                                  double rho, double nu, int precise, void* stream) {
    if (!spec_len_ok(nx) || !spec_len_ok(ny))                  // before the first launch: both axes must have an engine
        return fail(NNS_ERR_UNSUPPORTED, "spec_residual: nx=%d, ny=%d: powers of two in [64, 1024] (FFT engine) or any length 3 .. %d (dense fallback)", nx, ny, kDenseMaxLen);
    const int pr = spec_resolve_precise(precise, nu, nx, Lx, ny, Ly);       // one arithmetic for both passes
    if (int rc = xpass(u, v, p, r_u, r_v, r_div, batch, nx, ny, Lx, rho, nu, pr, S(stream))) return rc;
    return ypass(u, v, p, u_prev, v_prev, r_u, r_v, r_div, batch, nx, ny, dt, Ly, rho, nu, pr, S(stream));
}

// The arithmetic a `precise` request resolves to for a whole evaluation (0 = all-float32 transforms, 2 = float64 forward transforms): hosts that
// call the passes one by one -- the slab-decomposed path, nns/slab.py -- ask once and hand the answer to every pass, so that there is ONE policy
// implementation (NNS_SPEC_F64 included).
NNS_API int nns_spec_resolve_precise(int precise, double nu, int nx, double Lx, int ny, double Ly) {
    return spec_resolve_precise(precise, nu, nx, Lx, ny, Ly);
}
