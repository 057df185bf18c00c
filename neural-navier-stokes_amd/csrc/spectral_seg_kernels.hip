// Fourier-spectral residual on a SLAB-DECOMPOSED grid (nns/slab.py; SURVEY.md section 8 (e), BASELINE config 4): the passes that read and
// write the all-to-all buffers IN PLACE, so that no permuting copy stands between a collective and a kernel (kernels: spectral_fwd.h).
//
//   rank r owns rows [r nx/P, (r+1) nx/P) of every grid.  all-to-all #1 delivers [src][u, v, p][grid][nx/P rows][ny/P columns]:
//     nns_spec_residual_xpass_seg_f32      the column pass on that buffer (a column's rows in blocks of nx/P per source rank), partials written
//                                          in the same layout = the send buffer of all-to-all #2;
//   all-to-all #2 returns [src][P_u, P_v, P_d][grid][nx/P rows][ny/P columns] -- a ROW of a local grid is now P pieces of ny/P floats:
//     nns_residual_both_rowpass_halo_seg_f32   the fused row pass (stencil + spectral finish) reading its partials there (round 4: before, a
//     nns_spec_residual_ypass_seg_f32          copy kernel scattered them into row slabs first: 12 B/pt of HBM traffic and a launch per chunk)
//
// The reference has one device (src/neural_spectral/spectral_ode.py:155-156,165); the decomposition is the north star's.
#include "spectral_fwd.h"

using namespace nns;
using namespace nns::spec;

namespace {

// seg_cols = ny / P (columns per source rank): a power of two, at least one line's lane count (N / 16), so that every register slot of a
// line -- TPF consecutive columns starting at a multiple of TPF -- lies inside one source rank's piece
int check_part(const char* what, const float* pu, const float* pv, const float* pd, int ny, int seg_cols, long seg_stride, int batch, int nx, PartK& pk) {
    if (!pu || !pv || !pd) return fail(NNS_ERR_INVALID_ARG, "%s: the three partial fields are required", what);
    if (!pow2_in_range(ny)) return fail(NNS_ERR_UNSUPPORTED, "%s: ny=%d must be a power of two in [64, 1024] (the segmented layout exists for the FFT engine only)", what, ny);
    if (seg_cols < ny / 16 || seg_cols > ny || (seg_cols & (seg_cols - 1)))
        return fail(NNS_ERR_INVALID_ARG, "%s: seg_cols=%d must be a power of two in [ny / 16 = %d, ny = %d]", what, seg_cols, ny / 16, ny);
    const long block = (long)batch * nx * seg_cols;                       // one field of one source rank
    if (seg_stride < block) return fail(NNS_ERR_INVALID_ARG, "%s: seg_stride=%ld must be >= batch * nx * seg_cols = %ld", what, seg_stride, block);
    // 32-bit byte offsets from the field's first block: row offset (< one block) + piece offset (< ny / seg_cols blocks of seg_stride elements)
    const long span = ((long)(ny / seg_cols) - 1) * seg_stride + block;
    if (span >= (1L << 30)) return fail(NNS_ERR_UNSUPPORTED, "%s: the partial buffer spans %ld elements per field from its first block: beyond the 32-bit byte offsets of the segmented reads", what, span);
    pk = PartK{pu, pv, pd, __builtin_ctz((unsigned)seg_cols), (unsigned)(seg_stride * 4)};
    return NNS_OK;
}

}  // namespace

#define S(stream) reinterpret_cast<hipStream_t>(stream)

NNS_API int nns_spec_residual_xpass_seg_f32(const float* u, const float* v, const float* p, float* r_u, float* r_v, float* r_div,
                                            int batch, int nx, int ny, int seg_rows, long seg_stride, double Lx, double rho, double nu, int precise, void* stream) {
    if (seg_rows < 1) return fail(NNS_ERR_INVALID_ARG, "spec_residual_xpass_seg: seg_rows must be >= 1");
    if (!u || !v || !p || !r_u || !r_v || !r_div || batch < 1 || ny < 1) return fail(NNS_ERR_INVALID_ARG, "spec_residual_xpass_seg: bad args");
    if (Lx == 0 || rho == 0) return fail(NNS_ERR_INVALID_ARG, "spec_residual_xpass_seg: Lx, rho must be non-zero");
    if (!pow2_in_range(nx)) return fail(NNS_ERR_UNSUPPORTED, "spec_residual_xpass_seg: nx=%d must be a power of two in [64, 1024] (the segmented layout exists for the FFT engine only)", nx);
    if (seg_rows < 4 || seg_rows > nx || (seg_rows & (seg_rows - 1)) || seg_stride < (long)seg_rows * ny)
        return fail(NNS_ERR_INVALID_ARG, "spec_residual_xpass_seg: seg_rows=%d must be a power of two in [4, nx=%d] and seg_stride=%ld >= seg_rows * ny", seg_rows, nx, seg_stride);
    const double ks = 2.0 * M_PI / Lx;
    const SpecK k{ks / nx, ks / (rho * nx), nu * ks * ks / nx, 0.f};
    const bool f64 = !spec_f32_mode(precise, nu, nx, Lx);
    const SegK sg{__builtin_ctz((unsigned)seg_rows), seg_stride};
    return dispatch_n(nx, [&](auto n) {
        constexpr int N = decltype(n)::value;
        return f64 ? launch_xpass<N, double, true>(u, v, p, r_u, r_v, r_div, batch, ny, k, S(stream), sg)
                   : launch_xpass<N, float, true>(u, v, p, r_u, r_v, r_div, batch, ny, k, S(stream), sg);
    });
}

NNS_API int nns_spec_residual_ypass_seg_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                            const float* part_u, const float* part_v, const float* part_div, int seg_cols, long seg_stride,
                                            float* r_u, float* r_v, float* r_div, int batch, int nx, int ny, double dt, double Ly,
                                            double rho, double nu, int precise, void* stream) {
    if (!u || !v || !p || !u_prev || !v_prev || !r_u || !r_v || !r_div || batch < 1 || nx < 1) return fail(NNS_ERR_INVALID_ARG, "spec_residual_ypass_seg: bad args");
    if (Ly == 0 || rho == 0 || dt == 0) return fail(NNS_ERR_INVALID_ARG, "spec_residual_ypass_seg: Ly, rho, dt must be non-zero");
    PartK pk;
    if (int rc = check_part("spec_residual_ypass_seg", part_u, part_v, part_div, ny, seg_cols, seg_stride, batch, nx, pk)) return rc;
    const double ks = 2.0 * M_PI / Ly;
    const SpecK k{ks / ny, ks / (rho * ny), nu * ks * ks / ny, (float)(1.0 / dt)};
    const long nrows = (long)batch * nx;
    const bool f64 = !spec_f32_mode(precise, nu, ny, Ly);
    return dispatch_n(ny, [&](auto n) {
        constexpr int N = decltype(n)::value;
        return f64 ? launch_ypass<N, double, false, true>(u, v, p, u_prev, v_prev, r_u, r_v, r_div, nrows, k, S(stream), nullptr, nullptr, nullptr, 1, FdK{}, HaloK{}, pk)
                   : launch_ypass<N, float, false, true>(u, v, p, u_prev, v_prev, r_u, r_v, r_div, nrows, k, S(stream), nullptr, nullptr, nullptr, 1, FdK{}, HaloK{}, pk);
    });
}

NNS_API int nns_residual_both_rowpass_halo_seg_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                                   const float* halo_top, const float* halo_bot,
                                                   const float* part_u, const float* part_v, const float* part_div, int seg_cols, long seg_stride,
                                                   float* fd_r_u, float* fd_r_v, float* fd_r_div, float* sp_r_u, float* sp_r_v, float* sp_r_div,
                                                   int batch, int nx_local, int ny, long halo_field_stride, double dt, double dx, double Ly, double rho, double nu,
                                                   int precise, void* stream) {
    if (!u || !v || !p || !u_prev || !v_prev || !fd_r_u || !fd_r_v || !fd_r_div || !sp_r_u || !sp_r_v || !sp_r_div || batch < 1 || nx_local < 3)
        return fail(NNS_ERR_INVALID_ARG, "residual_both_rowpass_halo_seg: bad args");
    if (!halo_top || !halo_bot) return fail(NNS_ERR_INVALID_ARG, "residual_both_rowpass_halo_seg: halo_top and halo_bot are required");
    if (halo_field_stride != 0 && halo_field_stride < (long)batch * ny)
        return fail(NNS_ERR_INVALID_ARG, "residual_both_rowpass_halo_seg: halo_field_stride=%ld must be 0 (= batch * ny) or >= batch * ny = %ld", halo_field_stride, (long)batch * ny);
    if (Ly == 0 || rho == 0 || dt == 0 || dx == 0) return fail(NNS_ERR_INVALID_ARG, "residual_both_rowpass_halo_seg: dx, Ly, rho, dt must be non-zero");
    PartK pk;
    if (int rc = check_part("residual_both_rowpass_halo_seg", part_u, part_v, part_div, ny, seg_cols, seg_stride, batch, nx_local, pk)) return rc;
    const int nx = nx_local;
    const double ks = 2.0 * M_PI / Ly, dy = Ly / ny;
    const SpecK k{ks / ny, ks / (rho * ny), nu * ks * ks / ny, (float)(1.0 / dt)};
    const FdK fk{(float)(1.0 / (2 * dx)), (float)(1.0 / (2 * dy)), (float)(1.0 / rho), (float)nu, 1.0 / (dx * dx), 1.0 / (dy * dy),
                 (float)(1.0 / (dx * dx)), (float)(1.0 / (dy * dy))};
    const HaloK hk{halo_top, halo_bot, halo_field_stride > 0 ? halo_field_stride : (long)batch * ny};
    return dispatch_n(ny, [&](auto n) {
        constexpr int N = decltype(n)::value;
        const long nrows = (long)batch * nx;
        if (!spec_f32_mode(precise, nu, ny, Ly))
            return launch_ypass<N, double, true, true>(u, v, p, u_prev, v_prev, sp_r_u, sp_r_v, sp_r_div, nrows, k, S(stream), fd_r_u, fd_r_v, fd_r_div, nx, fk, hk, pk);
        return launch_rowmarch<N, true>(u, v, p, u_prev, v_prev, sp_r_u, sp_r_v, sp_r_div, fd_r_u, fd_r_v, fd_r_div, batch, nx, march_chunk_rows<N>(nrows, nx), k, fk, hk, S(stream), pk);
    });
}
