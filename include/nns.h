/*
 * nns.h -- C ABI of libnns_hip.so, the MI355X (gfx950) residual engine behind the call surface of
 * mhw32/neural-navier-stokes (src/boundary.py, src/chorin_fd, src/direct_fd, src/chorin_spectral,
 * src/neural_spectral).  The reference is pure Python with no FFI of its own:
 * the entry points below are what a ctypes binding on the reference side would bind for each of
 * its operators; every declaration cites the reference symbol (file:line) it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the name ends in
 *     _host; buffers are caller-owned and caller-allocated (workspace sizes via the *_workspace
 *     query functions); the library never allocates or frees device memory in a launch function;
 *   - one call = one stream-ordered enqueue on `stream` (a hipStream_t; NULL = default stream),
 *     no host synchronisation unless the function says so;
 *   - fields are C-contiguous [batch, nx, ny]; element [b][i][j] at (b*nx + i)*ny + j;
 *     suffix _f32 / _f64 is the storage type of the fields (arithmetic is done in that type
 *     unless stated);
 *   - return value: 0 on success, a negative nns_status otherwise; never throws, never aborts;
 *     nns_last_error() returns a thread-local message for the last failure;
 *   - boundary lists (nns_bc_list) are small host structs passed by pointer and copied into
 *     kernel arguments.
 */
#ifndef NNS_H
#define NNS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNS_VERSION_MAJOR 0
#define NNS_VERSION_MINOR 1

typedef enum nns_status {
    NNS_OK = 0,
    NNS_ERR_INVALID_ARG = -1,   /* bad size / null pointer / unsupported combination       */
    NNS_ERR_UNSUPPORTED = -2,   /* valid request this build has no kernel for              */
    NNS_ERR_LAUNCH = -3,        /* HIP reported a launch/runtime error (see last_error)    */
    NNS_ERR_WORKSPACE = -4      /* workspace too small                                     */
} nns_status;

/* ---- boundary conditions: src/boundary.py:1-86 ------------------------------------------- */
enum { NNS_BC_DIRICHLET = 0, NNS_BC_NEUMANN = 1 };                 /* .type,  :32,:54          */
enum { NNS_SIDE_LEFT = 0,   /* A[0, :]   src/boundary.py:39-40 */
       NNS_SIDE_RIGHT = 1,  /* A[-1, :]  :41-42 */
       NNS_SIDE_BOTTOM = 2, /* A[:, 0]   :43-44 */
       NNS_SIDE_TOP = 3 };  /* A[:, -1]  :45-46 */
#define NNS_MAX_BC 8
typedef struct nns_bc_list {
    int32_t n;                       /* number of entries, applied in list order (corners!)   */
    int32_t kind[NNS_MAX_BC];
    int32_t side[NNS_MAX_BC];
    double value[NNS_MAX_BC];
    double dx[NNS_MAX_BC];           /* each BC object carries its own dx, dy (:23)            */
    double dy[NNS_MAX_BC];
} nns_bc_list;

const char* nns_last_error(void);
int nns_version(void);                                  /* major*1000 + minor                  */
/* Fills name (<= cap bytes), CU count, and HBM bytes of the current device. */
int nns_device_info(char* name_host, int cap, int* cu_count_host, size_t* hbm_bytes_host);

/* DirichletBoundaryCondition.apply / NeumannBoundaryCondition.apply (src/boundary.py:34-48,
 * :56-86) for a whole list, in list order, in place, on every grid of the batch. */
int nns_bc_apply_f32(float* A, int batch, int nx, int ny, const nns_bc_list* bcs_host, void* stream);
int nns_bc_apply_f64(double* A, int batch, int nx, int ny, const nns_bc_list* bcs_host, void* stream);

/* ---- chorin_fd: src/chorin_fd/simulate.py ------------------------------------------------ */
/* _explicit_predictor_step (:63-91): AB2 advection (both differences along axis 0 -- reference
 * quirk) + AB2 5-point diffusion on the interior, edges copied from un/vn.  out: ui, vi. */
int nns_fd_predictor_explicit_f32(const float* un, const float* vn, const float* un1, const float* vn1,
                                  float* ui, float* vi, int batch, int nx, int ny,
                                  double dt, double dx, double dy, double nu, void* stream);
int nns_fd_predictor_explicit_f64(const double* un, const double* vn, const double* un1, const double* vn1,
                                  double* ui, double* vi, int batch, int nx, int ny,
                                  double dt, double dx, double dy, double nu, void* stream);

/* _semi_implicit_predictor_step (:93-167): AB2 advection + Crank-Nicolson ADI; the dense
 * np.linalg.solve calls (:137,:153,:159,:165) become constant-coefficient Thomas solves, one
 * thread per column, BOTH along axis 0 as in the reference (needs nx == ny).  work: 4*batch*nx*ny
 * elements of scratch. */
size_t nns_fd_predictor_adi_workspace(int batch, int nx, int ny, int elem_size);
int nns_fd_predictor_adi_f32(const float* un, const float* vn, const float* un1, const float* vn1,
                             float* ui, float* vi, float* work, int batch, int nx, int ny,
                             double dt, double dx, double dy, double nu, void* stream);
int nns_fd_predictor_adi_f64(const double* un, const double* vn, const double* un1, const double* vn1,
                             double* ui, double* vi, double* work, int batch, int nx, int ny,
                             double dt, double dx, double dy, double nu, void* stream);

/* The same predictor on a COLUMN slab [nx][nyl] of a square nx x nx grid (halo / boundary columns 0 and nyl-1 are copied
 * from un like any edge): both solves run along axis 0, so column slabs need no communication for them -- the
 * multi-GPU form of the ADI step (nns/slab.py: SlabChorinFD(method='semi_implicit')).  Identical arithmetic; only
 * the nx == ny check is waived. */
int nns_fd_predictor_adi_colslab_f32(const float* un, const float* vn, const float* un1, const float* vn1,
                                     float* ui, float* vi, float* work, int batch, int nx, int nyl,
                                     double dt, double dx, double dy, double nu, void* stream);
int nns_fd_predictor_adi_colslab_f64(const double* un, const double* vn, const double* un1, const double* vn1,
                                     double* ui, double* vi, double* work, int batch, int nx, int nyl,
                                     double dt, double dx, double dy, double nu, void* stream);

/* dx2dy2C of _get_pressure (:186-188): backward-difference divergence RHS, zero on the edge. */
int nns_fd_pressure_rhs_f32(const float* ui, const float* vi, float* C, int batch, int nx, int ny,
                            double dt, double dx, double dy, double rho, void* stream);
int nns_fd_pressure_rhs_f64(const double* ui, const double* vi, double* C, int batch, int nx, int ny,
                            double dt, double dx, double dy, double rho, void* stream);

/* The SOR loop of _get_pressure (:190-200): in-place lexicographic Gauss-Seidel with
 * over-relaxation, evaluated in anti-diagonal wavefront order (bitwise-identical update order),
 * several sweeps in flight skewed by two fronts.  Stops after the first sweep whose
 * max|p - pPrev| <= tol, or after max_sweeps (= nit-1) sweeps.  p is updated in place.
 * info[2*b] = sweeps done, info[2*b+1] = last err (as the field type), per grid b.
 * work: nns_fd_sor_workspace() bytes. One workgroup per grid ("replicas only" across GPUs). */
size_t nns_fd_sor_workspace(int batch, int nx, int ny, int elem_size);
/* (The sweeps run speculatively in batches; a stop inside a batch costs a restore + replay.  nns_fd_sor_hint_* take `hint` [batch][2] -- the info of the
 * PREVIOUS solve of the same grids in a time loop, device memory, may be the same buffer as info, NULL = none -- and size the first batch by its sweep
 * count: a scheduling hint only, the results do not depend on it.) */
int nns_fd_sor_f32(float* p, const float* C, float* info, void* work, int batch, int nx, int ny,
                   double dx, double dy, double beta, double tol, int max_sweeps, void* stream);
int nns_fd_sor_f64(double* p, const double* C, double* info, void* work, int batch, int nx, int ny,
                   double dx, double dy, double beta, double tol, int max_sweeps, void* stream);
int nns_fd_sor_hint_f32(float* p, const float* C, float* info, const float* hint, void* work, int batch, int nx, int ny,
                        double dx, double dy, double beta, double tol, int max_sweeps, void* stream);
int nns_fd_sor_hint_f64(double* p, const double* C, double* info, const double* hint, void* work, int batch, int nx, int ny,
                        double dx, double dy, double beta, double tol, int max_sweeps, void* stream);

/* ---- corrected / extended solver options (NOT reference behaviour) ----------------
 * Explicit predictor with the true y-advection  v d/dy  (the reference differences along x twice, :73-76,:82-85);
 * oracle: oracle/chorin_fd.py explicit_predictor_corrected. */
int nns_fd_predictor_explicit_corrected_f32(const float* un, const float* vn, const float* un1, const float* vn1,
                                            float* ui, float* vi, int batch, int nx, int ny,
                                            double dt, double dx, double dy, double nu, void* stream);
int nns_fd_predictor_explicit_corrected_f64(const double* un, const double* vn, const double* un1, const double* vn1,
                                            double* ui, double* vi, int batch, int nx, int ny,
                                            double dt, double dx, double dy, double nu, void* stream);
/* Semi-implicit (ADI) predictor with the second tridiagonal solve along axis 1, as an ADI scheme intends (the reference
 * runs both along axis 0 and therefore needs nx == ny); oracle: semi_implicit_predictor_corrected.  Same workspace as
 * nns_fd_predictor_adi; nx != ny allowed. */
int nns_fd_predictor_adi_corrected_f32(const float* un, const float* vn, const float* un1, const float* vn1,
                                       float* ui, float* vi, float* work, int batch, int nx, int ny,
                                       double dt, double dx, double dy, double nu, void* stream);
int nns_fd_predictor_adi_corrected_f64(const double* un, const double* vn, const double* un1, const double* vn1,
                                       double* ui, double* vi, double* work, int batch, int nx, int ny,
                                       double dt, double dx, double dy, double nu, void* stream);
/* Red-black SOR: the update formula, relaxation factor, stopping rule (max|p - pPrev| <= tol) and sweep cap of
 * nns_fd_sor, with the points of one colour ((i + j) even, then odd) relaxed in parallel: two barriers per sweep
 * instead of nx + ny fronts.  Same info layout.  oracle: get_pressure_redblack (bitwise).
 * Grids whose p and C fit LDS (2 nx ny elements <= 150 KB) are solved by one workgroup each, work may be NULL.
 * Larger grids run every half-sweep as a chip-wide launch: all 2 max_sweeps launches are enqueued at once and turn
 * themselves off on the device once a sweep's error is <= tol (no host round trip); they need
 * nns_fd_sor_redblack_workspace(batch, nx, ny, elem_size, max_sweeps) bytes of device memory in `work` (0 = fits LDS). */
size_t nns_fd_sor_redblack_workspace(int batch, int nx, int ny, int elem_size, int max_sweeps);
int nns_fd_sor_redblack_f32(float* p, const float* C, float* info, void* work, int batch, int nx, int ny,
                            double dx, double dy, double beta, double tol, int max_sweeps, void* stream);
int nns_fd_sor_redblack_f64(double* p, const double* C, double* info, void* work, int batch, int nx, int ny,
                            double dx, double dy, double beta, double tol, int max_sweeps, void* stream);

/* One red-black HALF-sweep on a row slab p[nxl][ny] whose rows 0 and nxl-1 are halo / boundary rows (not written);
 * local row i is global row gi0 + i (that fixes the colours).  Multi-workgroup.  *err_bits (4 bytes for f32, 8 for f64,
 * zeroed by the caller) receives max|p_new - p_old| as an IEEE bit pattern via an unsigned atomic max.  Building block
 * of the sharded pressure solve (nns/slab.py: SlabPressure) and of red-black SOR on grids too large for LDS. */
int nns_fd_sor_redblack_halfsweep_f32(float* p, const float* C, void* err_bits, int nxl, int ny, int gi0, int colour,
                                      double dx, double dy, double beta, void* stream);
int nns_fd_sor_redblack_halfsweep_f64(double* p, const double* C, void* err_bits, int nxl, int ny, int gi0, int colour,
                                      double dx, double dy, double beta, void* stream);
/* The same half-sweep as one link of a pre-enqueued chain: it runs only if *prev_err_bits (the previous sweep's error, same
 * bit-pattern format, already max-reduced over the ranks by the caller) is > tol -- the reference's `while err > tol`
 * (src/chorin_fd/simulate.py:190) evaluated on the device; a skipped colour-1 half-sweep stores the all-ones-but-sign (NaN)
 * pattern in *err_bits so that everything after it stays off.  No host round trip per sweep. */
int nns_fd_sor_redblack_halfsweep_gated_f32(float* p, const float* C, void* err_bits, const void* prev_err_bits, double tol,
                                            int nxl, int ny, int gi0, int colour, double dx, double dy, double beta, void* stream);
int nns_fd_sor_redblack_halfsweep_gated_f64(double* p, const double* C, void* err_bits, const void* prev_err_bits, double tol,
                                            int nxl, int ny, int gi0, int colour, double dx, double dy, double beta, void* stream);

/* spatial_coarsen (src/utils.py:13-60): block means over agg_x x agg_y cells of the [nt][nx][ny] sequences u, v, p into
 * [nt][nx/agg_x][ny/agg_y], one launch for the three fields.  The add order of numpy.mean (pairwise) is reproduced, so
 * float64 results are bit-identical to the reference's.  jfill = number of coarse COLUMNS the reference fills per row:
 * its column loop runs to ny // agg_x (:49), cells beyond stay 0; pass ny / agg_y for the plain block mean. */
int nns_coarsen_f32(const float* u, const float* v, const float* p, float* cu, float* cv, float* cp, int nt, int nx, int ny,
                    int agg_x, int agg_y, int jfill, void* stream);
int nns_coarsen_f64(const double* u, const double* v, const double* p, double* cu, double* cv, double* cp, int nt, int nx, int ny,
                    int agg_x, int agg_y, int jfill, void* stream);

/* step (:212-234) of the EXPLICIT method as ONE launch, one workgroup per grid: _explicit_predictor_step, the velocity boundary lists, _get_pressure
 * (right-hand side + the lexicographic SOR solve, tol / max_sweeps as nns_fd_sor_*), the pressure boundary list, _correction_step -- bitwise the
 * sequence nns_fd_predictor_explicit(_corrected) / nns_bc_apply x 2 / nns_fd_pressure_rhs / nns_fd_sor / nns_bc_apply / nns_fd_correction (the same
 * per-point functions).  p is updated in place (as the reference mutates it) and also written to p_copy when that is not NULL (a trajectory slot);
 * u_out, v_out receive the new velocities (they hold the intermediate ones on the way: they must not alias an input field); info [batch][2] =
 * (sweeps, last err); hint: as nns_fd_sor_hint_* (the previous step's info or NULL); work: nns_fd_sor_workspace(batch, nx, ny, elem) bytes.  Applies when p and its right-hand side fit one workgroup's LDS
 * (nns_fd_step_explicit_fits: 64 x 64 float64, 96 x 96 float32 and below); NNS_ERR_UNSUPPORTED otherwise: use the separate calls. */
int nns_fd_step_explicit_fits(int nx, int ny, int elem_size);
int nns_fd_step_explicit_f32(const float* un, const float* vn, const float* un1, const float* vn1, float* p, const nns_bc_list* u_bc, const nns_bc_list* v_bc,
                             const nns_bc_list* p_bc, float* u_out, float* v_out, float* p_copy, float* info, const float* hint, void* work, int batch, int nx, int ny,
                             double dt, double dx, double dy, double rho, double nu, double beta, double tol, int max_sweeps, int corrected, void* stream);
int nns_fd_step_explicit_f64(const double* un, const double* vn, const double* un1, const double* vn1, double* p, const nns_bc_list* u_bc, const nns_bc_list* v_bc,
                             const nns_bc_list* p_bc, double* u_out, double* v_out, double* p_copy, double* info, const double* hint, void* work, int batch, int nx, int ny,
                             double dt, double dx, double dy, double rho, double nu, double beta, double tol, int max_sweeps, int corrected, void* stream);

/* _correction_step (:204-210): u = u* - dt/(2dx) d0x p, v = v* - dt/(2dy) d0y p; edges from u*. */
int nns_fd_correction_f32(const float* ui, const float* vi, const float* p, float* u, float* v,
                          int batch, int nx, int ny, double dt, double dx, double dy, void* stream);
int nns_fd_correction_f64(const double* ui, const double* vi, const double* p, double* u, double* v,
                          int batch, int nx, int ny, double dt, double dx, double dy, void* stream);

/* ---- direct_fd: src/direct_fd/simulate.py (axis 1 = x there) ------------------------------- */
/* _build_up_b (:56-66). */
int nns_fd_build_b_f32(const float* u, const float* v, float* b, int batch, int nx, int ny,
                       double dt, double dx, double dy, double rho, void* stream);
int nns_fd_build_b_f64(const double* u, const double* v, double* b, int batch, int nx, int ny,
                       double dt, double dx, double dy, double rho, void* stream);
/* _pressure_poisson (:68-88): exactly nit Jacobi sweeps, the p BC list applied after every sweep.
 * p is updated in place; tmp is a scratch field of the same size (ping-pong). */
int nns_fd_jacobi_f32(float* p, float* tmp, const float* b, int batch, int nx, int ny,
                      double dx, double dy, int nit, const nns_bc_list* p_bc_host, void* stream);
int nns_fd_jacobi_f64(double* p, double* tmp, const double* b, int batch, int nx, int ny,
                      double dx, double dy, int nit, const nns_bc_list* p_bc_host, void* stream);
/* The u, v update of step (:98-118): upwind advection, central grad p / (2 rho), 5-point
 * diffusion.  un, vn -> u, v (edges copied); must not alias. */
int nns_fd_direct_update_f32(const float* un, const float* vn, const float* p, float* u, float* v,
                             int batch, int nx, int ny, double dt, double dx, double dy,
                             double rho, double nu, void* stream);
int nns_fd_direct_update_f64(const double* un, const double* vn, const double* p, double* u, double* v,
                             int batch, int nx, int ny, double dt, double dx, double dy,
                             double rho, double nu, void* stream);

/* ---- periodic-box Navier-Stokes residual (north-star operators; no reference symbol, SURVEY.md
 *      section 8 row a17; defined by oracle/periodic.py) --------------------------------------- */
/* r_u = (u-u_prev)/dt + u u_x + v u_y + p_x/rho - nu lap u ; r_v likewise ; r_div = u_x + v_y.
 * FD back-end: central differences, stencil = 5 or 9 (Mehrstellen) point Laplacian. */
int nns_fd_residual_f32(const float* u, const float* v, const float* p, const float* u_prev,
                        const float* v_prev, float* r_u, float* r_v, float* r_div,
                        int batch, int nx, int ny, double dt, double dx, double dy,
                        double rho, double nu, int stencil, void* stream);
int nns_fd_residual_f64(const double* u, const double* v, const double* p, const double* u_prev,
                        const double* v_prev, double* r_u, double* r_v, double* r_div,
                        int batch, int nx, int ny, double dt, double dx, double dy,
                        double rho, double nu, int stencil, void* stream);
/* The FD residual on local rows [row_begin, row_end) of a ROW SLAB [batch][nx_local][ny] (grids sharded by rows over ranks,
 * nns/slab.py): rows -1 and nx_local come from halo_top / halo_bot ([3 (u, v, p)][batch][ny] messages from the ring
 * neighbours).  Interior rows [1, nx_local-1) never touch the halos, so they can be launched while the exchange is in
 * flight and the two edge rows afterwards ("overlap interior compute with halo"). */
int nns_fd_residual_halo_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                             const float* halo_top, const float* halo_bot, float* r_u, float* r_v, float* r_div,
                             int batch, int nx_local, int ny, int row_begin, int row_end, double dt, double dx, double dy,
                             double rho, double nu, int stencil, void* stream);
int nns_fd_residual_halo_f64(const double* u, const double* v, const double* p, const double* u_prev, const double* v_prev,
                             const double* halo_top, const double* halo_bot, double* r_u, double* r_v, double* r_div,
                             int batch, int nx_local, int ny, int row_begin, int row_end, double dt, double dx, double dy,
                             double rho, double nu, int stencil, void* stream);
/* Vector-Jacobian product of the FD residual (the physics-informed loss' backward;
 * oracle/periodic.py: fd_residual_vjp).  g_* = dLoss/dr_*; outputs dLoss/du, /dv, /dp and, when the pointers are
 * non-null, dLoss/du_prev = -g_u/dt, dLoss/dv_prev = -g_v/dt.  Adjoint stencils (D^T = -D, L^T = L); p does not enter. */
int nns_fd_residual_bwd_f32(const float* u, const float* v, const float* g_u, const float* g_v, const float* g_div,
                            float* grad_u, float* grad_v, float* grad_p, float* grad_u_prev, float* grad_v_prev,
                            int batch, int nx, int ny, double dt, double dx, double dy,
                            double rho, double nu, int stencil, void* stream);
int nns_fd_residual_bwd_f64(const double* u, const double* v, const double* g_u, const double* g_v, const double* g_div,
                            double* grad_u, double* grad_v, double* grad_p, double* grad_u_prev, double* grad_v_prev,
                            int batch, int nx, int ny, double dt, double dx, double dy,
                            double rho, double nu, int stencil, void* stream);
/* Vector-Jacobian product of the spectral residual (oracle/periodic.py: spectral_residual_vjp): two fused passes like
 * the forward (columns, then rows), three packed forward (`precise` as for nns_spec_residual_f32) and three inverse (float32)
 * LDS-resident transforms per line -- the conjugate spectral multiplies.  grad_u_prev / grad_v_prev may be null.  Axis lengths as for
 * nns_spec_residual_f32: powers of two in [64, 1024] on the FFT engine, any other length 3 .. 2048 as circulant matrices in float64. */
int nns_spec_residual_bwd_f32(const float* u, const float* v, const float* g_u, const float* g_v, const float* g_div,
                              float* grad_u, float* grad_v, float* grad_p, float* grad_u_prev, float* grad_v_prev,
                              int batch, int nx, int ny, double dt, double Lx, double Ly, double rho, double nu,
                              int precise, void* stream);
/* Spectral back-end: d/dx <-> i kx, lap <-> -|k|^2 via LDS-resident 1-D FFTs (the operators are
 * separable, so no 2-D transform is materialised): pass 1 transforms columns (axis 0) and leaves
 * the x-part of the residual in r_u, r_v, r_div; pass 2 transforms rows (axis 1) and completes
 * them in place.
 * AXIS LENGTHS.  The FFT engine serves powers of two in [64, 1024] (every BASELINE.json size).  Any other length n in 3 .. 2048 --
 * the reference drivers' own 51 x 51 and 50 x 50 grids (src/chorin_fd/simulate.py:280-281, src/direct_fd/simulate.py:153-154), 96, 2048 --
 * takes, per axis, the same operator as a CIRCULANT matrix applied in float64 (csrc/spectral_dense.hip: O(n) multiply-adds per point,
 * exact to float32 output rounding, `precise` ignored; the first call with a new (n, L) builds the n-vector on the host and copies it,
 * synchronously).  The two axes of one call choose independently (e.g. 96 x 256).  Longer axes fail with NNS_ERR_UNSUPPORTED.
 * `precise` (every nns_spec_residual_* / nns_residual_both_* entry point) selects the arithmetic of the forward transforms:
 *   0  all-float32: the lines are forward-DIFFERENCED in physical space (exact in float32) and the spectral multiply becomes
 *      a bounded filter, so the white rounding noise of a float32 transform is not amplified by k.  First derivatives come
 *      out at float32 accuracy for any input; the viscous term keeps a relative amplification of rms nu pi N / (sqrt(3) L)
 *      per axis (1.9 at 1024^2, nu = 2 pi / 1000: 1.1e-6 rel-L2 against the float64 oracle; <= 4e-6 measured for nu <= 1);
 *   1  the library picks: all-float32 while that factor is <= 8, otherwise as 2.  Entry points that evaluate BOTH directions
 *      (nns_spec_residual_f32, nns_residual_both_f32, nns_spec_residual_bwd_f32) decide once from the larger of the two axes' factors, so
 *      an anisotropic grid never mixes arithmetic; the single-direction entry points (xpass / ypass / rowpass) decide from their own
 *      axis -- callers that combine them (nns/slab.py) resolve the policy themselves and pass 0 or 2.
 *      (API note: until round 1 `precise = 1` meant float64 forward transforms unconditionally; that is `precise = 2` now.)
 *   2  forward transforms and spectral multiply in float64, inverse in float32 (2-4e-7 rel-L2), whatever the viscosity.
 * nns_spec_residual_bwd_f32 follows the same rule (its three packed pairs per line are differenced the same way);
 * nns_spec_derivs_f32: 0 = plain all-float32, non-zero = float64 forward. */
int nns_spec_residual_f32(const float* u, const float* v, const float* p, const float* u_prev,
                          const float* v_prev, float* r_u, float* r_v, float* r_div,
                          int batch, int nx, int ny, double dt, double Lx, double Ly,
                          double rho, double nu, int precise, void* stream);
/* "Stencil + spectral residual on the same inputs" (the unit of BASELINE.json's metric) in two launches instead of three:
 * the spectral column pass, then ONE row pass that completes the spectral residual AND evaluates the FD 5-point residual
 * (the formula of nns_fd_residual_f32; second differences as differences of exact float32 first differences): a row pass holds
 * whole rows of u, v, p in registers, so the stencil's j-1 / j+1 neighbours are lane rotates.  Rows i-1 / i+1: in the all-float32
 * mode every line of a workgroup marches down a chunk of consecutive rows (row above parked in LDS, row below = the next row's
 * prefetch: no second read); in the float64-forward mode they are re-read from L2 / memory.
 * nx, ny powers of two in [64, 1024] run this fused form (ny = 1024: one row per wave, whole-wave DPP rotates; shorter rows share a
 * wave).  Results equal those of the two separate calls to rounding.  With ny outside the FFT engine's sizes the call IS the two
 * separate calls (nns_fd_residual_f32 + nns_spec_residual_f32, see its AXIS LENGTHS note); nx alone outside them only swaps the column pass.
 * The six output fields must not overlap the inputs or each other (rows i-1 / i+1 of the inputs are read while other rows'
 * outputs are being written).
 * Measured 11-24 % faster than the two calls at every size (tools/both_sizes_run.py). */
int nns_residual_both_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                          float* fd_r_u, float* fd_r_v, float* fd_r_div, float* sp_r_u, float* sp_r_v, float* sp_r_div,
                          int batch, int nx, int ny, double dt, double Lx, double Ly, double rho, double nu,
                          int precise, void* stream);
/* Its second launch alone: sp_r_* must hold the partials of nns_spec_residual_xpass_f32 on entry (the split form, for
 * timing the two launches separately and for callers that place an exchange between them). */
int nns_residual_both_rowpass_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                  float* fd_r_u, float* fd_r_v, float* fd_r_div, float* sp_r_u, float* sp_r_v, float* sp_r_div,
                                  int batch, int nx, int ny, double dt, double Lx, double Ly, double rho, double nu,
                                  int precise, void* stream);
/* The row pass on a ROW SLAB [batch][nx_local][ny] of grids sharded by rows over ranks (nns/slab.py):
 * the stencil's row above local row 0 / below local row nx_local-1 is read from halo_top / halo_bot, two
 * [3 (u, v, p)][batch][ny] messages holding the neighbour ranks' edge rows; dx is the grid spacing along x (the slab does not
 * know the global row count); nx_local >= 3, any value.  Same arithmetic per point as nns_residual_both_rowpass_f32.
 * halo_field_stride: elements between the u, v and p blocks of a halo message; 0 = batch * ny (a message made for exactly these grids).
 * A larger value lets a call on a CHUNK of the batch read its rows out of a message exchanged once for the whole batch
 * (halo_top + first_grid * ny, stride = whole_batch * ny): nns/slab.py pipelines batch chunks and sends one halo message per step. */
int nns_residual_both_rowpass_halo_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                       const float* halo_top, const float* halo_bot,
                                       float* fd_r_u, float* fd_r_v, float* fd_r_div, float* sp_r_u, float* sp_r_v, float* sp_r_div,
                                       int batch, int nx_local, int ny, long halo_field_stride, double dt, double dx, double Ly, double rho, double nu,
                                       int precise, void* stream);
/* The two halves of nns_spec_residual_f32, exposed for slab-decomposed (multi-GPU) use:
 * x-pass on a column slab [nx, ny_local] (needs complete columns), y-pass on a row slab
 * [nx_local, ny] (needs complete rows) which reads the x-parts from r_* and finishes them. */
int nns_spec_residual_xpass_f32(const float* u, const float* v, const float* p,
                                float* r_u, float* r_v, float* r_div, int batch, int nx, int ny,
                                double Lx, double rho, double nu, int precise, void* stream);
/* The column pass on a COLUMN SLAB whose rows arrive from an all-to-all in blocks of seg_rows rows per source rank:
 * u, v, p (and the partials written to r_*) are laid out [src][...][batch][seg_rows][ny] -- row r of grid b of a field
 * starts at field + (r / seg_rows) * seg_stride + b * seg_rows * ny + (r % seg_rows) * ny.  seg_rows: a power of two <= nx.
 * Reading the receive buffer and writing the send buffer of the return all-to-all in place removes the permuting copies on
 * both sides of the column pass.  Same arithmetic per column as nns_spec_residual_xpass_f32. */
int nns_spec_residual_xpass_seg_f32(const float* u, const float* v, const float* p, float* r_u, float* r_v, float* r_div,
                                    int batch, int nx, int ny, int seg_rows, long seg_stride, double Lx, double rho, double nu,
                                    int precise, void* stream);
int nns_spec_residual_ypass_f32(const float* u, const float* v, const float* p, const float* u_prev,
                                const float* v_prev, float* r_u, float* r_v, float* r_div,
                                int batch, int nx, int ny, double dt, double Ly,
                                double rho, double nu, int precise, void* stream);
/* The row passes on a ROW SLAB whose column-pass partials are read where the return all-to-all delivered them (round 4; no copy kernel
 * between the collective and the row pass): part_u / part_v / part_div point at source rank 0's block of each partial field in a buffer laid
 * out [src][field][batch][nx_local][seg_cols], seg_cols = ny / P; element (grid b, row i, column j) of a field sits at
 * part + (j / seg_cols) * seg_stride + (b * nx_local + i) * seg_cols + j % seg_cols -- a row is P contiguous pieces of seg_cols floats.
 * seg_cols: a power of two in [ny / 16, ny] (P <= 16: a register slot of a line never straddles two pieces); seg_stride >= batch * nx_local *
 * seg_cols (3 x that for the [src][3 fields] buffer of nns/slab.py).  r_* / sp_r_*: row slabs [batch][nx_local][ny], written only.
 * Same arithmetic per point as nns_spec_residual_ypass_f32 / nns_residual_both_rowpass_halo_f32 (bitwise the single-process result);
 * ny: a power of two in [64, 1024] (the segmented layout exists for the FFT engine only).
 * No reference counterpart: the reference runs on one device (src/neural_spectral/spectral_ode.py:155-156,165). */
int nns_spec_residual_ypass_seg_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                    const float* part_u, const float* part_v, const float* part_div, int seg_cols, long seg_stride,
                                    float* r_u, float* r_v, float* r_div, int batch, int nx, int ny, double dt, double Ly,
                                    double rho, double nu, int precise, void* stream);
int nns_residual_both_rowpass_halo_seg_f32(const float* u, const float* v, const float* p, const float* u_prev, const float* v_prev,
                                           const float* halo_top, const float* halo_bot,
                                           const float* part_u, const float* part_v, const float* part_div, int seg_cols, long seg_stride,
                                           float* fd_r_u, float* fd_r_v, float* fd_r_div, float* sp_r_u, float* sp_r_v, float* sp_r_div,
                                           int batch, int nx_local, int ny, long halo_field_stride, double dt, double dx, double Ly,
                                           double rho, double nu, int precise, void* stream);
/* The arithmetic a `precise` request resolves to for a WHOLE evaluation: 0 (all-float32 transforms on differenced lines) or 2 (float64
 * forward transforms), from both transformed axes and the NNS_SPEC_F64 override -- the one policy implementation, for hosts that call the
 * passes one by one (nns/slab.py). */
int nns_spec_resolve_precise(int precise, double nu, int nx, double Lx, int ny, double Ly);
/* Axis lengths outside the FFT engine's sizes run on circulant matrices (AXIS LENGTHS note above) whose table of (n, L) is built on the host,
 * allocated and copied SYNCHRONOUSLY at the first call that needs it, once per device -- not possible while a stream is being captured into a HIP
 * graph (such a call returns NNS_ERR_UNSUPPORTED).  This builds the table of one axis ahead of time on the current device. */
int nns_spec_dense_warmup(int n, double L);
/* Spectral derivatives of ONE real field (d/dx <-> i kx with the Nyquist mode dropped, lap <-> -|k|^2; definition
 * oracle/periodic.py: spectral_derivs): any of f_x, f_y, f_lap may be NULL.  precise as above. */
int nns_spec_derivs_f32(const float* f, float* f_x, float* f_y, float* f_lap, int batch, int nx, int ny,
                        double Lx, double Ly, int precise, void* stream);
/* 2-D real FFT pair with numpy.fft.rfft2 / irfft2 layout and normalisation (float32 transforms): spec is interleaved
 * complex64 [batch, nx, ny/2+1].  irfft2 uses spec as scratch for its column pass (spec is overwritten). */
int nns_spec_rfft2_f32(const float* f, float* spec, int batch, int nx, int ny, void* stream);
int nns_spec_irfft2_f32(float* spec, float* f, int batch, int nx, int ny, void* stream);

/* ---- neural_spectral field predictor: src/neural_spectral/spectral_ode.py, anode/ ------------ */
enum { NNS_ODE_EULER = 0, NNS_ODE_RK2 = 1, NNS_ODE_RK4 = 2 };   /* anode/scheme.py:21-42 */
/* ODEFunc (spectral_ode.py:14-34: Linear(K,hidden)-ReLU-Linear(hidden,hidden)-ELU-Linear(hidden,K), torch
 * nn.Linear layout W [out][in]) integrated by odesolver (anode/odesolver.py:21-37, time_stepper.py:35-45):
 * dt = 1/Nt, out[n] = y_{n+1} for n = 0..Nt-1, out is [Nt, mb, K].  One persistent kernel: weights resident
 * in LDS, linears on f32-input MFMA.  hidden must be 128, K <= 32. */
int nns_ode_mlp_fwd_f32(const float* z0, const float* W0, const float* b0, const float* W1, const float* b1,
                        const float* W2, const float* b2, float* out, int mb, int K, int hidden, int Nt,
                        int method, void* stream);
/* Backward of the above (what Checkpointing_Adjoint.backward computes, anode/adjoint.py:52-70, by
 * recomputation): given states = the forward's out and grad_out [Nt, mb, K], writes grad_z0 [mb, K] and the six
 * parameter gradients (zeroed and accumulated inside the call).  work: nns_ode_mlp_bwd_workspace(mb) bytes. */
size_t nns_ode_mlp_bwd_workspace(int mb);
int nns_ode_mlp_bwd_f32(const float* z0, const float* W0, const float* b0, const float* W1, const float* b1,
                        const float* W2, const float* b2, const float* states, const float* grad_out,
                        float* grad_z0, float* gW0, float* gb0, float* gW1, float* gb1, float* gW2, float* gb2,
                        void* work, int mb, int K, int hidden, int Nt, int method, void* stream);
/* The time-parallel form of that backward (anode/adjoint.py:38-70 walks the steps backwards one by one; every step's Jacobian
 * depends only on its own stored input state, so all of them can be formed at once):
 *   nns_ode_mlp_bwd_steps_f32  backward of `rows` INDEPENDENT single steps y -> y' of size dt: grad_y[r] = (dy'/dy)^T grad_out[r],
 *                              parameter gradients summed over the rows (work: nns_ode_mlp_bwd_workspace(rows) bytes); the six
 *                              parameter-gradient pointers may ALL be NULL (pass 1 below wants grad_y only: their products, the
 *                              zero-fills and the atomics are then skipped);
 *   nns_ode_adjoint_chain_f32  lam[Nt-1] = g[Nt-1], lam[s-1] = g[s-1] + lam[s] J[s]: the adjoint recurrence on the K x K step
 *                              Jacobians J [Nt][mb][K][K] (K <= 64), g, lam [Nt][mb][K].
 * Pass 1: rows (s, b, i) with grad_out = e_i give J; chain; pass 2: rows (s, b) with grad_out = lam[s][b] give the parameter
 * gradients and grad_z0 = grad_y of step 0 -- three launches of parallel work instead of Nt dependent steps. */
int nns_ode_mlp_bwd_steps_f32(const float* y, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                              const float* b2, const float* grad_out, float* grad_y, float* gW0, float* gb0, float* gW1, float* gb1,
                              float* gW2, float* gb2, void* work, int rows, int K, int hidden, double dt, int method, void* stream);
int nns_ode_adjoint_chain_f32(const float* J, const float* g, float* lam, int Nt, int mb, int K, void* stream);
/* Basis expansion (PDEFunc.forward, spectral_ode.py:71-79): pred[t][c][p] = sum_k coeff[t][k][c] basis[k][c][p];
 * coeff [T, K, C] (T = nt*mb), basis [K, C, P] (P = nx*ny), pred [T, C, P].  K <= 32. */
int nns_basis_expand_f32(const float* coeff, const float* basis, float* pred, int T, int K, int C, int P, void* stream);
/* Backward of nns_basis_expand_f32 for an arbitrary upstream gradient grad_pred [T, C, P]: gcoeff [T, K, C]
 * (zeroed inside, atomics), gbasis [K, C, P]. */
int nns_basis_expand_bwd_f32(const float* coeff, const float* basis, const float* grad_pred, float* gcoeff,
                             float* gbasis, int T, int K, int C, int P, void* stream);
/* Fused training loss (spectral_ode.py:182, torch.norm(pred - obs, 2)): accumulates sum (pred-obs)^2 into the
 * device double *sumsq (caller zeroes it; loss = sqrt) without materialising pred; obs is read once. */
int nns_basis_loss_fwd_f32(const float* coeff, const float* basis, const float* obs, double* sumsq,
                           int T, int K, int C, int P, void* stream);
/* Its gradients for g = scale * (pred - obs): gcoeff [T, K, C] (zeroed inside, atomics) and gbasis [K, C, P]. */
int nns_basis_loss_bwd_f32(const float* coeff, const float* basis, const float* obs, float scale,
                           float* gcoeff, float* gbasis, int T, int K, int C, int P, void* stream);
/* Loss and gradient in ONE sweep over the observations (training step, spectral_ode.py:178-190: forward, loss.backward()):
 * *sumsq += sum (pred - obs)^2 (zeroed by the caller) and gcoeff / gbasis = d(sumsq / 2) / d(coeff, basis), i.e. the gradient
 * WITHOUT the upstream / loss factor, which the caller applies afterwards (the gradient is linear in it).  obs, the only
 * large stream, is read once instead of twice. */
int nns_basis_loss_fused_f32(const float* coeff, const float* basis, const float* obs, double* sumsq, float* gcoeff, float* gbasis,
                             int T, int K, int C, int P, void* stream);

/* Per-pixel MLP forward (BasisFunc, spectral_ode.py:100-119: a stack of 1x1 Conv2d with ReLU between layers,
 * generalised to <= 8 layers of width <= 64): x [mb, widths[0], P] -> y [mb, widths[nlayers], P], P = nx*ny.
 * weights / biases: the layers' [C_out][C_in] matrices and [C_out] vectors packed back to back (device);
 * widths_host: nlayers+1 ints (host).  All layers are chained in MFMA accumulators (no LDS/HBM round trip
 * between layers).  bf16 == 0: float32 MFMA (exact fp32); bf16 != 0: bfloat16 operands, float32 accumulate. */
int nns_pixel_mlp_fwd_f32(const float* x, const float* weights, const float* biases, float* y, int mb, int P,
                          const int* widths_host, int nlayers, int bf16, void* stream);

/* Its backward (what autograd would do for the reference's Conv2d/ReLU stack, spectral_ode.py:100-119), fused in one
 * launch that recomputes the forward in registers: gy [mb, widths[nlayers], P] -> gx [mb, widths[0], P], gW / gB packed
 * like weights / biases (overwritten).  bf16 != 0: operands rounded to bfloat16 exactly as the forward does, float32
 * accumulation, any supported shape; bf16 == 0: float32 operands (v_mfma_f32_32x32x2_f32), widths <= 32 only (wider:
 * NNS_ERR_UNSUPPORTED).  Weight gradients contract over pixels through LDS images and are reduced over workgroups in a
 * fixed order.  workspace: nns_pixel_mlp_bwd_workspace bytes of device memory. */
int nns_pixel_mlp_bwd_workspace(const int* widths_host, int nlayers, size_t* bytes);
int nns_pixel_mlp_bwd_f32(const float* x, const float* gy, const float* weights, const float* biases,
                          float* gx, float* gW, float* gB, int mb, int P, const int* widths_host, int nlayers, int bf16,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- chorin_spectral (Chebyshev collocation): src/chorin_spectral/simulate.py ---------------- */
/* Row-major float64 GEMM on the matrix cores: C = alpha * op(A) op(B) + beta * C, op = transpose when the flag is
 * set; `batch` independent problems stored back to back.  Replaces the `@` products of _predictor_step
 * (:264-298) and _correction_step (:361-380). */
int nns_cheb_gemm_f64(const double* A, int lda, int transA, const double* B, int ldb, int transB, double* C, int ldc,
                      int M, int N, int K, double alpha, double beta, int batch, void* stream);
/* F = 2 f - 3 dt (un fx + vn fy) + dt (un1 f1x + vn1 f1y) + dt (fxx + fyy), elementwise on n values (:277-282). */
int nns_cheb_helmholtz_rhs_f64(const double* f, const double* un, const double* vn, const double* un1, const double* vn1,
                               const double* fx, const double* fy, const double* f1x, const double* f1y,
                               const double* fxx, const double* fyy, double* F, int n, double dt, void* stream);
/* out[i][j] = Hm[i][j] / (c0 + cx * lam_x[i] + cy * lam_y[j])  (:287-288, :372-373). */
int nns_cheb_diag_div_f64(const double* Hm, const double* lam_x, const double* lam_y, double* out, int ni, int nj,
                          double c0, double cx, double cy, void* stream);
/* full [Nx, Ny] = interior sol [(Nx-2), (Ny-2)] + boundary rows x0, xN [Ny-2] and columns y0, yN [Nx-2], corners 0
 * (:322-334). */
int nns_cheb_embed_f64(const double* sol, const double* x0, const double* xN, const double* y0, const double* yN,
                       double* full, int Nx, int Ny, void* stream);

/* ---- slab decomposition of one grid over the GPUs of a node: device-side message packing (nns/slab.py) --------------
 * No reference counterpart (src/neural_spectral/spectral_ode.py:155-156 runs on one device); SURVEY.md section 8 (e).
 * fields_host: a HOST array of nfields (<= 4) device pointers.  One launch each.
 * gather:  msg[f][o][e] = fields[f][o * outer_stride + line_off + e * elem_stride],  o < nouter, e < len;  scatter: the reverse.
 *   (a row of [B][nloc][ny] slabs: nouter = B, outer_stride = nloc * ny, line_off = row * ny, len = ny, elem_stride = 1;
 *    a column of an [nx][nyl] slab: nouter = 1, line_off = column, len = nx, elem_stride = nyl) */
int nns_slab_gather_lines_f32(const float* const* fields_host, int nfields, float* msg, long nouter, long outer_stride, long line_off,
                              long len, long elem_stride, void* stream);
int nns_slab_gather_lines_f64(const double* const* fields_host, int nfields, double* msg, long nouter, long outer_stride, long line_off,
                              long len, long elem_stride, void* stream);
int nns_slab_scatter_lines_f32(const float* msg, float* const* fields_host, int nfields, long nouter, long outer_stride, long line_off,
                               long len, long elem_stride, void* stream);
int nns_slab_scatter_lines_f64(const double* msg, double* const* fields_host, int nfields, long nouter, long outer_stride, long line_off,
                               long len, long elem_stride, void* stream);
/* transpose_pack: row slabs fields[f][b][i][j] -> send[d][f][b][i][jj], d = j / (ny / nranks): the all-to-all send buffer with every
 * destination's column block contiguous; transpose_unpack: recv[s][f][b][i][jj] -> fields[f][b][i][s * (ny / nranks) + jj]. */
int nns_slab_transpose_pack_f32(const float* const* fields_host, int nfields, float* send, int batch, int nloc, int ny, int nranks, void* stream);
int nns_slab_transpose_pack_f64(const double* const* fields_host, int nfields, double* send, int batch, int nloc, int ny, int nranks, void* stream);
int nns_slab_transpose_unpack_f32(const float* recv, float* const* fields_host, int nfields, int batch, int nloc, int ny, int nranks, void* stream);
int nns_slab_transpose_unpack_f64(const double* recv, double* const* fields_host, int nfields, int batch, int nloc, int ny, int nranks, void* stream);
/* transpose_pack of the batch chunk [grid0, grid0 + batch) of row slabs fields[f] [batch_total][nloc][ny] into send [d][f][batch][nloc][ny / nranks]
 * AND, in the same launch, the halo messages of the WHOLE local batch: first[f][b][j] = fields[f][b][0][j], last[f][b][j] = fields[f][b][nloc-1][j],
 * b < batch_total (first = last = NULL: the pack alone).  One launch where the slab step had three (round 4).  Needs 16-byte aligned pointers and
 * ny / nranks a multiple of the 16-byte vector (NNS_ERR_UNSUPPORTED otherwise: use the separate calls). */
int nns_slab_pack_halo_f32(const float* const* fields_host, int nfields, float* send, float* first, float* last, int batch_total, int grid0, int batch,
                           int nloc, int ny, int nranks, void* stream);
int nns_slab_pack_halo_f64(const double* const* fields_host, int nfields, double* send, double* first, double* last, int batch_total, int grid0, int batch,
                           int nloc, int ny, int nranks, void* stream);

/* ---- loss head of the physics-informed training step (the hypothesis of
 * src/neural_spectral/derivations/derivation.tex:25-34, which the reference states and never implements) ----------------------------------
 *   pred = state + mlp_out (channels u, v, p);  data = mean (pred - target)^2;  phys = mean r_u^2 + mean r_v^2 + w_div mean r_div^2 of the
 *   residual of pred;  total = data + lam phys.   Three HBM-bound passes replace the tensor ops between the MLP and the residual kernels
 *   (csrc/pinn_kernels.hip); the sums are deterministic (fixed grid, partials in double added in index order).
 * workspace: nns_pinn_workspace_bytes() bytes of device memory, 8-byte aligned, ZEROED ONCE by the caller (the kernels leave it reusable),
 * one per stream in flight. */
long nns_pinn_workspace_bytes(void);
/* assemble: out, state, target [batch][3][npix] (target NULL: no data term) -> u, v, p [batch][npix] = the channels of state + out;
 * u_prev, v_prev [batch][npix] = channels 0, 1 of state (both NULL when the caller already has them contiguous: batch = 1); the data term's
 * sum of squares stays in the workspace for nns_pinn_loss_f32. */
int nns_pinn_assemble_f32(const float* out, const float* state, const float* target, float* u, float* v, float* p, float* u_prev, float* v_prev,
                          void* workspace, int batch, long npix, void* stream);
/* loss: the residual fields r_u, r_v, r_div (n values each) of the assembled prediction -> out3 (device) = (total, data, phys); n_data = the
 * number of values under the data mean (3 batch npix; 0: no target).  When w_div != 1, r_div is SCALED IN PLACE by w_div so that the residual
 * adjoint applied to (r_u, r_v, r_div) as they stand gives (n / 2) d phys / d (u, v, p). */
int nns_pinn_loss_f32(const float* r_u, const float* r_v, float* r_div, long n, void* workspace, double n_data, double lam, double w_div, float* out3,
                      void* stream);
/* combine: grad_out [batch][3][npix] = up_data[0] c_data (pred - target) + up_phys[0] c_phys (g_u, g_v, g_p), pred = (u, v, p) [batch][npix],
 * (g_u, g_v, g_p) = the residual adjoint's output; up_* are DEVICE scalars (autograd's upstream gradient: no host read-back), c_data = 2 / n_data,
 * c_phys = 2 lam / n.  target NULL: the physics part alone (u, v, p, up_data unused). */
int nns_pinn_combine_f32(const float* g_u, const float* g_v, const float* g_p, const float* u, const float* v, const float* p, const float* target,
                         const float* up_data, const float* up_phys, double c_data, double c_phys, float* grad_out, int batch, long npix, void* stream);

/* ---- optimiser step (src/neural_spectral/spectral_ode.py:171,189; spectral_ode2.py:159,171; rnn.py:90,102; spectral_rnn.py:131,149:
 * torch.optim.Adam(model.parameters(), lr=1e-3) ... optimizer.step()) as ONE launch over all parameter tensors -------------------------------
 * Tables of ntensors device pointers (host arrays) and element counts; float32, contiguous; exp_avg / exp_avg_sq are the optimiser state, updated
 * in place like the parameters.  Arithmetic in torch's order (no amsgrad): g = grad (+ weight_decay p; maximize: -grad), m += (1 - beta1)(g - m),
 * v = beta2 v + (1 - beta2) g g, p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps); `step` counts from 1. */
int nns_adam_step_f32(float* const* params_host, const float* const* grads_host, float* const* exp_avg_host, float* const* exp_avg_sq_host,
                      const long* sizes_host, int ntensors, double lr, double beta1, double beta2, double eps, double weight_decay, long step,
                      int maximize, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NNS_H */
