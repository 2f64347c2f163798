// ocn_internal.h -- C++ launchers shared between the translation units of libocn_hip.
#pragma once
#include <vector>

#include "ocn_common.h"

namespace ocn {

constexpr int MAX_TUPLE = 8;

struct FieldTuple {
    double *f[MAX_TUPLE];
    int loc[MAX_TUPLE];
    int n;
};
struct StepTuple {
    double *U[MAX_TUPLE];
    const double *Gn[MAX_TUPLE];
    double *Gm[MAX_TUPLE];  // written by the cache mode, read otherwise
    int loc[MAX_TUPLE];
    int n;
};

// bottom / top boundary conditions of the fields of a FieldTuple (kind 0: default fill)
struct ZBcTuple {
    ZBc bottom[MAX_TUPLE], top[MAX_TUPLE];
};
int launch_fill_halos(const ocn_grid *grid, const FieldTuple &ft, int open_fill, int only_dir, hipStream_t stream,
                      const ZBcTuple *zbc = nullptr);
struct SideBcTuple {
    ZBc side[6][MAX_TUPLE];  // west, east, south, north, bottom, top of every field of a tuple
};
int launch_fill_halos_general(const ocn_grid *grid, const FieldTuple &ft, int open_fill, hipStream_t stream, const SideBcTuple *bcs = nullptr);
int launch_apply_flux_bcs(const ocn_grid *grid, const FieldTuple &G, const FieldTuple &fields, const ZBcTuple &zbc, hipStream_t stream);
int launch_apply_flux_bcs_lateral(const ocn_grid *grid, const FieldTuple &G, const FieldTuple &fields, const SideBcTuple &bcs, hipStream_t stream);
int launch_advection_timescale(const ocn_grid *grid, const double *u, const double *v, const double *w, double *out, hipStream_t stream);
int launch_hasnan(const double *a, long long n, int *flag, hipStream_t stream);
int launch_profile_marker(hipStream_t stream);
int wait_stream(hipStream_t stream, double seconds, const char *who);
int launch_hydrostatic_pressure(const ocn_grid *grid, const TermsDev &t, double *pHY, hipStream_t stream, const int32_t *irange = nullptr);
int launch_stepper(const ocn_grid *grid, const StepTuple &st, int mode, double dt, double c1, double c2, hipStream_t stream);
int launch_source_term(const ocn_grid *grid, const double *u, const double *v, const double *w, double dt, int out_mode,
                       double *out, long long ld1, long long ld2, hipStream_t stream, int perm_dim = -1);
int launch_set_source(int Nx, int Ny, int Nz, const double *R, const double *dzc, int Hz, double *out, int complex_out,
                      long long ld1, long long ld2, hipStream_t stream);
int launch_spectral_solve(int nxh, int Ny, int Nz, const double *lx, const double *ly, const double *lz, double *b,
                          int zero_mode_here, int joff, int koff, hipStream_t stream, double m = 0.0, int shifted = 0);
int launch_implicit_free_surface_rhs(const ocn_grid *grid, const double *u, const double *v, const double *eta, double grav, double dt,
                                     double *Qu, double *Qv, double *rhs, hipStream_t stream);
int launch_barotropic_pressure_correction(const ocn_grid *grid, double *u, double *v, const double *eta, double grav, double dt,
                                          hipStream_t stream);
int launch_copy_real(const ocn_grid *grid, const double *phi, double *p, hipStream_t stream, int real_source, int perm_dim = -1);
int launch_pressure_correct(const ocn_grid *grid, double *u, double *v, double *w, const double *p, double dt, hipStream_t stream);
int launch_main_diagonal(const ocn_grid *grid, int nxh, const double *lx, const double *ly, double *D, hipStream_t stream);
int launch_tridiag_z(int Nx, int Ny, int Nz, const double *a, const double *b, const double *c, const double *f, double *t,
                     double *phi, hipStream_t stream, int keep_storage = 0);
int launch_tridiag_z_real(int Nx, int Ny, int Nz, const double *a, const double *b, const double *c, const double *f, double *t, double *phi,
                          hipStream_t stream);
int launch_remove_mean_mode_real(long long s3, int Nz, double *phi, hipStream_t stream);
int launch_tridiag_z_strided(int ni, int nj, long long sj, long long sk, int Nz, const double *a, const double *b, const double *c,
                             const double *f, double *t, double *phi, hipStream_t stream, int keep_storage = 0);
int launch_main_diagonal_strided(const ocn_grid *grid, int ni, int nj, long long sj, long long sk, const double *lx, const double *ly,
                                 double *D, hipStream_t stream);
int launch_remove_mean_mode(long long s3, int Nz, double *phi, hipStream_t stream);
// column FFTs (colfft.hip): mode 0 forward (natural -> stage order), 1 inverse (stage order -> natural),
// 2 forward + spectral solve + inverse.  N in {64, 128, 256, 512}.
bool colfft_supported(int N);
int colfft_wavenumber(int N, int p);
std::vector<double> colfft_twiddles(int N);
int launch_colfft(int N, int mode, double *data, long long col_stride, long long batch_stride, int ncols, int nbatch,
                  const double *tw, const double *lx, const double *ly, const double *lc, double scale, int inner,
                  hipStream_t stream, int zero_mode = 1);
int launch_colfft_dct_solve(int N, double *data, long long col_stride, int ncols, const double *tw, const double *wd, const double *lx,
                            const double *ly, const double *lz, double scale, int inner, hipStream_t stream);
int launch_colfft_dct_to_field(int N, double *data, long long col_stride, long long batch_stride, int ncols, int nbatch, const double *tw,
                               const double *wd, double scale, int inner, int pdim, double *p, long long p0, long long ps2, long long ps3,
                               hipStream_t stream);
// slab pipeline of the distributed solver (colfft.hip): real y transform and z transform into / out of the all-to-all layout
bool realfft_y_supported(int Ny);
int launch_realfft_y(int Ny, int inverse, const double *rhs, double *spec, double *p, long long p_s2, long long p_s3, int nx, int Nz,
                     const double *twH, const double *twN, hipStream_t stream, const ocn_grid *grid = nullptr,
                     const double *u = nullptr, const double *v = nullptr, const double *w = nullptr, double dt = 1.0, int kc = 0,
                     long long chunk = 0, int scale_dz = 0, double scale = 1.0);
// transpose-free x direction of the distributed pressure solve (xtri.hip)
bool xtri_supported(int R, int Nxg);
int launch_xtri_sweep(double *a1, const double *ly, const double *lz, int NyH, int nx, int Nz, double dx, double scale, double *gsend,
                      hipStream_t stream);
int launch_xtri_finish(double *a1, const double *ly, const double *lz, int NyH, int nx, int Nz, double dx, const double *grecv, int rank,
                       int R, hipStream_t stream);
int launch_vector_invariant(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu, double *Gv,
                            hipStream_t stream, const double *eta = nullptr, double grav = 0.0);
int launch_split_explicit_forcing(const ocn_grid *grid, const double *Gun, const double *Gum, const double *Gvn, const double *Gvm, double chi,
                                  double *GU, double *GV, hipStream_t stream);
int launch_split_explicit_substeps(const ocn_grid *grid, int n, const double *weights, double dtau, double grav, double H, double *eta,
                                   double *U, double *V, double *etab, double *Ub, double *Vb, const double *GU, const double *GV,
                                   hipStream_t stream);
int launch_split_explicit_substeps_blocked(const ocn_grid *grid, int n, const double *weights, double dtau, double grav, double H, double *eta,
                                           double *U, double *V, double *etab, double *Ub, double *Vb, const double *GU, const double *GV,
                                           double *work, hipStream_t stream);
int launch_split_explicit_substeps_ab3(const ocn_grid *grid, int n, const double *weights, double dtau, double grav, double H, const double *coef,
                                       double *eta, double *U, double *V, double *etab, double *Ub, double *Vb, const double *GU, const double *GV,
                                       double *work, hipStream_t stream);
int launch_split_explicit_dist_begin(const ocn_grid *grid, int W, const double *eta, const double *U, const double *V, const double *GU,
                                     const double *GV, double *work, double *send_west, double *send_east, hipStream_t stream);
int launch_split_explicit_dist_run(const ocn_grid *grid, int W, int n, const double *weights, double dtau, double grav, double H, double *eta,
                                   double *U, double *V, double *work, const double *recv_west, const double *recv_east, hipStream_t stream);
int launch_barotropic_correct_w(const ocn_grid *grid, const double *us, const double *vs, double *u, double *v, double *w, const double *U,
                                const double *V, const double *Us, const double *Vs, double H, hipStream_t stream);
int launch_barotropic_mode(const ocn_grid *grid, const double *u, const double *v, double *U, double *V, hipStream_t stream);
int launch_barotropic_corrector(const ocn_grid *grid, double *u, double *v, const double *U, const double *V, double *Ub, double *Vb, double H,
                                hipStream_t stream);
int launch_plane_halo(const ocn_grid *grid, double *plane, hipStream_t stream);
int launch_w_from_continuity(const ocn_grid *grid, const double *u, const double *v, double *w, hipStream_t stream);
int launch_barotropic_gradient(const ocn_grid *grid, double grav, const double *eta, double *Gu, double *Gv, hipStream_t stream);
int launch_free_surface_ab2(const ocn_grid *grid, const double *w, double *eta, double *Gn, const double *Gm, double dt, double chi,
                            hipStream_t stream);
int launch_halo_plane_x(const ocn_grid *grid, double *field, int loc, int which, double *buf, int unpack, hipStream_t stream);
int launch_halo_pack_x_fields(const ocn_grid *grid, const FieldTuple &ft, double *west, double *east, int unpack, hipStream_t stream);
int launch_rowdct(int Nx, int Ny, int Nz, int mode, double *data, const double *tw, const double *wd, const double *lx, const double *ly,
                  const double *lz, double shift, int shifted, hipStream_t stream);
int launch_colfft_slab_z(int Nz, int inverse, const double *in, double *out, int nx, int NyH, int R, const double *tw, hipStream_t stream);
// row FFTs (rowfft.hip): inverse = 0: [div(u,v,w)/dt | real_in] -> half spectrum;  1: half spectrum -> rows of haloed p
bool rowfft_supported(int Nx);
void rowfft_twiddles(int Nx, std::vector<double> &twM, std::vector<double> &twN);
int launch_rowfft(const ocn_grid *grid, int inverse, const double *u, const double *v, const double *w, const double *real_in,
                  double dt, double *spec, double *p, const double *twM, const double *twN, double scale, hipStream_t stream,
                  int scale_dz = 0);
int launch_halo_pack_x(const ocn_grid *grid, const double *field, int loc, double *west, double *east, int unpack, hipStream_t stream);
int launch_transpose(int mode, int nx, int Ny, int Nz, int R, const double *src, double *dst, hipStream_t stream);

}  // namespace ocn

namespace ocn_strict {
int launch_momentum_tendencies_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, double *Gu,
                                       double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin = nullptr);
int launch_tracer_tendency_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, const double *c,
                                   double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse = nullptr);
int launch_momentum_extra_general(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w, double *Gu,
                                  double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin = nullptr);
int launch_tracer_diffusion_general(const ocn_grid *grid, double kappa, const double *kappa_e, const double *c, double *Gc,
                                    const int32_t *range, hipStream_t stream);
int launch_pressure_planes(const ocn_grid *grid, double *p, double *u, double dt, double *west, double *east, int unpack, hipStream_t stream);
int launch_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                               double *Gv, double *Gw, const int32_t *range, const ocn::FuseArgs *fuse, hipStream_t stream);
int launch_tracer_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                           double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse = nullptr);
int launch_tracer_pair_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *const c[2],
                                double *const Gc[2], const int32_t *range, hipStream_t stream, const ocn::TracerFuse fuse[2], int *launched);
int launch_momentum_centered2(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu, double *Gv,
                              double *Gw, const int32_t *range, hipStream_t stream);
int launch_tracer_centered2(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                            double *Gc, const int32_t *range, hipStream_t stream);
int launch_momentum_extra(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w,
                          double *Gu, double *Gv, double *Gw, const int32_t *range, hipStream_t stream,
                          const ocn::MomentumFinal *fin = nullptr);
int launch_hydrostatic_momentum(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w, double *Gu,
                                double *Gv, const ocn::MomentumFinal &mf, const ocn::HydroFuse &hf, hipStream_t stream);
int launch_tracer_diffusion(const ocn_grid *grid, double kappa, const double *kappa_e, const double *c, double *Gc,
                            const int32_t *range, hipStream_t stream);
int launch_amd_fused(const ocn_grid *grid, double Cnu, const double *u, const double *v, const double *w, double *nu_e, int ntr,
                     const double *Ck, const double *const *c, double *const *kappa_e, hipStream_t stream, const int32_t *irange = nullptr);
int launch_amd_viscosity(const ocn_grid *grid, double Cnu, const double *u, const double *v, const double *w, double *nu_e,
                         hipStream_t stream);
int launch_amd_diffusivity(const ocn_grid *grid, double Ck, const double *u, const double *v, const double *w, const double *c,
                           double *kappa_e, hipStream_t stream);
}
namespace ocn_fast {
int launch_momentum_tendencies_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, double *Gu,
                                       double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin = nullptr);
int launch_tracer_tendency_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, const double *c,
                                   double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse = nullptr);
int launch_momentum_extra_general(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w, double *Gu,
                                  double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin = nullptr);
int launch_tracer_diffusion_general(const ocn_grid *grid, double kappa, const double *kappa_e, const double *c, double *Gc,
                                    const int32_t *range, hipStream_t stream);
int launch_pressure_planes(const ocn_grid *grid, double *p, double *u, double dt, double *west, double *east, int unpack, hipStream_t stream);
int launch_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                               double *Gv, double *Gw, const int32_t *range, const ocn::FuseArgs *fuse, hipStream_t stream);
int launch_tracer_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                           double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse = nullptr);
int launch_tracer_pair_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *const c[2],
                                double *const Gc[2], const int32_t *range, hipStream_t stream, const ocn::TracerFuse fuse[2], int *launched);
int launch_momentum_centered2(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu, double *Gv,
                              double *Gw, const int32_t *range, hipStream_t stream);
int launch_tracer_centered2(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                            double *Gc, const int32_t *range, hipStream_t stream);
int launch_momentum_extra(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w,
                          double *Gu, double *Gv, double *Gw, const int32_t *range, hipStream_t stream,
                          const ocn::MomentumFinal *fin = nullptr);
int launch_hydrostatic_momentum(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w, double *Gu,
                                double *Gv, const ocn::MomentumFinal &mf, const ocn::HydroFuse &hf, hipStream_t stream);
int launch_tracer_diffusion(const ocn_grid *grid, double kappa, const double *kappa_e, const double *c, double *Gc,
                            const int32_t *range, hipStream_t stream);
int launch_amd_fused(const ocn_grid *grid, double Cnu, const double *u, const double *v, const double *w, double *nu_e, int ntr,
                     const double *Ck, const double *const *c, double *const *kappa_e, hipStream_t stream, const int32_t *irange = nullptr);
int launch_amd_viscosity(const ocn_grid *grid, double Cnu, const double *u, const double *v, const double *w, double *nu_e,
                         hipStream_t stream);
int launch_amd_diffusivity(const ocn_grid *grid, double Ck, const double *u, const double *v, const double *w, const double *c,
                           double *kappa_e, hipStream_t stream);
}

// advection = UpwindBiased(order=5): tendencies.hip compiled with OCN_UPWIND=1
namespace ocn_strict_up {
int launch_momentum_tendencies_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, double *Gu,
                                       double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin = nullptr);
int launch_tracer_tendency_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, const double *c,
                                   double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse = nullptr);
int launch_momentum_extra_general(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w, double *Gu,
                                  double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin = nullptr);
int launch_tracer_diffusion_general(const ocn_grid *grid, double kappa, const double *kappa_e, const double *c, double *Gc,
                                    const int32_t *range, hipStream_t stream);
int launch_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                               double *Gv, double *Gw, const int32_t *range, const ocn::FuseArgs *fuse, hipStream_t stream);
int launch_tracer_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                           double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse = nullptr);
int launch_tracer_pair_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *const c[2],
                                double *const Gc[2], const int32_t *range, hipStream_t stream, const ocn::TracerFuse fuse[2], int *launched);
}
namespace ocn_fast_up {
int launch_momentum_tendencies_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, double *Gu,
                                       double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin = nullptr);
int launch_tracer_tendency_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, const double *c,
                                   double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse = nullptr);
int launch_momentum_extra_general(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w, double *Gu,
                                  double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin = nullptr);
int launch_tracer_diffusion_general(const ocn_grid *grid, double kappa, const double *kappa_e, const double *c, double *Gc,
                                    const int32_t *range, hipStream_t stream);
int launch_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                               double *Gv, double *Gw, const int32_t *range, const ocn::FuseArgs *fuse, hipStream_t stream);
int launch_tracer_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                           double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse = nullptr);
int launch_tracer_pair_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *const c[2],
                                double *const Gc[2], const int32_t *range, hipStream_t stream, const ocn::TracerFuse fuse[2], int *launched);
}
