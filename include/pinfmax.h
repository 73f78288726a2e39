/*
 * pinfmax.h -- C ABI of libpinfmax_hip.so, the MI355X (gfx950) implementation
 * of PINOCCHIO's collapse-time hot path (Fmax sweep + 2LPT/3LPT displacements).
 *
 * Plain C, plain pointers and sizes.  One context = one MPI rank = one GPU =
 * one x-slab of the grid, exactly the reference's 1-D PFFT decomposition
 * (src/initialization.c:1317-1325).  Every entry point names the reference
 * interface it replaces (file:line relative to the reference tree).  The
 * reference-side adapter that maps PINOCCHIO's globals onto these calls is
 * pinocchio_amd/host/pf_compat.c; INTEGRATION.md shows the link line.
 *
 * Conventions (src/pinocchio.c:259-263, src/fmax.c): every function returns
 * int, 0 = ok, non-zero = error after printing "ERROR on task %d: ..." on
 * stdout; the caller aborts (MPI_Abort).  All ranks call every function
 * collectively and in the same order (PFFT plans and reductions are collective
 * in the reference too).  Not thread-safe per context (MPI_THREAD_FUNNELED).
 * There is NO CPU fallback: without a HIP device pf_create fails.
 */
#ifndef PINFMAX_H
#define PINFMAX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_NBINS 210     /* NBINS, src/pinocchio.h:65 (spline knots, Fmax PDF bins) */
#define PF_MAX_SMOOTH 64 /* upper bound on Smoothing.Nsmooth accepted by pf_sweep */

typedef struct pf_ctx pf_ctx;

/* grid_data / FFT decomposition (src/pinocchio.h:295-308, src/fmax-pfft.c:80-134) */
typedef struct {
  int64_t n;        /* GSglobal[_x_] = [_y_] = [_z_] : cubic grid (the reference accepts any GridSize: src/fmax-pfft.c:139-188).
                       Power of two in 16..2048: the hand-written transform passes (any nranks, fp64 or fp32 fields).
                       n = 8 m, m = 2^a 3^b 5^c, up to 2048 (24, 40, 200, 384, 768, 1000, 1536 ...): the hand-written passes on
                       stage plans of radices 8, 5, 4, 3, 2 -- built into the kernels for 200, 384, 400, 640, 768, 800, 1000, 1280,
                       1536, 1600 and 2000 (the fast way: 768^3 runs at 0.84 of the per-cell rate of 1024^3), a run-time table for the
                       others -- any nranks (a power of two), fp64 or fp32 fields.  Any other even size in 4..2048: hand-written
                       chirp-z (Bluestein) transforms on the power-of-two stages, one per component, one rank and fp64 fields only
                       (csrc/pf_gfft.hip; no library transform anywhere).  pf_transform_path() says which */
  int     rank;     /* ThisTask */
  int     nranks;   /* NTasks; x-slabs, nranks must divide n (src/fmax-pfft.c:95-111 without the ragged last slab).  A power of two for n = 2^k; any
                       divisor that leaves slabs of two planes and more for n = 8 m that is not a power of two (96^3 on 3, 120^3 on 6, 200^3 on 5 ...) */
  int     device;   /* HIP device ordinal of this rank */
  int     field_bytes; /* 8: fp64 density/derivative fields (reference); 4: fp32 fields, fp64 collapse solve */
  int     flags;    /* PF_FLAG_* */
} pf_config;

#define PF_FLAG_TIMING 1  /* record per-kernel HIP-event timings (pf_kernel_stats) */
/* Environment, read once per context in pf_create (DESIGN.md section 6 has the whole table).  One of them changes what is computed:
   PF_EXACT_LIBM=1 makes the collapse solve call the reference's own libm functions (cos x3, acos, pow, pow, log10, exp, IEEE / and sqrt)
   instead of the series / table / hardware-seeded forms of the default arithmetic.  It is a DIAGNOSTIC mode -- the flavour in which
   the oracle's solver fed with the device's Hessian reproduces the device's Fmax bit for bit (tests) -- not a supported speed: 1.6x the
   step time, kernels with scratch, never tuned or profiled. */
/* a -DDOUBLE_PRECISION_PRODUCTS build (src/Makefile:68, src/pinocchio.h:219-225: PRODFLOAT double): Fmax and the Vel*
   fields of product_data are doubles.  Fmax is then kept and compared in fp64 (no rounding of the running maximum to fp32
   between radii, cf. src/collapse_times.c:587-590) and the displacements leave the z-pass as the doubles it computes.
   fp64 fields only.  pf_get_products / pf_update_products write doubles at the layout's offsets; pf_select_sorted and
   pf_get_block (fp32 by definition) refuse. */
#define PF_FLAG_DOUBLE_PRODUCTS 2

/* layout of the caller's product_data record (src/pinocchio.h:233-259).
   Offsets in bytes; a negative offset means "field absent".  pf_layout_3lpt()
   fills the -DTWO_LPT -DTHREE_LPT float layout (56 B). */
typedef struct {
  size_t stride;     /* sizeof(product_data) */
  int    off_Rmax, off_Fmax, off_Vel, off_Vel_2LPT, off_Vel_3LPT_1, off_Vel_3LPT_2;
} pf_product_layout;
void pf_layout_3lpt(pf_product_layout *l);

/* cputime_data (src/pinocchio.h:368-378), seconds, device time by HIP events.  fmax, deriv, coll, lpt are spans between events
   on the streams the phases run on.  In a sweep whose collapse passes ran beside the z-passes of the following radii (the
   default, pf_solve_ran_beside_zpass) `coll` is what they add to the sweep beyond the derivative passes, so that
   deriv + coll <= fmax as in the reference's report (src/fmax.c:160-170); with every kernel in line (PF_SOLVE_BESIDE_Z=0)
   both are the plain spans. */
typedef struct {
  double fmax, deriv, fft, coll, lpt, mem_transf;
} pf_cputime;

/* Device memory a context of this configuration holds, in bytes: *at_create by pf_create itself, *peak once a sweep and the LPT part
   have run (the second field set is allocated at first use).  pf_create compares *peak with the free memory of the device
   (hipMemGetInfo) BEFORE allocating anything and fails in the usual format when it does not fit (PF_PREFLIGHT=0: no check).  The
   host's counterpart of the memory report of src/allocations.c:60-160.  Needs no device. */
int pf_plan_bytes(const pf_config *cfg, size_t *at_create, size_t *peak);

/* --- life cycle: replaces set_one_grid + compute_fft_plans + the FFT-buffer
       part of allocate_main_memory (src/fmax-pfft.c:80-188, src/allocations.c:382) --- */
int  pf_create(pf_ctx **out, const pf_config *cfg);
/* finalize_fft (src/fmax-pfft.c:231-252) + release of all device memory */
int  pf_destroy(pf_ctx *ctx);
const char *pf_last_error(void);

/* --- multi-GPU exchange (replaces the MPI_Alltoall inside pfft_execute,
       src/fmax-pfft.c:197,211).  One all-to-all per 3-D FFT: every rank sends
       bytes_per_peer bytes from sendbuf + q*bytes_per_peer to rank q and
       receives into recvbuf + p*bytes_per_peer from rank p, ordered on `stream`
       (a hipStream_t).  Install either a callback or the built-in RCCL
       exchange (grouped ncclSend/ncclRecv) created from a broadcast
       ncclUniqueId (128 bytes).  Not needed when nranks == 1. --- */
typedef int (*pf_alltoall_fn)(void *user, const void *sendbuf, void *recvbuf,
                              size_t bytes_per_peer, void *stream);
int pf_set_exchange(pf_ctx *ctx, pf_alltoall_fn fn, void *user);
/* Optional companion for band-limited spectra (smoothing radii whose Gaussian window is below 2^-60 beyond |k| = band):
   of each of the P blocks only one row range carries data.  Rank r sends send_bytes bytes from
   sendbuf + q*block_bytes + send_off to every rank q (nothing when send_bytes == 0) and receives recv_bytes[p] bytes into
   recvbuf + p*block_bytes + recv_off[p] from every rank p.  Without it the full blocks go through pf_alltoall_fn. */
typedef int (*pf_alltoallv_fn)(void *user, const void *sendbuf, void *recvbuf, size_t block_bytes, size_t send_off,
                               size_t send_bytes, const size_t *recv_off, const size_t *recv_bytes, void *stream);
int pf_set_exchange_rows(pf_ctx *ctx, pf_alltoallv_fn fn, void *user);
int pf_rccl_available(void);                        /* 1 when librccl can be bound in this process (host side only: no
                                                       communicator, no collective -- what the ranks vote on before any of them
                                                       enters ncclCommInitRank) */
#define PF_RCCL_ID_BYTES 128                         /* sizeof(ncclUniqueId); checked at build time against rccl.h */
int pf_rccl_unique_id(void *id128);                 /* rank 0: ncclGetUniqueId */
int pf_init_rccl(pf_ctx *ctx, const void *id128);   /* all ranks: ncclCommInitRank; the context owns the communicator */
/* version code of the RCCL bound at run time (0: none bound); *build_code: NCCL_VERSION_CODE of the header the library was built
   against.  Binding refuses a run-time library of another major version (types and enumerators come from that header). */
int pf_rccl_version(int *build_code);
int pf_release_rccl(pf_ctx *ctx);                   /* ncclCommDestroy + callbacks cleared (also done by pf_destroy) */
int pf_rccl_comm_count(pf_ctx *ctx);                /* ncclCommCount of the built-in exchange's communicator: the number of ranks
                                                       RCCL itself sees (the MPI_Comm_size of FFT_Comm, src/initialization.c:1317);
                                                       0 when the built-in exchange is not installed, -1 on error */
/* small reductions (MPI_Reduce/MPI_Bcast at src/collapse_times.c:656-667,
   src/fmax.c:527): sum `count` doubles / uint64 in place over all ranks */
typedef int (*pf_allreduce_fn)(void *user, void *buf, size_t count, int is_u64, void *stream);
int pf_set_allreduce(pf_ctx *ctx, pf_allreduce_fn fn, void *user);
/* in-process fabric: P contexts (ranks 0..P-1) driven by P host threads on ONE GPU meet in the
   exchange through device-to-device copies.  Bring-up/test transport for the slab code path on a
   single-GPU box; multi-GPU runs use pf_init_rccl. */
typedef struct pf_fabric pf_fabric;
pf_fabric *pf_fabric_create(int nranks);
void pf_fabric_destroy(pf_fabric *f);
int pf_fabric_attach(pf_fabric *f, pf_ctx *ctx);
/* test knob: every all-to-all first idles its stream for that long, so that the copies land late and anything the
   pipelined exchange fails to wait for reads stale data */
int pf_fabric_set_delay(pf_fabric *f, int microseconds);
/* self-test of the installed exchange (pattern through the all-to-all and the all-reduce), any nranks >= 1 */
int pf_debug_exchange(pf_ctx *ctx, size_t bytes_per_peer);
/* Measurement aid: ONE rank of an nranks-rank decomposition on its own.  The all-to-all hands this rank's own blocks back to it
   (copied for the first `copies` calls, so that the receive buffers hold finite, field-like numbers; afterwards nothing moves) and
   the all-reduce leaves the rank's contribution.  The kernels then run on the rank's slab of the full-size box with the launch
   geometry of the real run; the results are NOT those of the box.  Used by `bench.py --slab-of P` to time the compute side of a
   configuration whose box does not fit one GPU (BASELINE config 5: 2048^3 with fp32 fields on eight GPUs). */
int pf_set_loopback_exchange(pf_ctx *ctx, int copies);   /* refused when a real exchange (RCCL, callbacks, fabric) is installed */
/* ... with ONE exception, which is how a configuration that fits no single GPU is CHECKED rank by rank on one (BASELINE config 5,
   tests/test_gpu_config5.py): a context that keeps the whole spectrum (PF_REPLICATE_DK=1) and whose density came from
   pf_genic_density -- a function of (seed, cosmology) alone -- generates every rank's slab of delta(k) itself; the sweep of such a
   context exchanges nothing, so its Fmax / Rmax / variance contributions / histogram ARE those of this rank's slab of the box.
   (The LPT part still transposes the source spectra: its displacements behind a loopback are no box's.) */
int pf_loopback_active(pf_ctx *ctx);                      /* 1 when the loopback stands in for the exchange (results are no box's) */
/* device pointers + size (bytes) of the exchange buffers, so that a host
   harness can wrap them (e.g. torch tensors for torch.distributed) */
int pf_exchange_buffers(pf_ctx *ctx, void **sendbuf, void **recvbuf, size_t *bytes);
/* stream the work is enqueued on (hipStream_t); pf_set_stream adopts a caller stream.  Two internal streams run beside it, ordered
   against it by events and joined before an entry point returns: the communication stream of a pipelined multi-rank run and the
   solve stream of a sweep (the collapse solve of radius i beside the z-pass of radius i + 1: the default with fp32 fields, PF_SOLVE_BESIDE_Z=1
   with fp64 fields, where it gains nothing since round 5; PF_SOLVE_BESIDE_Z=0: every kernel in line).  What
   follows an entry point on this stream sees all of its results. */
int pf_set_stream(pf_ctx *ctx, void *stream);
void *pf_get_stream(pf_ctx *ctx);

/* --- inputs --- */
/* kdensity[0] (src/fmax-pfft.c:366): this rank's x-slab of the half-spectrum,
   fp64 [n/nranks][n][n/2+1][2], pre-multiplied by N^3 (src/GenIC.c:430-445).
   host pointer; uploaded (and converted to fp32 when field_bytes == 4). */
int pf_set_density(pf_ctx *ctx, const double *kdensity_slab);
/* params.use_transposed_fft (PFFT_TRANSPOSED_OUT on slabs, src/fmax-pfft.c:92, 271-281): with on != 0 every spectrum that
   crosses this interface -- pf_set_density, pf_get_density, pf_get_kvector, pf_forward_transform (out),
   pf_reverse_transform / pf_derivative (in) -- is this rank's KY-slab in the memory order [ky_local][kx][kz][2]
   (ky_local = n/nranks rows starting at rank * n/nranks) instead of the kx-slab [kx_local][ky][kz][2].  The device
   layout is a ky-slab anyway, so the regrouping all-to-all of the non-transposed boundary disappears.  Call it right
   after pf_create; results do not depend on it. */
int pf_set_transposed_spectra(pf_ctx *ctx, int on);
/* 1 if this context (nranks > 1) keeps the whole delta(k) on every rank, so that the second derivatives of the sweep and
   the Zel'dovich displacements need no all-to-all (each rank transforms every x-line and stores its own slab): chosen at
   pf_create -- up to four ranks, or PF_REPLICATE_DK=0|1.  Results do not depend on it. */
int pf_replicated_spectrum(pf_ctx *ctx);
/* synthetic delta(k) generated in HBM (bench / large property tests):
   Philox-4x32 white noise, P(k) ~ k^slope inside the Nyquist sphere, DC and
   Nyquist planes zero, sigma(R=0) = sigma0.  numpy mirror:
   pinocchio_amd/synth.py philox_density. */
int pf_synth_density(pf_ctx *ctx, uint64_t seed, double sigma0, double slope);
/* GenIC_large (src/GenIC.c:73-460) on the device: PINOCCHIO's own Gaussian initial conditions generated straight
   into HBM -- one gsl_rng_ranlxd1 stream per (kx,ky) column seeded from the MT19937 seed plane along the spiral
   (src/GenIC.c:840-990), Eisenstein & Hu P(k) (src/cosmo.c:1443-1497) times PkNorm, Hermitian kz = 0 plane, Nyquist
   planes and DC zero, final factor N^3.  Replaces GenIC + pf_set_density (no host array, no 8.6 GB upload at 1024^3);
   every rank generates its own k-space slab.  BoxSize in true Mpc (params.BoxSize_htrue), PkNorm as printed by
   normalize_PowerSpectrum (src/cosmo.c:1058-1075) or from pf_pk_norm (sigma8^2 / top-hat variance at 8/h Mpc). */
typedef struct {
  double Omega0, OmegaBaryon, Hubble100, PrimordialIndex;
  double BoxSize_true_Mpc;
  double PkNorm;
  unsigned int RandomSeed;
  int FixedIC;    /* params.FixedIC: "non-random modules of the Fourier modes", no Rayleigh factor -log(ampl) (src/GenIC.c:375) */
  int PairedIC;   /* params.PairedIC: every phase shifted by pi (src/GenIC.c:371) */
  /* pk_n > 0: a tabulated spectrum instead of Eisenstein & Hu (FileWithInputSpectrum <file> or CAMBTable: WhichSpectrum 2 / 5) --
     P(k) = PkNorm 10^my_spline_eval(SPLINE[SP_PK], log10 k) / k^3 (PowerSpec_Tabulated, src/cosmo.c:1432-1435) with the knots of
     SPLINE[SP_PK]: pk_logk[i] = log10 k [true 1/Mpc], pk_logk3p[i] = log10(k^3 P) (read_Pk_from_file :1099-1170,
     read_Pk_table_from_CAMB :1290-1330); host arrays, copied.  The cosmology fields are then unused. */
  int pk_n;
  const double *pk_logk, *pk_logk3p;
  /* the other forms of PowerSpectrum() (src/cosmo.c:953-1007).  spectrum: 0 = by pk_n as above (Eisenstein & Hu or the table);
     3 = the Efstathiou fit (FileWithInputSpectrum "Efstathiou": PowerSpec_Efstathiou :1437-1440, Gamma = SHAPE_EFST = 0.21);
     4 = a power law k^PrimordialIndex ("PowerLaw", :1442-1445).  WDM_PartMass_in_kev > 0: every form is multiplied by the
     warm-dark-matter cut-off Tf^2 of Bode, Ostriker & Turok (:987-1005; needs Omega0, OmegaBaryon, Hubble100);
     UnitLength_in_cm: the reference's global of that name (0 = its default 3.085678e24, i.e. Mpc) */
  int spectrum;
  double WDM_PartMass_in_kev, UnitLength_in_cm;
} pf_genic_params;
int pf_pk_norm(const pf_genic_params *p, double sigma8, double *pknorm);
int pf_genic_density(pf_ctx *ctx, const pf_genic_params *p);
/* SPLINE[SP_INVGROW] knots (src/cosmo.c:401): x = log10 D, y = log10 a, n knots
   (natural cubic spline coefficients are computed on the host, GSL cspline).
   ismooth = -1: one spline for every radius (non SCALE_DEPENDENT build);
   ismooth >= 0: SPLINE_INVGROW[ismooth] (src/initialization.c:1704-1708). */
int pf_set_invgrow(pf_ctx *ctx, int ismooth, const double *x, const double *y, int n);
/* growth multipliers applied by compute_derivative when ScaleDep.order = 1..4
   (src/fmax-pfft.c:344-364): g[0]=GrowingMode, g[1]=GrowingMode_2LPT,
   g[2]=GrowingMode_3LPT_1 (carrying its minus sign, src/cosmo.c:1810),
   g[3]=GrowingMode_3LPT_2, all at the target redshift, scale-independent. */
int pf_set_growth(pf_ctx *ctx, const double g[4]);
/* SCALE_DEPENDENT build: the multiplier of ScaleDep.order = order (1..4) depends on |k| --
   GrowingMode*(z, k_module) = sign * pow(10., InterpolateGrowth(z, k, SP_GROW*)) (src/cosmo.c:1728-1755, 1789-1819),
   log-linear between nk k-bins at 10^(logkmin + j dlogk) (NkBINS = 10, LOGKMIN = -3, DELTALOGK = 0.5,
   src/def_splines.h:40-42).  log10_growth[j] = my_spline_eval(SPLINE[SP_GROW* + j], -log10(1+z)) evaluated by the
   caller at the target redshift; sign = -1 for order 3 (src/cosmo.c:1810).  |k| is taken in rad/cell exactly as
   compute_derivative passes it (src/fmax-pfft.c:315-359).  nk = 0 returns to the scalar of pf_set_growth. */
int pf_set_growth_table(pf_ctx *ctx, int order, const double *log10_growth, int nk, double logkmin, double dlogk, double sign);

/* TABULATED_CT build (src/collapse_times.c:780-1231, the BILINEAR_SPLINE interpolation of :40): per smoothing radius
   a table of ell() on 100 x 50 x 50 nodes in (delta, x, y) = (l1+l2+l3, l1-l2, l2-l3) / sqrt(Smoothing.Variance[ismooth])
   and one natural cubic spline in delta per (x, y) node; the per-cell pass then interpolates instead of solving.
   pf_set_tabulated_ct(ns, Smoothing.Variance): pf_sweep builds the table of each radius right before its
   collapse-time pass (src/fmax.c:103-106); ns = 0 returns to the direct solve.
   pf_ct_build = initialize_collapse_times(ismooth, .) alone: the table is computed on the device (with the
   inverse-growth spline of that radius) and, if table_host is not NULL, copied out as CT_table
   [iy][ix][id] (100*50*50 doubles, what the reference writes to its CTtable file);
   pf_ct_load installs a table read from such a file (params.CTtableFile) instead of computing it.
   A following pf_collapse_times(ismooth) uses the table in place. */
int pf_set_tabulated_ct(pf_ctx *ctx, int nsmooth, const double *variance);
/* How the collapse pass reads the table (interpolate_collapse_time, src/collapse_times.c:1139-1231): 0 = BILINEAR_SPLINE, the
   source's own define (default); 1 = a build with -DTRILINEAR (eight table entries, no splines); 2 = -DALL_SPLINE (sixteen
   node splines, then gsl_spline2d's bicubic on the 4 x 4 grid around the cell) -- the three options of
   tests/Readme_Pinocchio_tests_V5_1.txt.  The table and its file are the same for all three. */
int pf_set_ct_interpolation(pf_ctx *ctx, int flavour);
/* What fills the table: model 0 = ELL_CLASSIC (ell_classic + InverseGrowingMode, src/collapse_times.c:114-221, 404-415);
   model 1 = ELL_SNG (src/collapse_times.c:222-400, 416-426): per node one adaptive RKF45 integration of the
   nine-equation system of Nadkarni-Ghosh & Singhal (2016) -- the step, error control and accept/reject logic of
   gsl_odeiv2_step_rkf45 / control_standard_new(1e-6, 1e-6, 1, 1) / evolve_apply -- with
   cosmo = {Omega0, OmegaLambda, OmegaRad, OmegaK} for OmegaMatter(z) / OmegaLambda(z) (src/cosmo.c:1675-1718) and
   D_in[ismooth] = GrowingMode(1/1e-5 - 1, k of the radius) (:353-361).  With pf_set_tabulated_ct the model fills the
   table (250 000 integrations per radius); without it -- a reference build with -DELL_SNG and no -DTABULATED_CT -- every
   cell integrates its own ellipsoid in the collapse pass (k_collapse_sng: correct, and ~10^3 times the work of the table
   at 1024^3).  pf_set_modified_gravity adds the MOD_GRAV_FR force modification. */
int pf_set_collapse_model(pf_ctx *ctx, int model, const double cosmo[4], int nsmooth, const double *D_in);
/* -DMOD_GRAV_FR on top of ELL_SNG: the force in the velocity equations is enhanced by 1 + ForceModification(size, a, delta)
   (src/collapse_times.c:271-273, 295-312), Hu-Sawicki f(R) with |f_R0| = fr0 (the FR0 of the build; 0 switches it off),
   h_over_c = 100 / c[km/s] (src/cosmo.c:109), size[ismooth] = the ODE parameter of that radius: Smoothing.Radius[ismooth],
   and Smoothing.Radius[ismooth-1] for the last one (:378-388). */
int pf_set_modified_gravity(pf_ctx *ctx, double fr0, double h_over_c, int nsmooth, const double *size);
int pf_ct_build(pf_ctx *ctx, int ismooth, double variance, double *table_host);
int pf_ct_load(pf_ctx *ctx, int ismooth, double variance, const double *table_host);

/* --- the path --- */
/* compute_fmax's radius loop (src/fmax.c:66-150): for each radius (CELL units,
   Rsmooth = Radius/CellSize, src/fmax.c:233) second derivatives + collapse
   times; Smoothing.TrueVariance[0..ns-1] out (src/collapse_times.c:670). */
int pf_sweep(pf_ctx *ctx, int ns, const double *radius_cells, double *true_variance);
/* The reference chooses the order of the displacements at compile time (src/Makefile: -DTWO_LPT, -DTHREE_LPT;
   src/fmax.c:300-336, src/LPT.c:30, 78, 113, 214).  order 3 (default): both; 2: -DTWO_LPT alone -- the 2LPT source and its
   displacement, no 3LPT sources, no Hessian of the 2LPT potential; 1: neither -- Zel'dovich displacements only, no second
   derivatives on re-entry.  Columns of orders that are not computed are zero (their fields do not exist in such a build's
   product_data: give them negative offsets in pf_product_layout). */
int pf_set_lpt_order(pf_ctx *ctx, int order);
/* compute_fmax goes straight from the last radius to compute_displacements(1, 0, z) (src/fmax.c:150-163): with on != 0 the
   collapse pass of the last radius of every following pf_sweep also writes the 2LPT / 3LPT sources of src/LPT.c:64-93
   from the six components it holds, and the next pf_displacements(1, 0) starts from them instead of reading the six
   fields again (same arithmetic, same sums: results do not change).  Off by default: a sweep that is not followed by
   the displacements would write three fields for nothing. */
int pf_set_sources_in_sweep(pf_ctx *ctx, int on);
/* compute_second_derivatives (src/fmax.c:225-258): six Hessian fields at one radius */
int pf_second_derivatives(pf_ctx *ctx, double radius_cells);
/* compute_collapse_times (src/collapse_times.c:431-673) on the resident Hessian */
int pf_collapse_times(pf_ctx *ctx, int ismooth, double *true_variance);
/* compute_displacements(compute_sources, recompute_sd, z) (src/fmax.c:292-367):
   2LPT/3LPT sources + 12 displacement fields; growth from pf_set_growth.
   With compute_sources = 0 the resident LPT spectra are reused
   (RECOMPUTE_DISPLACEMENTS re-entry, src/fragment.c:398-410). */
int pf_displacements(pf_ctx *ctx, int compute_sources, int recompute_sd);
/* Fmax_PDF (src/fmax.c:509-550): 210-bin histogram of (int)(Fmax*10), summed over ranks */
int pf_fmax_pdf(pf_ctx *ctx, unsigned long long hist[PF_NBINS]);

/* --- outputs --- */
/* products[] of this rank's slab into the caller's AoS (host), index
   i = z + n*(y + n*x_local) (src/pinocchio.h:84-85) */
int pf_get_products(pf_ctx *ctx, void *products_host, const pf_product_layout *layout);
/* the same columns merged into records the caller already holds: bytes of the record not named by a
   non-negative offset keep their host value (Fmax/Rmax, the *_prev copies of a RECOMPUTE_DISPLACEMENTS
   build filled by shift_all_displacements, src/fragment.c:832-850).  Used after a re-entrant
   compute_displacements(0, 0, z) (src/fragment.c:398-410), which rewrites only the Vel* fields. */
int pf_update_products(pf_ctx *ctx, void *products_host, const pf_product_layout *layout);
/* First stage of fragmentation on the device: the cells of this rank's slab with Fmax >= flast (update_distmap,
   src/distribute.c:695) in order of descending Fmax (sort_and_organize, src/fragment.c:484-503; index_compare_F
   :118-126; equal keys, which qsort leaves unspecified, by ascending index).  cell_index = z + n*(y + n*x_local)
   (src/pinocchio.h:84-85).  *count receives the number selected; the first min(*count, capacity) entries are
   copied (either array may be NULL). */
int pf_select_sorted(pf_ctx *ctx, float flast, size_t capacity, unsigned int *cell_index, float *fmax, size_t *count);
/* Per-particle payload of one block of the "timeless snapshot" (write_timeless_snapshot, src/write_snapshot.c:207-342)
   for this rank's slab, from the SoA columns in HBM: name = "ID  " (1 + global index as MYIDTYPE of id_bytes = 4 or 8,
   :648-664), "FMAX" float, "RMAX" int, "ZEL " / "2LPT" / "31PT" / "32PT" float[3] per particle (:700-855). */
int pf_get_block(pf_ctx *ctx, const char *name, int id_bytes, void *host);
/* debug / test taps (host copies, fp64): second_derivatives[0][i] of the last
   pf_second_derivatives (i = 0..5 <-> 11,22,33,12,13,23; src/LPT.c:36-44),
   compact [n/nranks][n][n]; LPT source spectra kvector_2LPT/3LPT_1/3LPT_2
   (which = 0,1,2) regrouped to the boundary layout [n/nranks][n][n/2+1][2]
   (one all-to-all when nranks > 1; collective). */
int pf_get_second_derivative(pf_ctx *ctx, int i, double *host);
int pf_get_kvector(pf_ctx *ctx, int which, double *host);
int pf_get_density(pf_ctx *ctx, double *host);  /* resident delta(k), boundary layout */
/* test tap: rows kx0 .. kx0 + nkx - 1 of the replicated spectrum (pf_replicated_spectrum) as this rank transforms it,
   [nkx][n (ky)][n/2+1] complex fp64, natural order (gathered -- or, behind the loopback exchange, generated -- first) */
int pf_debug_replicated_rows(pf_ctx *ctx, int kx0, int nkx, double *host);
/* The FFT-module seam of the reference (src/pinocchio.h:551-562), host in and host out in the boundary layouts
   (this rank's x-slab: real [n/nranks][n][n], spectrum [n/nranks][n][n/2+1][2]); collective over the ranks.
   forward_transform / reverse_transform (src/fmax-pfft.c:191-228): unnormalised r2c, and c2r followed by the
   1/N^3 normalisation.  Callers outside the hot path: the density writer (src/pinocchio.c:146-150) and
   ReadWhiteNoise.c:161-222. */
int pf_forward_transform(pf_ctx *ctx, const double *real_host, double *spec_host);
int pf_reverse_transform(pf_ctx *ctx, const double *spec_host, double *real_host);
/* compute_derivative(ThisGrid, first_derivative, second_derivative) (src/fmax-pfft.c:255-441) on a caller-held
   spectrum: real = c2r[ spec * G * exp(-k^2 rs^2/2) * growth(order) ] / N^3, with G = k_a k_b / k^2 (a, b in 1..3),
   i k_a / k^2 (one of them 0), -1/k^2 (both 0) (greens_function :444-456, the re/im swap :379-384); the k = 0 mode
   is left as it is.  rs in cells (Rsmooth); order = ScaleDep.order: 0 no growth, 1..4 the multiplier of
   pf_set_growth / pf_set_growth_table.  The spectrum on the host is not modified (the reference filters
   cvector_fft in place; nothing reads it afterwards). */
int pf_derivative(pf_ctx *ctx, const double *spec_host, int first_derivative, int second_derivative, double rs_cells,
                  int order, double *real_host);
/* test tap: the elementary functions the solver's default arithmetic uses on the device, which =
   0 a/b, 1 sqrt(a), 2 acos(a), 3 log10(a), 4 sin (b != 0) or cos (b == 0) of a in [0, pi/3], 5 a^0.333333333333333, 6 a/9,
   7 exp(a), 8 10^a, 9 / 10 the raw hardware seeds v_rcp_f64(a) / v_rsq_f64(a) */
int pf_debug_math(pf_ctx *ctx, int which, const double *a, const double *b, size_t count, double *out);
/* measurement aid (bench.py): GB/s of a kernel that only reads (kind 0), only writes (1) or copies (2) one spectrum-sized field of this
   context, `reps` launches between two events -- the streaming rates of this memory system, beside which the transform passes are
   reported.  No counterpart in the reference. */
int pf_debug_stream_rate(pf_ctx *ctx, int kind, int reps, double *gbps);
/* test tap without a context: ONE pass kernel on a batch of lines, host in / host out in fp64 (converted to field_bytes on
   the way); every instantiation of the hand-written transforms -- N = 2048 of BASELINE config 5 included, whose box does
   not fit one GPU -- can so be compared line by line with an independent transform (tests/test_gpu_lines.py).
   pass 0 / 1: the strided x / y pass, inverse / forward, complex [nouter][n][ncols] in and out, with the k-space factor
   `mul` (0 one, 1 k, 2 k^2, 3 i k) along the transformed axis, loads beyond |wavenumber| > band treated as zeros, and with
   pre != 0 the first-pass filter exp(-k^2 rs^2 / 2) growth / k^2 of compute_derivative (src/fmax-pfft.c:366-373);
   pass 2: z-pass c2r, complex [nouter][n/2+1] -> real [nouter][n], unnormalised (reverse_transform without its 1/N^3);
   pass 3: z-pass r2c, real [nouter][n] -> complex [nouter][n/2+1] (forward_transform);
   pass 4: the six-rows-to-three-invariants z-pass of the sweep, complex [6][nouter][n/2+1] -> real [3][nouter][n], followed by
           one more double: 1.0 when a cell raised the q == 0 flag (see pf_debug_invariant_reruns). */
int pf_debug_lines(int field_bytes, int n, int pass, int mul, int band, int nouter, int ncols, int pre, double rs,
                   double growth, int outer_offset, const double *in, double *out);
/* test tap without a context: one operation of the packed (re, im) fp32 algebra the sixteen-point strided pass is written in
   (csrc/pf_fft16.h) on `count` pairs: which = 0 / 1 a +- i b, 2 / 3 +- i a, 4 / 5 a * b, a * conj(b), 6 / 7 the same with b[0] in scalar
   registers, 8 / 9 a * (c +- i s) for a constant, 10 a - i b */
int pf_debug_pk(int which, const float *a, const float *b, float *out, int count);
/* test tap without a context: the chirp-z 3-D transforms of the general path (csrc/pf_gfft.hip: any even n in 4..2048, no library) on
   host arrays in the natural layouts; dir > 0: spectrum [n][n][n/2+1] complex -> real [n][n][n] (unnormalised), dir < 0: real -> spectrum */
int pf_debug_gfft(int n, int dir, const double *in, double *out);
/* ... and ONE chirp-z pass on a few lines of n points (any even n in 4..2048, so that the convolution lengths 1024, 2048 and 4096 --
   whose n^3 boxes no test can afford -- are run too).  mode 0: complex lines laid out [n][nlines] as the x- and y-passes meet them,
   dir > 0 inverse / < 0 forward; mode 1: Hermitian rows [nlines][n/2+1] -> real rows [nlines][n]; mode 2: real rows -> Hermitian rows */
int pf_debug_gfft_lines(int n, int mode, int dir, int nlines, const double *in, double *out);
/* test tap without a context: ONE strided (inverse) launch with several jobs as the passes of the sweep issue them: job j transforms
   input field in_of[j] (of `nin` complex fields [nouter][n][ncols], fp64 on the host) with the factor mul[j] (0 one, 1 k, 2 k^2, 3 i k)
   along the transformed axis into out[j] ([njobs][nouter][n][ncols]); jobs on the same input must be adjacent.  n a power of two. */
int pf_debug_strided_jobs(int field_bytes, int n, int njobs, int nin, const int *in_of, const int *mul, int nouter, int ncols,
                          const double *in, double *out);
/* how many sweeps of this context were repeated with six components per cell because the invariant z-pass met a tensor
   with q == 0 that is not exactly isotropic (the reference's "already diagonal" branch, src/collapse_times.c:722-727) */
int pf_debug_invariant_reruns(pf_ctx *ctx);
/* 1 when the last sweep of this context ran the collapse solve of its invariant radii on the solve stream, beside the z-pass of
   the radius that follows (PF_SOLVE_BESIDE_Z): HIP-event spans then overlap (the solve trails into the passes after
   that z-pass too) -- per-kernel times of pf_kernel_stats are spans, not kernel times or shares of the step; 0 when every kernel ran in line */
int pf_solve_ran_beside_zpass(pf_ctx *ctx);
/* which transforms serve the context's grid size (the reference plans any GridSize, src/fmax-pfft.c:139-188): 0 the hand-written
   power-of-two passes, 1 the hand-written passes with mixed-radix stage plans (n = 8 m, m = 2^a 3^b 5^c; any number of ranks that divides n),
   2 chirp-z transforms, one 3-D transform per component (any other even n on one rank, or PF_GENERAL=1) */
int pf_transform_path(pf_ctx *ctx);
/* 1: in the default (fast) arithmetic the inverse growing mode of radius `ismooth` (-1: the shared spline) comes from the
   polynomial table built from its knots by pf_set_invgrow (csrc/pf_gtab.h; *max_rel_err: its largest relative error against
   the composite 10^(-S(log10 D)) in long double; NaN when the build was refused before it got to check one: meaningful with
   status 1, or with status 0 of a table refused for its error); 0: the series forms are used (table refused, PF_GTAB=0, PF_EXACT_LIBM=1) */
int pf_invgrow_table_status(pf_ctx *ctx, int ismooth, double *max_rel_err);
/* per-cell solver on a list of Hessians d[6*count] -> F[count] (tests of
   inverse_collapse_time, src/collapse_times.c:679-776), ismooth selects the spline */
int pf_collapse_cells(pf_ctx *ctx, int ismooth, const double *d, size_t count, double *F);

/* --- measurement --- */
int pf_get_cputime(pf_ctx *ctx, pf_cputime *t);
int pf_reset_cputime(pf_ctx *ctx);
/* per-kernel-class HIP-event statistics over the launches since the last reset
   (PF_FLAG_TIMING).  Fills up to `max` entries; returns the count in *n. */
typedef struct {
  char     name[48];
  uint64_t launches;
  double   total_ms;
  double   alg_bytes;   /* algorithmic HBM bytes summed over those launches */
} pf_kernel_stat;
int pf_kernel_stats(pf_ctx *ctx, pf_kernel_stat *out, int max, int *n);
int pf_reset_kernel_stats(pf_ctx *ctx);
int pf_synchronize(pf_ctx *ctx);
/* bytes of device memory held by the context */
size_t pf_device_bytes(pf_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PINFMAX_H */
