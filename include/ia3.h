/* ia3.h — C ABI of libia3.so: MI355X-native kernels for ImageAnalysis3's per-FOV spot-calling path.
 *
 * The reference has no FFI on this path (pure Python); these entry points are what
 * thin ctypes shims carrying the reference's dotted names bind.  Each entry cites the reference
 * interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - stacks are C-order (Z, X, Y) ("z, x, y" = array axes 0,1,2), dtype IA3_U16 or IA3_F32;
 *   - host pointers are borrowed for the duration of the call; outputs are caller-allocated;
 *   - every function returns 0 on success or a negative IA3_E* code; ia3_last_error() gives
 *     the message of the last failure on the calling thread;
 *   - device state is created lazily by the first call in a process (safe after fork);
 *   - `ia3_stack` handles keep a stack resident in HBM so filter -> seed -> fit chain without
 *     host round trips;
 *   - threads and streams: every host thread that calls the library gets HIP streams of its own; `_dev` entry
 *     points queue their kernels on the calling thread's stream and may return before they have run, calls of one
 *     thread execute in the order they were made, and anything that returns data to the host (downloads, seed and
 *     spot tables, drifts) waits for what it needs.  A resident stack is handed to ANOTHER host thread only after
 *     ia3_sync() on the thread that produced it; ia3_fit_fovs, which works on threads of its own, orders them after
 *     the calling thread's stream by itself.
 */
#ifndef IA3_H
#define IA3_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IA3_U16 0
#define IA3_F32 1

#define IA3_OK 0
#define IA3_EINVAL (-1)   /* bad argument (shim raises ValueError / IndexError as the reference does) */
#define IA3_EHIP (-2)     /* HIP runtime / kernel failure */
#define IA3_ENOMEM (-3)
#define IA3_ECAPACITY (-4) /* caller-provided output buffer too small; *n_out holds the needed size */
#define IA3_EUNSUPPORTED (-5)

#define IA3_MODE_REFLECT 0  /* scipy.ndimage 'reflect' (half-sample symmetric) */
#define IA3_MODE_NEAREST 1  /* scipy.ndimage 'nearest' (edge replicate) */
#define IA3_MODE_CONSTANT 2 /* map_coordinates 'constant' (cval) */

typedef struct ia3_stack ia3_stack;

/* ---- runtime ------------------------------------------------------------------------------- */
int ia3_init(int device);                 /* select device (default: LOCAL_RANK or 0); idempotent */
const char* ia3_last_error(void);
const char* ia3_version(void);
int ia3_device_name(char* buf, int len);
int ia3_sync(void);                       /* hipStreamSynchronize on the library stream */
void* ia3_stream(void);                   /* hipStream_t the kernels are launched on */
int ia3_release_workspace(void);          /* drop cached device scratch buffers */
/* The long Gaussian passes of get_seeds keep a whole z column in registers, one kernel per stack depth: the library is
   built with the depths of FOLD_DEPTHS (25 30 33 35 40 45 50 60), any other depth from 16 to 64 planes is compiled at run
   time (hiprtc: ~35 s once per depth, dtype and machine, kept in IA3_RTC_CACHE or <library dir>/_rtc; IA3_RTC=0 turns it
   off) the first time it is used.  This call does that ahead of time.  Returns 1 = built in, 2 = compiled at run time (or
   taken from the cache), 0 = not available (the sliding-window kernels run: same results, slower), < 0 = error.  The
   reference takes any single_im_size (io_tools/load.py:166-180). */
int ia3_prepare_depth(int dtype, int Z);
/* the scratch cache: out6 = {idle bytes, bytes in use, blocks, hipMalloc calls, hipFree calls, ms spent in both} since
   the library was loaded; a steady-state loop adds no calls (each hipFree synchronises the device) */
int ia3_workspace_stats(double* out6);
/* per-kernel timing with HIP events on the library stream (off by default) */
int ia3_profile_enable(int on);
int ia3_profile_collect(char* buf, int len); /* "kernel,count,total_ms\n" lines since last collect */
/* Test / tuning knobs; results never depend on them.  IA3_TUNE_GAUSS_CERT: guard distance (f64 ulps) of the
 * certified fused-multiply-add path of the long Gaussian passes; -2 default (4R+8), -1 path off (always the
 * reference operation sequence), large values send every output through the reference sequence after the check. */
#define IA3_TUNE_GAUSS_CERT 1
/* IA3_TUNE_DFT_VALU: 1 = the upsampled-DFT contractions of the phase correlation run on the vector unit (first
 * version) instead of the f64 matrix cores; shifts agree to rounding. */
#define IA3_TUNE_DFT_VALU 2
/* IA3_TUNE_UPLOAD_THREADS: helper threads that copy a host array into the pinned staging ring of ia3_stack_upload
 * (environment IA3_UPLOAD_THREADS); 0 (default: it measured fastest, 56 GB/s) = one plain hipMemcpyAsync from
 * pageable memory. */
#define IA3_TUNE_UPLOAD_THREADS 3
/* IA3_TUNE_SEED_DENSE: 1 = get_seeds always runs all three passes of the background filter on the whole stack; 0
 * (default) = the lazy form: the axis-0 pass everywhere, the other two only around candidate maxima (seed.hip).  Seeds
 * are identical either way. */
#define IA3_TUNE_SEED_DENSE 4
/* IA3_TUNE_FIT_NBLIST: longest per-seed neighbour list the fit kernel reads (default and maximum 64); seeds with more
 * overlapping neighbours scan the seed list instead — same neighbours in the same order.  Tests lower it to drive
 * ordinary fields through the scan. */
#define IA3_TUNE_FIT_NBLIST 5
/* IA3_TUNE_FFT_C2C: 1 = the phase correlation transforms the (real) stacks with complex-to-complex FFTs (first
 * version); 0 (default) = real-input transforms on half spectra.  Shifts agree to rounding. */
#define IA3_TUNE_FFT_C2C 6
/* IA3_TUNE_FIT_FUSE: 1 (default) = in ia3_fit_run a seed whose ball overlaps no other seed's gets its first fit and
 * sweep 1 from ONE wavefront (same voxels, gathered once; one hand-over); 0 = two work-list positions as for every other
 * seed.  Tables are identical bit for bit. */
#define IA3_TUNE_FIT_FUSE 7
/* IA3_TUNE_GAUSS_FOLD: 1 (default) = the axis-0 pass of a long filter (radius >= 16) over a stack of 30, 40 or 50
 * planes holds the column in registers and folds the border into the weights (certified like the fused path, same
 * fallback); 0 = the sliding-window pass for every depth.  Results are identical bit for bit. */
#define IA3_TUNE_GAUSS_FOLD 8
/* IA3_TUNE_SEED_STRIPS: 1 (default) = where the column kernel of IA3_TUNE_GAUSS_FOLD runs for the seed detector and the
 * row length is a multiple of 32, the lower bound of the lazy background filter is made from minima that kernel takes from
 * its registers (per group of planes, row and 32 columns); 0 = from a pass over the stored axis-0 result (per plane).  Only
 * the number of first-stage candidates can differ, never a seed. */
#define IA3_TUNE_SEED_STRIPS 9
/* IA3_TUNE_FIT_WAVES: persistent wavefronts of the fit kernel per SIMD, 1 or 2 (default 2: the kernel is built for 256
 * registers).  Tables are identical bit for bit. */
#define IA3_TUNE_FIT_WAVES 10
/* IA3_TUNE_FIT_MERGE: 1 (default) = once the first launch (first fits + sweep 1) has left fewer than 1 seed in 64
 * unconverged, all remaining sweeps go out in one launch, ordered by the work list's dependencies alone (a seed's
 * sweep k+1 no longer waits for the slowest fit of sweep k anywhere in the batch); 0 = two sweeps per launch throughout.
 * Tables are identical bit for bit. */
#define IA3_TUNE_FIT_MERGE 11
/* IA3_TUNE_WARP_ONEPASS: the cubic warp's spline prefilter along the two long axes.  64 (default) = every sample read
 * once and written once: the anticausal recursion of a tile starts from a value certified by two bounding chains over
 * the samples behind it (warp.hip), with the two-sweep recursion over the rest of the line where the chains do not
 * meet; 1..63 = the same with that many warm-up samples (tests: short warm-ups fail often and drive lines through the
 * fallback); 0 = two sweeps over whole lines; -1 = two sweeps and the one-output-per-thread gather.  Results are
 * identical bit for bit. */
#define IA3_TUNE_WARP_ONEPASS 12
/* IA3_TUNE_FIT_KDQ: entries of the per-voxel queue of the Voronoi tie queries on the device (1..24, default 24); a query
 * that needs more is finished on the host with the same tree (tests lower it to drive fields through that path).  Tables
 * are identical bit for bit. */
#define IA3_TUNE_FIT_KDQ 13
/* IA3_TUNE_FIT_MEMO: 1 (default) = a repeat sweep does not run a refit whose inputs — the image ball and the records of
 * the overlapping seeds — are what they were at the seed's previous refit: the fit would return the same row and the
 * seed is marked converged, as the reference's loop does after repeating it (External/Fitting_v4.py:651-680); 0 = every
 * refit is run.  Tables and sweep counts are identical bit for bit; the evaluation counters count the fits that ran. */
#define IA3_TUNE_FIT_MEMO 14
/* IA3_DEBUG_FIT_MAXFEV: PROFILING ONLY, changes results: > 0 caps the function evaluations of every fit (MINPACK's maxfev),
 * which splits the fit kernel's time into its fixed and its per-evaluation part; 0 (default) = the reference's limits. */
#define IA3_DEBUG_FIT_MAXFEV 100
/* IA3_DEBUG_FIT_WAITBOUND: TESTS ONLY: polls after which a refit that waits for the fits it depends on gives up and the
 * launch aborts with IA3_EHIP ("a dependency wait exceeded its bound"); <= 0 (default) = 2^22 polls (~28 s, never reached
 * by a live kernel).  A small value makes ordinary waits of a crowded field trip it, which is how the abort path is
 * exercised; the call fails, nothing else changes. */
#define IA3_DEBUG_FIT_WAITBOUND 101
int ia3_set_tuning(int key, int value);

/* ---- device-resident stacks ----------------------------------------------------------------- */
int ia3_stack_upload(const void* host, int dtype, int Z, int X, int Y, ia3_stack** out);
int ia3_stack_alloc(int dtype, int Z, int X, int Y, ia3_stack** out);
/* Raw uint16 movie file -> resident (frames, X, Y) stack: what DaxReader(dax).loadAll() followed by an upload does
 * (visual_tools.py:974-1083), read in pieces through pinned staging buffers so file read and PCIe copy overlap.
 * offset_bytes: start of the first frame; big_endian != 0 swaps bytes on the device. */
int ia3_stack_load_file(const char* path, long long offset_bytes, int frames, int X, int Y, int big_endian,
                        ia3_stack** out);
int ia3_stack_wrap(void* devptr, int dtype, int Z, int X, int Y, ia3_stack** out); /* borrow device memory */
/* io_tools/load.py:524-550 split_im_by_channels on a resident raw movie (frames, X, Y): frames start, start+step, .. */
int ia3_stack_deinterleave(const ia3_stack* raw, int start, int step, int Z, ia3_stack** out);
/* device buffers for run-constant data (correction profiles) */
int ia3_buffer_upload(const void* host, size_t bytes, void** devptr);
void ia3_buffer_free(void* devptr);
int ia3_stack_download(const ia3_stack* s, void* host);
int ia3_stack_info(const ia3_stack* s, int* dtype, int* Z, int* X, int* Y, void** devptr);
void ia3_stack_free(ia3_stack* s);

/* ---- filters ---------------------------------------------------------------------------------
 * scipy.ndimage.gaussian_filter semantics: per axis 0,1,2 a 1-D correlation accumulated in
 * float64 in NI_Correlate1D's order, re-quantised to the stack dtype after every axis (float32
 * rounds, uint16 truncates).  `weights` (2*radius+1 doubles) may be NULL, then they are
 * computed from (sigma, truncate) as scipy does.
 * Replaces: scipy.ndimage.gaussian_filter as called at spot_tools/fitting.py:92,99 and
 * correction_tools/filter.py:16. */
int ia3_gaussian_filter(const void* im, int dtype, int Z, int X, int Y, double sigma, double truncate,
                        int mode, const double* weights, int radius, void* out);
int ia3_gaussian_filter_dev(const ia3_stack* im, double sigma, double truncate, int mode,
                            const double* weights, int radius, ia3_stack* out);

/* correction_tools/filter.py:14-19 gaussian_high_pass_filter(image, sigma=5, truncate=2):
 * out = image - lowpass(nearest); out[lowpass > image] = 0, in the stack dtype. */
int ia3_gaussian_highpass(const void* im, int dtype, int Z, int X, int Y, double sigma, double truncate,
                          const double* weights, int radius, void* out);
int ia3_gaussian_highpass_dev(const ia3_stack* im, double sigma, double truncate,
                              const double* weights, int radius, ia3_stack* out);

/* correction_tools/filter.py:22-42 Remove_Hot_Pixels (and its twin corrections.py:490-510):
 * z-vote of im > hot_th * mean(4 rolled neighbours) -> column replaced by its 4-neighbour mean. */
int ia3_remove_hot_pixels(const void* im, int dtype, int Z, int X, int Y, double hot_pix_th, double hot_th,
                          void* out, int* n_hot);

/* Pre-corrections of io_tools/load.py:337-384 (correct_fov_image), uint16 outputs with NumPy's cast rules.
 * corrections.py:479-487 Z_Shift_Correction: out = u16(f32(im) / median_z * median_all); medians_out (Z+1
 * floats, per-plane then whole-stack) may be NULL. */
int ia3_z_shift_correction(const void* im, int dtype, int Z, int X, int Y, void* out_u16, float* medians_out);
/* io_tools/load.py:373-384: out = u16(f32(im) / profile[None]); profile (X,Y), prof_dtype 1 = float32, 2 = float64 */
int ia3_illumination_correct(const void* im_u16, int Z, int X, int Y, const void* profile, int prof_dtype,
                             void* out_u16);
/* io_tools/load.py:348-370: outs[i] = u16(clip(sum_j ims[j] * profile[i,j])); profile (C,C,X,Y), C <= 8 */
int ia3_bleedthrough_correct(const void* const* ims_u16, int C, int Z, int X, int Y, const void* profile,
                             int prof_dtype, void* const* outs_u16);

/* ---- seeding: spot_tools/fitting.py:20-154 get_seeds ------------------------------------------ */
typedef struct ia3_seed_params {
  double th_seed;               /* threshold on max_im - min_im */
  double gfilt_size;            /* 0.75 ; 0 = no front filter */
  double background_gfilt_size; /* 7.5  ; 0 = no background filter */
  int filt_size;                /* 3 */
  int min_edge_distance;        /* 2 */
  int use_dynamic_th;           /* 1 */
  int dynamic_niters;           /* 10 */
  int min_dynamic_seeds;        /* 1 */
  int remove_hot_pixel;         /* 1 */
  int hot_pixel_th;             /* 3 */
  int max_num_seeds;            /* <=0: unlimited */
  int th_compare_f32;           /* 1: each threshold level is rounded to float32 before the compare
                                   (NumPy>=2 weak-scalar rule for a Python-float th_seed, which is
                                   what fit_fov_image's float(th_seed) produces); 0: float64 compare
                                   (np.float64 th_seed, e.g. the percentile path) */
  const double* w_front; int r_front; /* optional explicit taps (NULL: from sigma, truncate 4) */
  const double* w_back;  int r_back;
} ia3_seed_params;

/* out_zxyh: capacity x 4 doubles [z,x,y,h], brightest first. */
int ia3_dog_seed(const void* im, int dtype, int Z, int X, int Y, const ia3_seed_params* p,
                 double* out_zxyh, int capacity, int* n_out, double* th_used);
int ia3_dog_seed_dev(const ia3_stack* im, const ia3_seed_params* p,
                     double* out_zxyh, int capacity, int* n_out, double* th_used);

/* The two filtered stacks get_seeds compares (spot_tools/fitting.py:83-96), as the seed detector computes them: `front`
 * = scipy.ndimage.gaussian_filter(im, sigma_front) complete, `back_axis0` = the FIRST pass of
 * gaussian_filter(im, sigma_back), i.e. scipy.ndimage.gaussian_filter1d(im, sigma_back, axis=0) — the detector runs the
 * other two passes of the background filter only around candidate maxima.  Both 'reflect', truncate 4, bit-identical
 * to SciPy.  On stacks of 30 / 40 / 50 planes with the default sigmas (0.75, 7.5) the two axis-0 passes share one
 * launch; every other case runs the separate filters.  Outputs must not alias the input or each other. */
int ia3_dog_filters_dev(const ia3_stack* im, double sigma_front, double sigma_back, ia3_stack* front, ia3_stack* back_axis0);

/* ---- the pre-correction chain of io_tools/load.py:323-384 on stacks that stay resident ---------------------------
 * ia3_remove_hot_pixels_dev works in place; float_arith != 0 on a uint16 stack = the chain's
 * corrections.Remove_Hot_Pixels(im.astype(np.float32), dtype=np.uint16) (float32 votes and means, one truncation at
 * the end).  Profiles are device buffers from ia3_buffer_upload; prof_dtype 1 = float32, 2 = float64.  Outputs of the
 * bleedthrough mix must not alias its inputs; the other two may run in place. */
int ia3_remove_hot_pixels_dev(ia3_stack* im, double hot_pix_th, double hot_th, int float_arith, int* n_hot);
int ia3_z_shift_correction_dev(const ia3_stack* im, ia3_stack* out_u16);
int ia3_illumination_correct_dev(const ia3_stack* im_u16, const void* profile_dev, int prof_dtype, ia3_stack* out_u16);
int ia3_bleedthrough_correct_dev(ia3_stack* const* ims_u16, int C, const void* profile_dev, int prof_dtype,
                                 ia3_stack* const* outs_u16);
/* the step-API twins in classes/preprocess.py (DaxProcesser._corr_illumination :605-680, ._corr_bleedthrough :464-541):
 * same quotient / mix (the mix accumulated in float64), then `(im - min) / (max - min) * 65535 + 0` when rescale != 0,
 * clip to [0, 65535], truncation */
int ia3_illumination_rescale_dev(const ia3_stack* im_u16, const void* profile_dev, int prof_dtype, int rescale,
                                 ia3_stack* out_u16);
int ia3_bleedthrough_rescale_dev(ia3_stack* const* ims_u16, int C, const void* profile_dev, int prof_dtype, int rescale,
                                 ia3_stack* const* outs_u16);

/* ---- background level: io_tools/load.py:642-687 find_image_background -----------------------------------
 * counts = np.histogram(im, bins=edges); highest strict local maximum of the counts (scipy.signal.find_peaks,
 * height halved from size/50 at most max_iter times); background = centre of that bin; np.nanmedian(im) when no
 * bin qualifies.  edges: the reference's np.arange(iinfo(dtype).min, iinfo(dtype).max, bin_size) as float64
 * (n_edges <= 16001). */
int ia3_find_background(const void* im, int dtype, int Z, int X, int Y, const double* edges, int n_edges,
                        int max_iter, double* background);
int ia3_find_background_dev(const ia3_stack* im, const double* edges, int n_edges, int max_iter, double* background);
/* spot_tools/fitting.py:246-258 (normalize_local=True): the same statistic over
 * io_tools/crop.py:59-88 generate_neighboring_crop(center, crop_size) for each of n centres (n x 3 float32, the
 * fitted [z,x,y] columns of the spot table), one block per spot. */
int ia3_local_background_dev(const ia3_stack* im, const float* centers_zxy, int n, int crop_size,
                             const double* edges, int n_edges, int max_iter, double* backgrounds);

/* ---- legacy per-cell seeding: visual_tools.py:1775-1870 get_seed_in_distance (+ :348-381 get_seed_points_base),
 * the seeder of classes/__init__.py:57-88 `_fit_single_image`.  Crop of +-seed_radius (x, y) / +-seed_radius/2 (z)
 * around `center` (NULL: whole image), scipy-default Gaussian filters (reflect, truncate 4), int64-truncated
 * rank-filter tests, strict '>' threshold lowered over np.linspace(1, 1/dynamic_iters, dynamic_iters) until enough
 * seeds lie within seed_radius, brightest num_seeds first.  th_seed is the resolved threshold (the shim evaluates
 * seed_by_per).  out_zxyh: capacity x 4 int64 [z, x, y, h]. */
typedef struct ia3_legacy_seed_params {
  int num_seeds;                 /* 0 = keep all */
  double seed_radius;            /* 30 */
  double gfilt_size;             /* 0.75 */
  double background_gfilt_size;  /* 10 */
  int filt_size;                 /* 3 */
  double th_seed;                /* 300 */
  int dynamic;                   /* 1 */
  int dynamic_iters;             /* 10 */
  int min_dynamic_seeds;         /* 2 */
  int hot_pix_th;                /* 4 */
} ia3_legacy_seed_params;
int ia3_seed_in_distance(const void* im, int dtype, int Z, int X, int Y, const double* center,
                         const ia3_legacy_seed_params* p, int64_t* out_zxyh, int capacity, int* n_out);

/* ---- fitting: External/Fitting_v4.py:559-683 iter_fit_seed_points (+ GaussianFit :165-396) ---- */
typedef struct ia3_fit_params {
  int radius_fit;            /* 5 */
  double min_delta_center;   /* 1.0  (firstfit) */
  double max_delta_center;   /* 2.5  (repeatfit) */
  int n_max_iter;            /* 10 */
  double max_dist_th;        /* 0.1 */
  double min_w, max_w, init_w; /* 0.5, 4, 1.5 */
  /* 0: External/Fitting_v4.py (production).  1: External/Fitting_v3.py:50-262,312-425, the fitter behind the
   * legacy classes/__init__.py:57-88 `_fit_single_image`: per-axis start widths init_w_zxy (the reference's
   * global _sigma_zxy = 1.35, 1.9, 1.9), its to_center (:81-87) and MINPACK's default maxfev. */
  int model_variant;
  double init_w_zxy[3];
} ia3_fit_params;

typedef struct ia3_fitter ia3_fitter;

/* centers_zxy: n x 3 doubles (the reference's `centers.T`).  The stack must stay alive until
 * ia3_fit_destroy. */
int ia3_fit_create(const ia3_stack* im, const double* centers_zxy, int n, const ia3_fit_params* p,
                   ia3_fitter** out);
int ia3_fit_first(ia3_fitter* f);                 /* .firstfit()  */
int ia3_fit_repeat(ia3_fitter* f, int* n_iter);   /* .repeatfit() */
int ia3_fit_run(ia3_fitter* f);                   /* .firstfit(); .repeatfit() in one persistent launch */
/* ps: n x 11 float32 rows [h,z,x,y,bk,sz,sx,sy,sin_t,sin_p,eps] (NaN rows for failed fits);
 * success: n bytes; nvox: n ints (voxels used by the last fit of each seed); any may be NULL. */
int ia3_fit_results(ia3_fitter* f, float* ps, uint8_t* success, int* nvox);
/* same, plus n_iter of the last repeatfit, with a single stream synchronisation */
int ia3_fit_results_ex(ia3_fitter* f, float* ps, uint8_t* success, int* nvox, int* n_iter);
/* nfev: n ints, model evaluations spent on each seed so far (first fit + sweeps); MINPACK stops a fit at maxfev */
int ia3_fit_nfev(ia3_fitter* f, int* nfev);
int ia3_fit_stats(ia3_fitter* f, int64_t* total_fits, int64_t* total_nfev);
/* the fitter's 32 device counters: [0] fits, [1] evaluations, [2] voxel evaluations; [8..31] shader cycles per phase in
 * a profiling build of the fit kernel (-DIA3_FIT_STAMPS, scripts/fit_stamps.sh), zero in the shipped library */
int ia3_fit_counters(ia3_fitter* f, uint64_t* out32);
void ia3_fit_destroy(ia3_fitter* f);

/* External/Fitting_v4.py:165-396 GaussianFit(im, X, center, ...).fit() for a batch of explicit voxel lists
 * (<= 512 voxels each): vals / coords_zxy concatenated, off[n_fits+1]; cfg4 = (delta_center, min_w, max_w,
 * init_w) per fit; kind = dtype class of the caller's values (0 float32, 1 integer, 2 float64), which selects
 * NumPy's arithmetic for the start point.  ps: n_fits x 11 (NaN row if < 10 voxels); xs: n_fits x 10
 * unconstrained solution (may be NULL); info: n_fits x 2 (success, nfev) (may be NULL). */
int ia3_gaussfit_voxels(const double* vals, const int* coords_zxy, const int* off, int n_fits,
                        const double* centers, const double* cfg4, const int* kind, float* ps, double* xs,
                        int* info);

/* one call: upload + firstfit + repeatfit */
int ia3_fit_seeds(const void* im, int dtype, int Z, int X, int Y, const double* centers_zxy, int n,
                  const ia3_fit_params* p, float* out_ps, uint8_t* success, int* n_iter);

/* seed + fit chained on a resident stack (the bench's hot path): fit_fov_image
 * (spot_tools/fitting.py:169-262) without the optional normalisation.  out_rows: capacity x 11
 * float32, NaN rows and out-of-image centres removed (fitting.py:232-237). */
int ia3_fit_fov_dev(const ia3_stack* im, const ia3_seed_params* sp, const ia3_fit_params* fp,
                    float* out_rows, int capacity, int* n_rows, int* n_seeds, int* n_iter);

/* fits run, model evaluations and voxel evaluations (sum over fits of evaluations x voxels) of the calling thread's
 * last ia3_fit_fov_dev: what the counted-flop rate of the fit kernel is computed from (bench.py). */
int ia3_fit_fov_stats(int64_t* fits, int64_t* nfev, int64_t* voxel_evals);
/* of the calling thread's last ia3_fit_fov_dev: shader cycles its fit waves spent waiting for the fits a refit depends on
 * (the reference's ordered sweeps, Fitting_v4.py:651-680), and the cycles those waves lived: the dependency-wait share */
int ia3_fit_fov_wait_share(int64_t* wait_cycles, int64_t* wave_cycles);

/* A batch of independent FOVs in one call: the per-image tasks the reference spreads over an mp.Pool
 * (classes/field_of_view.py:1129-1142, worker classes/batch_functions.py:60).  Each job is either a host stack
 * (uploaded here through pinned staging buffers while other jobs compute) or a resident one; `in_flight` jobs
 * (<= 0: 4, at most 64) are seeded side by side on library-owned threads and streams (at most 16) and fitted in groups
 * by one more.  Resident jobs may still be in production on the calling thread's stream: the batch waits for it.  Per job the outputs are those
 * of ia3_fit_fov_dev; the call returns the first failing job's code (every job's own code is in `rc`). */
typedef struct ia3_fov_job {
  const void* host;       /* (Z,X,Y) stack of `dtype` in host memory, or NULL */
  const ia3_stack* dev;   /* resident stack (used when not NULL) */
  float* rows;            /* capacity x 11 float32 */
  int capacity;
  int n_rows, n_seeds, n_iter, rc;          /* out */
  long long fits, nfev, voxel_evals;        /* out: see ia3_fit_fov_stats */
} ia3_fov_job;
int ia3_fit_fovs(ia3_fov_job* jobs, int n_jobs, int dtype, int Z, int X, int Y, const ia3_seed_params* sp,
                 const ia3_fit_params* fp, int in_flight);

/* ---- drift ------------------------------------------------------------------------------------
 * alignment_tools.py:286-328 fftalign_2d: (xt, yt) of the normalised full cross-correlation peak of two
 * 2-D float64 images inside a +-max_disp window around `center`. */
int ia3_fftalign_2d(const double* im1, int s1x, int s1y, const double* im2, int s2x, int s2y,
                    const double* center, double max_disp, int* out_xy);
/* alignment_tools.py:330-353 fft3d_from2d (gb <= 1): integer (tz, tx, ty) from z- then y-max-projections. */
int ia3_fft3d_from2d(const void* im1, const void* im2, int dtype, int Z, int X, int Y, double max_disp,
                     int* out_zxy);
int ia3_fft3d_from2d_dev(const ia3_stack* im1, const ia3_stack* im2, double max_disp, int* out_zxy);
/* skimage.registration.phase_cross_correlation(reference, moving, upsample_factor) as called at
 * correction_tools/alignment.py:491-494,631-632 and classes/preprocess.py:831-835 (published algorithm; pinned
 * against scikit-image 0.18.3 for normalization None, tests/golden/phase.npz).  normalization: 1 = "phase", 0 = None.  shift[3] = (dz, dx, dy) to apply to `moving`. */
int ia3_phase_xcorr3d(const void* ref, const void* mov, int dtype, int Z, int X, int Y, int upsample,
                      int normalization, double* shift, double* err, double* phasediff);
int ia3_phase_xcorr3d_dev(const ia3_stack* ref, const ia3_stack* mov, int upsample, int normalization,
                          double* shift, double* err, double* phasediff);
/* correction_tools/alignment.py:527-695 align_image, phase-correlation path (use_autocorr=True), on resident stacks: crop
 * after crop (crops: n_crops x 3 x 2 ints, [start, stop) per axis) skimage's phase_cross_correlation(ref crop, src crop,
 * upsample); as soon as >= min_good_drifts crops are in and >= min_good_drifts of them lie within drift_diff_th of their
 * mean, the mean of those is the drift (:664-674, flag 0); with no such agreement after the last crop, the mean of the
 * two closest drifts and the one nearest to both (:676-693, flag 1).  drifts_out (n_crops x 3, may be NULL) / n_used: the
 * per-crop drifts that were measured. */
int ia3_align_image_dev(const ia3_stack* src, const ia3_stack* ref, const int* crops, int n_crops, int upsample,
                        int normalization, int min_good_drifts, double drift_diff_th, double* drift, int* flag,
                        double* drifts_out, int* n_used);

/* Many images against ONE reference bead image (every movie of a run: classes/batch_functions.py:169-206): the half
 * spectra of the reference crops are computed once and kept on the device (105 MB per 50 x 512 x 512 crop), so a crop
 * costs two transforms instead of three.  ia3_align_image_ref = ia3_align_image_dev with such a reference; drifts are
 * identical.  A drift reference may be shared by several host threads; free it after their calls have returned. */
typedef struct ia3_drift_ref ia3_drift_ref;
int ia3_drift_ref_create(const ia3_stack* ref, const int* crops, int n_crops, ia3_drift_ref** out);
void ia3_drift_ref_free(ia3_drift_ref* r);
int ia3_align_image_ref(const ia3_stack* src, ia3_drift_ref* ref, int upsample, int normalization, int min_good_drifts,
                        double drift_diff_th, double* drift, int* flag, double* drifts_out, int* n_used);

/* ---- warp -------------------------------------------------------------------------------------
 * correction_tools/translate.py:5-31 warp_3d_image and its inlined twins (io_tools/load.py:438-453,
 * classes/preprocess.py:918-946): out = map_coordinates(im, grid (+ field) - drift, order, mode, cval).
 * orders 0 (nearest sample) and 1 with mode constant|nearest; order 3 (B-spline prefilter) with mode nearest (the production twins: padded by
 * 12, tuned kernels) or constant (warp_3d_image's default border mode: no padding, mirror-boundary prefilter, cval outside,
 * plain kernels; every axis >= 2 samples).
 * field: NULL or a (3,Z,X,Y) displacement field, field_dtype 1 = float32, 2 = float64; add 16 to form the
 * coordinates as (grid - drift) + field, the order of classes/preprocess.py:923-935 (DaxProcesser._warp_image),
 * instead of (grid + field) - drift. */
int ia3_warp3d(const void* im, int dtype, int Z, int X, int Y, const double* drift, const void* field,
               int field_dtype, int order, int mode, double cval, void* out);
int ia3_warp3d_dev(const ia3_stack* im, const double* drift, const void* field_dev, int field_dtype,
                   int order, int mode, double cval, ia3_stack* out);

/* new resident stack = s[z0:z1, x0:x1, y0:y1] (drift crops, correction_tools/alignment.py:617-622) */
int ia3_stack_crop(const ia3_stack* s, int z0, int z1, int x0, int x1, int y0, int y1, ia3_stack** out);

/* ---- whole round-folder movies: the per-image task of classes/batch_functions.py:60-302 batch_process_image_to_spots
 * (fanned out over an mp.Pool by classes/field_of_view.py:1027-1142), as ONE pipelined call over many movies -------------
 * Per movie: raw (frames, X, Y) uint16 movie from host memory or a .dax file -> split_im_by_channels
 * (io_tools/load.py:524-550) -> hot pixels, z shift, bleedthrough, illumination (:323-384) -> bead drift against the
 * resident reference bead stack (align_image, phase correlation) -> cubic warp with drift + dense chromatic field
 * (:424-453) -> (Gaussian high-pass :489-498) -> get_seeds + firstfit + repeatfit + row filters of every selected channel
 * (spot_tools/fitting.py:169-237) with that channel's seeding threshold.  Results are those of correct_fov_image followed
 * by fit_fov_image movie by movie; what changes is the schedule: one library thread uploads movie k+1 while
 * `correct_threads` others run the corrections, drift and warps of the movies before it on streams of their own and one
 * more fits the channels of SEVERAL movies with one group fitter (fit_group_images images per group, see ia3_fit_fovs). */
#define IA3_MOVIE_MAXCH 8
typedef struct ia3_movie_params {
  int frames, X, Y;                    /* raw movie */
  int Z;                               /* planes per channel */
  int n_load;                          /* channels taken out of the movie (<= IA3_MOVIE_MAXCH) */
  int load_start[IA3_MOVIE_MAXCH];     /* first frame of each; frames start, start + load_step, ... */
  int load_step;                       /* number of colours in the movie */
  int n_sel;                           /* selected channels: corrected images / spot tables come back for these */
  int sel[IA3_MOVIE_MAXCH];            /* index into the loaded channels */
  int hot_pixel_corr; double hot_pixel_th;   /* io_tools/load.py:323-334 (hot_pix_th 0.5, float32 arithmetic) */
  int z_shift_corr;                    /* :337-345 */
  int n_bleed;                         /* :348-370: loaded channels mixed by the bleedthrough profile (0 = off) */
  int bleed_idx[IA3_MOVIE_MAXCH];      /* ... in the profile's channel order */
  const void* bleed_profile; int bleed_dtype;            /* device buffer (C,C,X,Y); 1 float32, 2 float64 */
  const void* illum_profile[IA3_MOVIE_MAXCH]; int illum_dtype[IA3_MOVIE_MAXCH];   /* per loaded channel, NULL = none (:373-384) */
  int drift_idx;                       /* loaded channel holding the beads; < 0: no drift measurement */
  const ia3_stack* ref_bead;           /* corrected reference bead stack, resident */
  ia3_drift_ref* drift_ref;            /* optional: a drift reference made from ref_bead and `crops` (else made per call) */
  int n_crops; int crops[8][3][2];     /* generate_drift_crops */
  int precision_fold, normalization, min_good_drifts; double drift_diff_th;
  int warp;                            /* 0: images are never resampled (warp_image=False, or a silent call: :434-436) */
  int warp_always[IA3_MOVIE_MAXCH];    /* per SELECTED channel: 1 = a chromatic channel, resampled whatever the drift;
                                          0 = resampled only when the drift is non-zero (:427) */
  const void* chrom_field[IA3_MOVIE_MAXCH]; int chrom_dtype[IA3_MOVIE_MAXCH];     /* per SELECTED channel: (3,Z,X,Y) device buffer or NULL */
  double highpass_sigma, highpass_truncate;   /* sigma <= 0: no high-pass */
  int fit_spots;
  ia3_seed_params seed[IA3_MOVIE_MAXCH];      /* per SELECTED channel (th_seed differs by channel, batch_functions.py:10-17) */
  ia3_fit_params fit;
  /* spot_tools/fitting.py:240-258: heights divided by the image's background level (1, normalize_background) or by the
   * level of each spot's neighbourhood of +-crop_size (2, normalize_local); bg_edges / bg_max_iter as in
   * ia3_find_background.  0 = heights as fitted. */
  int normalize; int bg_crop_size; const double* bg_edges; int bg_n_edges; int bg_max_iter;
  int correct_threads;                 /* <= 0: 2 */
  int fit_group_images;                /* <= 0: 12 */
  int upload_ahead;                    /* raw movies resident ahead of the corrections; <= 0: 2 */
} ia3_movie_params;
typedef struct ia3_movie_job {
  const void* host_raw;                /* (frames, X, Y) uint16 in host memory, or NULL to read `path` */
  const char* path; long long offset_bytes; int big_endian;
  double drift_in[3]; int measure_drift;      /* measure_drift 0: drift_in is used as it is (flag 0) */
  void* images_out[IA3_MOVIE_MAXCH];   /* per selected channel: host buffer for the corrected (Z,X,Y) uint16 image, or NULL */
  float* rows[IA3_MOVIE_MAXCH]; int capacity[IA3_MOVIE_MAXCH];      /* per selected channel: capacity x 11 float32 */
  /* out */
  double drift[3]; int drift_flag;
  int n_rows[IA3_MOVIE_MAXCH], n_seeds[IA3_MOVIE_MAXCH], n_iter[IA3_MOVIE_MAXCH];
  int rc;
  double t_upload_ms, t_correct_ms, t_fit_ms;   /* host wall time this movie spent in each stage */
  double stamps[6];   /* ms since the call began: upload begin / end, corrections begin / seeded, (last) group fit begin / end */
} ia3_movie_job;
int ia3_process_movies(ia3_movie_job* jobs, int n_jobs, const ia3_movie_params* p);

#ifdef __cplusplus
}
#endif
#endif /* IA3_H */
