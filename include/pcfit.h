/*
 * pcfit.h - C ABI of libpcfit_hip.so: the slice -> circle fit -> boundary ->
 * residual path of a cylinder point-cloud analyzer on MI355X (gfx950).
 *
 * The reference this replaces is pure Python and has no FFI of its own; its
 * boundary is a set of Python functions.  Each entry point below names the
 * reference function (file:line, relative to the reference checkout) whose
 * arithmetic it performs for a whole batch of slices.  The Python drop-in
 * (pointcloud_amd/) binds these with ctypes.
 *
 * Conventions
 *  - Plain C: pointers and sizes only.  Functions return 0 (PCFIT_OK) or a
 *    negative PCFIT_ERR_* code; pcfit_last_error() gives the text for the
 *    calling thread.  Nothing throws, nothing aborts.
 *  - Slices are CSR segments: point i of slice s is (x[offsets[s]+i],
 *    y[offsets[s]+i]), offsets has S+1 int64 entries.
 *  - "_dev" entry points take DEVICE pointers and a HIP stream handle
 *    (hipStream_t cast to void*, NULL = default stream).  They enqueue work and
 *    return; they never allocate, free or synchronise.  Scratch memory comes
 *    from the caller (pcfit_*_workspace_bytes tells how much).
 *  - Entry points without "_dev" take HOST pointers (numpy buffers): they
 *    allocate device memory, copy in, run the same kernels, copy out,
 *    synchronise and free.  They are re-entrant: a call's results depend on its
 *    arguments only.  What the library keeps between calls is invisible in the
 *    results - the per-device cache of device blocks and the cache of RANSAC
 *    sampling tables (both mutex-guarded), the last-error text of each thread,
 *    and two counters for tools (pcfit_debug_qhull_reasons).
 *  - Per-slice status codes (int32): 0 ok, 1 too few points, 2 no 2-D
 *    variation (rank deficient), 3 RANSAC consensus below the minimum,
 *    4 singular system.
 *  - Everything is FP64; indices are int64 (slice-local unless stated),
 *    masks uint8.
 */
#ifndef PCFIT_H
#define PCFIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCFIT_OK 0
#define PCFIT_ERR_INVALID (-1)     /* bad argument */
#define PCFIT_ERR_HIP (-2)         /* a HIP runtime call failed */
#define PCFIT_ERR_WORKSPACE (-3)   /* workspace missing or too small */
#define PCFIT_ERR_UNSUPPORTED (-4) /* size beyond what the kernels handle */

#define PCFIT_ST_OK 0
#define PCFIT_ST_SPARSE 1
#define PCFIT_ST_RANK 2
#define PCFIT_ST_RANSAC 3
#define PCFIT_ST_SINGULAR 4

/* ---- library / device plumbing -------------------------------------- */
int pcfit_version(void);
const char *pcfit_last_error(void);
int pcfit_device_count(int *count);
/* The current device is a property of the calling THREAD (as in HIP): a new
 * thread starts on device 0 whatever its parent selected.  A scan handle
 * (pcfit_scan_upload) belongs to the device current in the thread that made it. */
int pcfit_set_device(int device);
int pcfit_get_device(int *device);
int pcfit_malloc(void **ptr, size_t bytes);
int pcfit_free(void *ptr);
int pcfit_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int pcfit_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int pcfit_memset(void *dst_dev, int value, size_t bytes, void *stream);
int pcfit_stream_synchronize(void *stream);
/* The host-pointer entry points keep the device blocks they allocated for the
 * next call (per device, at most PCFIT_POOL_BYTES bytes from the environment,
 * default 16 GiB; first use of fresh device memory costs more than the kernels).
 * This gives every cached block back to the driver.  No counterpart in the
 * reference; the _dev entry points never allocate and are not concerned. */
int pcfit_release_cache(void);

/* ---- RANSAC sampling tables (host side, exact integer work) --------- */
/* PCG64 state numpy's default_rng(seed) starts from: state_hi, state_lo,
 * inc_hi, inc_lo.  circle_fitting.py:124. */
int pcfit_seed_state(uint64_t seed, uint64_t state[4]);
/* The index triples rng.choice(n, size=3, replace=False) returns on calls
 * 0..trials-1 (circle_fitting.py:126).  The draw precedes every rejection, so
 * the table depends on (state, n, trials) only.  out: trials*3 int32. */
int pcfit_choice3_table(const uint64_t state[4], int64_t n, int32_t trials, int32_t *out);

/* ---- slice binning: analyzer.py:49-52 -------------------------------- */
/* xyz: N rows of (x, y, z).  Window s is the closed interval
 * [z_lo[s], z_hi[s]]; windows are ascending and near-uniform (as
 * analyzer.py:240-243 builds them); a point may lie in up to two consecutive
 * windows (shared closed edge) or in none.  Step 1 counts: offsets[S+1] is the
 * exclusive prefix sum of the window populations (device memory).  Step 2
 * scatters x and y slice after slice, keeping the original order inside a
 * slice (what points[mask] gives); perm (nullable) receives the source row of
 * every output point.  x_out/y_out/perm must hold offsets[S] entries.
 * window_geom: three doubles in HOST memory, {z_lo[0], z_hi[0], z_lo[S-1]}
 * (the device arrays decide membership; these only locate the candidate
 * window).  At most 16384 windows per call.  The same workspace must be passed
 * to both steps. */
size_t pcfit_bin_workspace_bytes(int64_t n_points, int64_t n_slices);
int pcfit_bin_count_dev(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                        int64_t n_slices, const double *window_geom, int64_t *offsets, void *workspace,
                        size_t workspace_bytes, void *stream);
int pcfit_bin_scatter_dev(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                          int64_t n_slices, const double *window_geom, const int64_t *offsets, double *x_out,
                          double *y_out, int64_t *perm, void *workspace, size_t workspace_bytes, void *stream);

/* The same two steps for many windows and large scans, in two levels (the
 * host-level calls take this form by themselves from 8 M points and 257
 * windows on): step 1 sorts x, y and a 16-bit key of every member into groups
 * of 64 or 128 consecutive windows inside the workspace and counts; step 2
 * sorts every group into its windows.  Members, order and bytes are those of
 * pcfit_bin_count_dev / pcfit_bin_scatter_dev (analyzer.py:49-52); the output
 * leaves in runs of 32 to 64 points instead of point by point.  Nothing is
 * read back in between: the counts stay on the device.
 * max_entries: how many (point, group) entries the workspace holds; <= 0 for
 * 2 * n_points, which always suffices (a point on the edge shared by two
 * groups makes two entries, any other member one, a point in no window
 * none).  status (device, one int32): 0; 1 when the entries exceed
 * max_entries, 2 when one group has 2^32 entries or more - offsets are then
 * all zero and nothing is scattered.  want_perm must be the same in both
 * steps (and in the workspace size); perm may be NULL in step 2. */
size_t pcfit_bin2_workspace_bytes(int64_t n_points, int64_t n_slices, int want_perm, int64_t max_entries);
int pcfit_bin2_count_dev(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                         int64_t n_slices, const double *window_geom, int want_perm, int64_t max_entries,
                         int64_t *offsets, int32_t *status, void *workspace, size_t workspace_bytes, void *stream);
int pcfit_bin2_scatter_dev(int64_t n_points, int64_t n_slices, const double *window_geom, int want_perm,
                           int64_t max_entries, const int64_t *offsets, double *x_out, double *y_out, int64_t *perm,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ---- algebraic fit: fit_circle_pratt, circle_fitting.py:35-54 -------- */
/* cxcyr: S*3 (cx, cy, R).  status (nullable): 0, 1 (n < 3) or 4 (exactly
 * singular normal equations: the minimum-norm solution is returned, as the
 * reference's lstsq fallback does).  max_slice_len: largest slice population
 * (lets the launch keep a slice in registers); 0 if unknown. */
int pcfit_kasa_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                         int64_t max_slice_len, double *cxcyr, int32_t *status, void *stream);

/* ---- geometric refinement: lm_circle_geometric, circle_fitting.py:57-109 */
/* init, out: S*3.  iters (nullable): S*4 = {iterations the reference loop
 * runs, accepted steps, sweeps over the points that only built the normal
 * equations, sweeps that evaluated a candidate}.  max_slice_len:
 * largest slice population (lets the launch keep a slice in registers); pass 0
 * if unknown. */
int pcfit_lm_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                       int64_t max_slice_len, const double *init, double huber_delta, int32_t max_iter,
                       double lm_lambda0, double tol_update, double tol_cost, double *out, int32_t *iters,
                       void *stream);

/* Kasa start then LM in one launch, each slice read from HBM once
 * (circle_fitting.py:217-222 chains the two the same way).  kasa_out, iters,
 * status nullable. */
int pcfit_fit_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                        int64_t max_slice_len, double huber_delta, int32_t max_iter, double lm_lambda0,
                        double tol_update, double tol_cost, double *kasa_out, double *out, int32_t *iters,
                        int32_t *status, void *stream);

/* The same, followed by the residual statistics of pcfit_residuals_batch_dev
 * about the fitted circle (histogram, Rmax / Rmin and their indices), all in
 * one launch: 16 bytes of HBM traffic per point for the whole fit + residual
 * pass.  At most 4096 bins.  max_iter = 0 (the Kasa circle is the result: the
 * algebraic fit + residual statistics of analyzer.py:83-92 without the
 * refinement) runs a kernel of its own without the LM loop's code. */
int pcfit_fit_residuals_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                  int64_t max_slice_len, double huber_delta, int32_t max_iter, double lm_lambda0,
                                  double tol_update, double tol_cost, double *kasa_out, double *out, int32_t *iters,
                                  int32_t *status, int32_t n_bins, const double *edges, int64_t *hist,
                                  double *rmax_rmin, int64_t *argmax_argmin, void *stream);

/* ---- the same fits and statistics for LONG slices ------------------------
 * The entry points above give a slice to one workgroup, which fills the GPU
 * when there are many slices.  These cut every slice into segments of 8192
 * points (one workgroup each) and combine the partial results per slice in a
 * fixed order, so a scan of a few dozen windows of a million points each also
 * uses the whole chip.  Same arguments and results as their namesakes
 * (fit_circle_pratt circle_fitting.py:35-54, lm_circle_geometric :57-109, the
 * residual statistics analyzer.py:83-92 / gui.py:972-987, boundary_by_angle_max
 * boundary.py:7-42); sums are taken in another order, so fitted parameters agree
 * to rounding (~1e-15), integer results exactly.  total_points = offsets[n_slices].
 * workspace: pcfit_split_workspace_bytes(total_points, n_slices, bins) bytes of
 * device memory (bins = 0 unless pcfit_angle_max_split_dev is called).
 * pcfit_lm_split_dev / pcfit_fit_split_dev run at most max_iter + 1 rounds of
 * two launches.  The rounds are enqueued eight at a time and the calling thread
 * reads back, two batches late, how many slices are unfinished, so these two
 * calls return only when the rounds they enqueued have run (unlike the other
 * *_dev calls).  On a stream that is being captured nothing is read back and
 * all max_iter + 1 rounds are enqueued; finished slices cost next to nothing
 * in the later rounds. */
size_t pcfit_split_workspace_bytes(int64_t total_points, int64_t n_slices, int32_t n_bins);
int pcfit_kasa_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                         int64_t total_points, double *cxcyr, int32_t *status, void *workspace,
                         size_t workspace_bytes, void *stream);
int pcfit_lm_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                       int64_t total_points, const double *init, double huber_delta, int32_t max_iter,
                       double lm_lambda0, double tol_update, double tol_cost, double *out, int32_t *iters,
                       void *workspace, size_t workspace_bytes, void *stream);
/* Kasa start then LM; kasa_out, out and status are required here. */
int pcfit_fit_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                        int64_t total_points, double huber_delta, int32_t max_iter, double lm_lambda0,
                        double tol_update, double tol_cost, double *kasa_out, double *out, int32_t *iters,
                        int32_t *status, void *workspace, size_t workspace_bytes, void *stream);
int pcfit_residuals_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                              int64_t total_points, const double *cxcyr, double *theta, double *dr, double *radius,
                              int32_t n_bins, const double *edges, int64_t *hist, double *rmax_rmin,
                              int64_t *argmax_argmin, void *workspace, size_t workspace_bytes, void *stream);
int pcfit_angle_max_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                              int64_t total_points, const double *cxcyr, int32_t bins, int64_t *sel,
                              void *workspace, size_t workspace_bytes, void *stream);

/* ---- RANSAC: _ransac_circle, circle_fitting.py:112-143 --------------- */
/* triples: n_tables tables of trials*3 int32 (pcfit_choice3_table), device
 * memory; table_of_slice[s] selects the table built for that slice's n.
 * r_range: NULL or two doubles in HOST memory.  Outputs: model S*3, score S
 * (-1 when no trial produced a circle), best_trial S, mask one byte per point
 * (same CSR layout as x), status S (0, 1 or 3).  The earliest trial of
 * maximal score wins. */
size_t pcfit_ransac_workspace_bytes(int64_t n_slices);
int pcfit_ransac_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                           double inlier_tol, int32_t trials, double min_inliers_frac, const double *r_range,
                           const int32_t *triples, const int32_t *table_of_slice, double *model, int32_t *score,
                           int32_t *best_trial, uint8_t *mask, int32_t *status, void *workspace,
                           size_t workspace_bytes, void *stream);

/* pcfit_ransac_batch_dev without a workspace argument, for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * per-slice scratch (8 bytes per slice) lives in a fixed array inside the
 * library's device image, PCFIT_RANSAC_STATIC_MAX_SLICES entries = 512 KB on
 * every GPU the library runs on, held until the process ends; more slices than
 * that give PCFIT_ERR_UNSUPPORTED.  The call allocates and frees nothing.  All
 * calls on one GPU share that array: two calls that overlap in time on the same
 * GPU (different streams or threads) race; calls on one stream, or on different
 * GPUs, do not.  Same results as pcfit_ransac_batch_dev. */
#define PCFIT_RANSAC_STATIC_MAX_SLICES 65536
int pcfit_ransac_batch_static_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                  double inlier_tol, int32_t trials, double min_inliers_frac, const double *r_range,
                                  const int32_t *triples, const int32_t *table_of_slice, double *model,
                                  int32_t *score, int32_t *best_trial, uint8_t *mask, int32_t *status, void *stream);

/* ---- residual profile + histogram: gui.py:972-987, analyzer.py:83-92 -- */
/* Per point theta = (atan2 + 2pi) mod 2pi, dr = hypot - R and radius = hypot
 * (each output nullable); per slice numpy.histogram(dr, bins=H, range=(edges[0], edges[H]))
 * with the H+1 edges given (device memory; H may be 0), Rmax/Rmin and the first
 * slice-local index attaining each.  With all three per-point outputs NULL (and
 * at most 4096 bins) the statistics come from one streamed pass that takes the
 * correctly rounded hypot only of the points a quicker radius cannot decide:
 * the same counts, extremes and indices (slices of up to 2^31 - 1 points, as
 * for the fit kernels). */
int pcfit_residuals_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                              const double *cxcyr, double *theta, double *dr, double *radius, int32_t n_bins,
                              const double *edges, int64_t *hist, double *rmax_rmin, int64_t *argmax_argmin,
                              void *stream);

/* ---- boundary by angular bins: boundary_by_angle_max, boundary.py:7-42 - */
/* sel: S*bins; the slice-local index of the first point of largest radius in
 * each angular bin about (cx, cy) = cxcyr[3s], cxcyr[3s+1], or -1. */
int pcfit_angle_max_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                              const double *cxcyr, int32_t bins, int64_t *sel, void *stream);

/* ---- boundary and residual statistics about the same circle, one sweep ------------------------------ */
/* boundary_by_angle_max (boundary.py:7-42: sel as pcfit_angle_max_batch_dev) and the residual statistics of
 * pcfit_residuals_batch_dev without per-point outputs (gui.py:972-987, analyzer.py:83-92: hist, rmax_rmin,
 * argmax_argmin) for every slice, each slice read from HBM once instead of three times: both start from a
 * point's offsets from the centre and its distance.  The same indices, counts and extremes as the two calls.
 * 1 <= bins <= 4096, 0 <= n_bins <= 4096 (edges: n_bins + 1 doubles in device memory), slices of up to
 * 2^31 - 1 points. */
int pcfit_boundary_residuals_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                       const double *cxcyr, int32_t bins, int64_t *sel, int32_t n_bins,
                                       const double *edges, int64_t *hist, double *rmax_rmin,
                                       int64_t *argmax_argmin, void *stream);

/* ---- unique rows, rank check, hull: circle_fitting.py:14-19, 182-198 -- */
/* ux, uy: the slice's distinct points sorted by (x, y), written at the slice's
 * own CSR offset; n_unique[s] of them (all of them, whatever the slice's
 * length: slices beyond 8192 points are sorted through the workspace).  hull_idx (same CSR capacity): indices
 * into the slice's unique array of its strictly convex hull, counter-
 * clockwise, starting at index 0 or, when hull_start (nullable, S entries,
 * -1 = default) names a hull vertex, at that vertex.  n_hull[s] = 0 and
 * status 2 when the unique points have no 2-D variation; status 1 when fewer
 * than 3 points. */
/* In the host-pointer forms below (pcfit_unique_hull_batch, pcfit_hybrid_batch) a hull_start entry of
 * PCFIT_HULL_START_QHULL asks for the vertex scipy.spatial.ConvexHull(xy).vertices lists first
 * (circle_fitting.py:189-190), found by a replay of Qhull's construction on the device
 * (pcfit_qhull_start_batch_dev; the few slices that kernel hands back are replayed on host cores as in
 * pcfit_qhull_order_batch); the walk is then also reduced to Qhull's vertex set (a vertex between two
 * hull edges that Qhull merges - points exactly in line - is left out, n_hull goes down). */
#define PCFIT_HULL_START_QHULL (-2)

/* The order of the hull candidates of a whole call (the hull_order argument of pcfit_hybrid_batch and of the
 * scan forms).  RANSAC samples candidates by index (circle_fitting.py:126 on :191-192's array), so the order
 * decides which triples are drawn and, on slices with outliers on the hull, the inlier set and the fit.
 *   PCFIT_HULL_ORDER_QHULL      the order of circle_fitting.py:189-192's xy[hull.vertices], element for
 *                               element: candidates, inlier mask and fit are the reference's.  This is what
 *                               the Python layers of this package pass unless told otherwise.
 *   PCFIT_HULL_ORDER_CANONICAL  counter-clockwise from the lexicographically smallest vertex, every vertex
 *                               of the exact hull kept: no replay of Qhull (about a third faster on a
 *                               100 M-point scan); same vertex set and cyclic order, other sampled triples.
 * There is no default and no process-wide setting: every call states its order, so calls from several
 * threads (analyzer.py:203-207 enters the numerics from a thread pool) cannot change each other's results. */
#define PCFIT_HULL_ORDER_CANONICAL 0
#define PCFIT_HULL_ORDER_QHULL 1
size_t pcfit_unique_hull_workspace_bytes(int64_t total_points, int64_t n_slices);
int pcfit_unique_hull_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                int64_t max_slice_len, double *ux, double *uy, int64_t *n_unique,
                                int64_t *hull_idx, int64_t *n_hull, int32_t *status, const int64_t *hull_start,
                                void *workspace, size_t workspace_bytes, void *stream);

/* ---- ovality + k=2 Fourier amplitude: ovality_and_fourier, boundary.py:56-90 */
/* scal: S*5 = Rmax, Rmin, ovality_abs, ovality_pct, k2_amp (NaN for an empty
 * slice or a NaN circle).  dev_uniform (nullable): row s holds the deviation
 * profile resampled on K_s = max(1024, n_s) uniform angles, NaN from K_s to the
 * end of the row; rows are grid_len apart, grid_len >= max(1024, longest slice).
 * Slices of at most 8192 points (PCFIT_ERR_UNSUPPORTED beyond). */
int pcfit_ovality_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                            int64_t max_slice_len, const double *cxcyr, double *scal, double *dev_uniform,
                            int64_t grid_len, void *stream);
/* The same for slices of any length (the reference takes any: with
 * use_hull_vertices_only=False the candidates are all distinct points of the
 * slice, circle_fitting.py:196-198).  Slices of up to 8192 points run as above;
 * longer ones sort their angles through the workspace
 * (pcfit_ovality_workspace_bytes(total_points, n_slices) bytes of device memory,
 * not needed - may be NULL - when max_slice_len <= 8192).
 * total_points = offsets[n_slices]. */
size_t pcfit_ovality_workspace_bytes(int64_t total_points, int64_t n_slices);
int pcfit_ovality_long_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                           int64_t total_points, int64_t max_slice_len, const double *cxcyr, double *scal,
                           double *dev_uniform, int64_t grid_len, void *workspace, size_t workspace_bytes,
                           void *stream);

/* ---- host-pointer convenience forms (numpy in, numpy out) ------------ */
int pcfit_kasa_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices, double *cxcyr,
                     int32_t *status);
int pcfit_lm_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices, const double *init,
                   double huber_delta, int32_t max_iter, double lm_lambda0, double tol_update, double tol_cost,
                   double *out, int32_t *iters);
int pcfit_ransac_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                       double inlier_tol, int32_t trials, double min_inliers_frac, const double *r_range,
                       uint64_t seed, double *model, int32_t *score, int32_t *best_trial, uint8_t *mask,
                       int32_t *status);
int pcfit_residuals_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                          const double *cxcyr, double *theta, double *dr, double *radius, int32_t n_bins,
                          const double *edges, int64_t *hist, double *rmax_rmin, int64_t *argmax_argmin);
int pcfit_angle_max_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                          const double *cxcyr, int32_t bins, int64_t *sel);
int pcfit_boundary_residuals_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                   const double *cxcyr, int32_t bins, int64_t *sel, int32_t n_bins,
                                   const double *edges, int64_t *hist, double *rmax_rmin, int64_t *argmax_argmin);
/* offsets: S+1 out; x_out/y_out/perm: capacity entries (2*N always suffices);
 * *total receives offsets[S].  Returns PCFIT_ERR_INVALID if capacity is short. */
int pcfit_bin_slices(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t n_slices,
                     int64_t *offsets, double *x_out, double *y_out, int64_t *perm, int64_t capacity,
                     int64_t *total);
int pcfit_unique_hull_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices, double *ux,
                            double *uy, int64_t *n_unique, int64_t *hull_idx, int64_t *n_hull, int32_t *status,
                            const int64_t *hull_start);
int pcfit_ovality_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                        const double *cxcyr, double *scal, double *dev_uniform, int64_t grid_len);

/* The order of `ConvexHull(xy).vertices` (circle_fitting.py:189-190: the RANSAC candidates in the order the
 * sampled indices refer to).  scipy walks Qhull's facet list from its head, so the first vertex depends on
 * Qhull's incremental construction; this call replays that construction (libqhull 2020.2 under scipy's
 * options, 2-D, points in general position) on n_threads host cores (0 = one per core, at most 16) for
 * slices of distinct points laid out like the ux / uy / n_unique of pcfit_unique_hull_batch (n_unique NULL:
 * every slice is full).  start[s]: index of the first vertex in the slice's array, -1 when there is no hull.
 * Nullable: sure[s] = 0 when a decision fell inside Qhull's roundoff bands (nearly collinear hull points,
 * points on an edge of the hull).  Of Qhull's facet merging the replay performs the one case such data
 * needs - two facets of which one's centrum lies within the centrum radius of the other's line become one
 * facet, the vertex between them goes - and nothing else, so sure[s] = 0 marks slices on which more of the
 * merging could have played a part (none of those tried differed from scipy's list), not slices known to
 * differ; walk (capacity as ux) + n_walk[s]: all vertices in scipy's order.  No GPU work. */
int pcfit_qhull_order_batch(const double *ux, const double *uy, const int64_t *offsets, const int64_t *n_unique,
                            int64_t n_slices, int32_t n_threads, int64_t *start, uint8_t *sure, int64_t *walk,
                            int64_t *n_walk);

/* The same replay on the device, for slices that are there already (ux, uy, n_unique, n_hull as
 * pcfit_unique_hull_batch_dev left them; select NULL = every slice, else only where select[s] ==
 * PCFIT_HULL_START_QHULL): one workgroup per slice.  start[s] = the first vertex, -1 where there is no hull or
 * the slice was not selected, PCFIT_QHULL_START_DEFER where the kernel met anything but the plain course of
 * the construction (a decision inside Qhull's roundoff bands, a tie for the farthest point, a flat initial
 * simplex, more than 479 facets alive at once) - those slices are for pcfit_qhull_order_batch.
 * The points may also be the slices as they are (n_unique NULL: any order, duplicates allowed; start[s] then
 * names one of the copies of the vertex).
 * hull_x, hull_y, hull_idx (all three or none): the slices' hulls as pcfit_unique_hull_batch_dev left them in
 * ux, uy, hull_idx - vertex i of slice s at (hull_x, hull_y)[offsets[s] + hull_idx[offsets[s] + i]], i <
 * n_hull[s]; they need not index the array in ux, uy.  With them the kernel takes the four coordinate extremes
 * Qhull starts from among the hull's vertices instead of by a pass over the slice.
 * max_hull: the largest n_hull of the batch, or -1 (unknown): the kernel runs in the form with the fewest
 * facet slots that holds twice that many, so that more slices are in flight on a compute unit.
 * scratch: total_points * 48 bytes of device memory, aligned to 16 (the kernel keeps the points outside of the
 * growing hull as (x, y) pairs in it, so that every pass over them streams). */
#define PCFIT_QHULL_START_DEFER (-3)
int pcfit_qhull_start_batch_dev(const double *ux, const double *uy, const int64_t *offsets, const int64_t *n_unique,
                                const int64_t *n_hull, const int64_t *select, int64_t n_slices, int64_t total_points,
                                void *scratch, size_t scratch_bytes, int64_t *start, const double *hull_x,
                                const double *hull_y, const int64_t *hull_idx, int64_t max_hull, void *stream);

/* How many slices that kernel handed back since the last reset, by cause (out16[1..13]: flat simplex, simplex
 * orientation, first partition near a threshold, tie for the farthest point, horizon near the threshold, more
 * than 8 visible facets, new facets, facet slots, nearly collinear facets, partition near a threshold, pieces
 * of an outside set, vertex count, steps; out16[14], out16[15]: insertions made and points visited after the
 * first partition).  For tools and tests; synchronises the device.  The kernel counts from the first call of
 * this function in the process on (call it once with reset = 1 before the launches to be counted): launches of a
 * process that never calls it write nothing but their results. */
int pcfit_debug_qhull_reasons(int64_t *out16, int reset);

/* _has_enough_2d_variation (circle_fitting.py:14-19) per slice: ok[s] = 1 when
 * the second singular value of the centred points exceeds eps*(first + eps). */
int pcfit_rank_check_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices, double eps,
                           uint8_t *ok);
/* _circle_from_3pts (circle_fitting.py:22-32) for n_triples rows of
 * {x1,y1,x2,y2,x3,y3}: out n_triples*3 (cx, cy, R), ok 0 where |det| < 1e-12. */
int pcfit_circle3_batch(const double *triples, int64_t n_triples, double *out, uint8_t *ok);

/* ---- the whole hybrid fit for a batch of slices ----------------------- */
/* fit_circle_hybrid_at_z (circle_fitting.py:146-232) for S slices whose points
 * are already cut to their z window, followed by the per-slice metrics of
 * CylinderAnalyzer.process_slice (analyzer.py:83-95).  Host pointers.
 * In: r_range NULL or 2 doubles; use_hull = use_hull_vertices_only;
 *     hull_order PCFIT_HULL_ORDER_QHULL or PCFIT_HULL_ORDER_CANONICAL (above): the order of
 *     every slice's hull candidates when hull_start is NULL;
 *     hull_start NULL, or S entries that decide slice by slice instead of hull_order:
 *     -1 = from the lexicographic minimum, PCFIT_HULL_START_QHULL = Qhull's order,
 *     otherwise the index, in the slice's unique array, of the vertex to start from.
 * Out (capacity of every per-point array = offsets[S]):
 *     status S (0 ok / 1 / 2 / 3: the reference returns None), fit S*3,
 *     ux, uy + n_unique S        the sorted distinct points (fit_points_xy), all of them for
 *                                slices of any length; ux and uy may be NULL when the caller
 *                                does not want them: with hull candidates (and no hull_start)
 *                                the hull kernel then drops provably interior points of slices
 *                                longer than 1024 before sorting, and n_unique counts what it
 *                                kept; with use_hull = 0 the candidates are all distinct points
 *                                and n_unique counts them either way,
 *     cand_offsets S+1, cand_x, cand_y, mask     candidates_xy and inlier_mask,
 *     radius, theta              hypot and angle of every candidate about the fit,
 *     argmax_argmin S*2          candidate indices of the largest / smallest radius,
 *     ov_scal S*5, dev_uniform (nullable, rows grid_len apart)   ovality_and_fourier. */
int pcfit_hybrid_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                       double inlier_tol, int32_t max_trials, double min_inliers_frac, const double *r_range,
                       double huber_delta, int32_t use_hull, const int64_t *hull_start, int32_t hull_order,
                       uint64_t seed, int32_t *status, double *fit, double *ux, double *uy, int64_t *n_unique,
                       int64_t *cand_offsets, double *cand_x, double *cand_y, uint8_t *mask, double *radius,
                       double *theta, int64_t *argmax_argmin, double *ov_scal, double *dev_uniform,
                       int64_t grid_len);

/* A whole scan in one call: CylinderAnalyzer.analyze_with_progress /
 * analyze_file (analyzer.py:131-243) cut the scan into the windows
 * [z_lo[s], z_hi[s]] (analyzer.py:49-52) and run process_slice on each; here
 * xyz (n_points*3, row-major) goes to the device once, is binned there
 * (pcfit_bin_slices' rules: uniform windows, file order kept inside a window)
 * and the hybrid chain above runs on the device-resident slices.  Only
 * per-slice results and the candidates come back:
 *     offsets n_slices+1         CSR offsets of the windows (n_points_slice = their differences),
 *     the outputs of pcfit_hybrid_batch without ux / uy; the candidate arrays
 *     (cand_x, cand_y, mask, radius, theta) hold cand_capacity entries and the
 *     call fails with PCFIT_ERR_INVALID, naming the count, when more are needed.
 * hull_order as in pcfit_hybrid_batch (per-slice start vertices are not offered here). */
int pcfit_scan_hybrid_batch(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                            int64_t n_slices, double inlier_tol, int32_t max_trials, double min_inliers_frac,
                            const double *r_range, double huber_delta, int32_t use_hull, int32_t hull_order,
                            uint64_t seed, int64_t *offsets, int32_t *status, double *fit, int64_t *n_unique,
                            int64_t *cand_offsets, double *cand_x, double *cand_y, uint8_t *mask, double *radius,
                            double *theta, int64_t *argmax_argmin, double *ov_scal, double *dev_uniform,
                            int64_t grid_len, int64_t cand_capacity);

/* ---- the step before the path: XYZ text (host cores only, no GPU work) ---- */
/* load_txt_points (utils/data_io.py:8-27; the GUI's loader gui.py:105-117 is the
 * same rule without '//' and ';' comments) parsed by n_threads host threads
 * (0 = one per core, at most 16).  text holds the file's bytes.  Lines made only of
 * digits, signs, '.', 'e', 'E', commas and blanks are decided here (correctly
 * rounded, like float()); comment lines are dropped; every other non-blank line is
 * returned as a row with deferred[i] = 1 and {NaN, first byte, one past last byte}
 * in rows, for the caller to judge by the literal rule.  rows capacity*3,
 * deferred capacity, where capacity must be at least the number of lines
 * (pcfit_count_text_lines; PCFIT_ERR_INVALID otherwise, *n_rows = the need);
 * *n_rows receives the row count.  Rows keep file order. */
int pcfit_count_text_lines(const char *text, size_t n_bytes, int32_t n_threads, int64_t *n_lines);
int pcfit_parse_xyz_text(const char *text, size_t n_bytes, int32_t n_threads, double *rows, uint8_t *deferred,
                         int64_t capacity, int64_t *n_rows);

/* ---- a scan that stays on the device --------------------------------- */
/* The reference keeps one scan in memory and analyses it again and again (the
 * GUI's Analyze button with new parameters, visualize_slice: gui.py:31,1500).
 * pcfit_scan_upload copies xyz (n_points*3, row-major) to the current device
 * once and, if z_min_max is given, stores numpy's points[:,2].min() and .max()
 * there (analyzer.py:236: what calculate_z_centers needs; both NaN if any z is
 * NaN, or if the scan is empty); pcfit_scan_hybrid_resident is
 * pcfit_scan_hybrid_batch on that copy; pcfit_scan_free releases it (to the
 * block cache).  A handle belongs to the device it was made on.  Beside the
 * rows the handle keeps the z column by itself (8 bytes a point more): the
 * count passes of the binning read it instead of the rows (24 bytes a point);
 * PCFIT_SCAN_Z_COLUMN=0 in the environment, or too little memory, leaves it out. */
typedef struct pcfit_scan pcfit_scan;
int pcfit_scan_upload(const double *xyz, int64_t n_points, pcfit_scan **scan, double *z_min_max);
int pcfit_scan_free(pcfit_scan *scan);
/* pcfit_bin_slices on the resident copy (analyzer.py:49-52 for any windows: the GUI's
 * single-slice inspection, gui.py:1478-1501, reads one window of the loaded scan). */
int pcfit_scan_bin_slices(const pcfit_scan *scan, const double *z_lo, const double *z_hi, int64_t n_slices,
                          int64_t *offsets, double *x_out, double *y_out, int64_t *perm, int64_t capacity,
                          int64_t *total);
/* Members of every window of the resident scan without materialising them:
 * counts[s] = number of points with z_lo[s] <= z <= z_hi[s] (the n0 of
 * analyzer.py:49-54 for all windows at once; what balancing the windows over
 * several GPUs needs).  The count pass of the binning alone. */
int pcfit_scan_window_counts(const pcfit_scan *scan, const double *z_lo, const double *z_hi, int64_t n_slices,
                             int64_t *counts);
int pcfit_scan_hybrid_resident(const pcfit_scan *scan, const double *z_lo, const double *z_hi, int64_t n_slices,
                               double inlier_tol, int32_t max_trials, double min_inliers_frac, const double *r_range,
                               double huber_delta, int32_t use_hull, int32_t hull_order, uint64_t seed,
                               int64_t *offsets, int32_t *status, double *fit, int64_t *n_unique, int64_t *cand_offsets,
                               double *cand_x, double *cand_y, uint8_t *mask, double *radius, double *theta,
                               int64_t *argmax_argmin, double *ov_scal, double *dev_uniform, int64_t grid_len,
                               int64_t cand_capacity);

/* ---- one scan over several GPUs: what a rank needs of it (host cores + one upload) ---- */
/* A rank that analyses the windows [z_lo[a], z_hi[b]] of a scan needs only the rows with
 * z inside that range (analyzer.py:49-52 reads nothing else for those windows).  These
 * three calls let every rank find its windows and upload its rows without a pass of the
 * interpreter over the scan: numpy's points[:,2].min() / .max() (analyzer.py:236; both NaN
 * if any z is NaN), the window populations n0 of analyzer.py:49-54 for all windows (same
 * closed-interval rule as the device binning), and pcfit_scan_upload restricted to the rows
 * with z_lo <= z <= z_hi, which keep their file order (so every window of the range holds
 * the same points in the same order as in the whole scan).  n_threads: 0 = one per core,
 * at most 16. */
int pcfit_host_z_range(const double *xyz, int64_t n_points, int32_t n_threads, double *z_min_max);
int pcfit_host_window_counts(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                             int64_t n_slices, int32_t n_threads, int64_t *counts);
int pcfit_scan_upload_z_range(const double *xyz, int64_t n_points, double z_lo, double z_hi, pcfit_scan **scan,
                              int64_t *n_kept);

/* ---- the step after the path: CSV rows (host cores only) -------------------- */
/* The body of slice_points.csv as save_results writes it (utils/data_io.py:43-46:
 * to_csv(index=False, float_format="%.6f")): n_cols columns of n_rows doubles, one
 * column after the other in cols; is_int[c] != 0 prints column c as an integer;
 * NaN gives an empty field; fields joined by ',', rows ended by '\n'.  *length
 * receives the byte count; PCFIT_ERR_INVALID if capacity is short (call again
 * with *length bytes). */
int pcfit_format_csv_rows(const double *cols, int64_t n_rows, int32_t n_cols, const uint8_t *is_int,
                          int32_t n_threads, char *out, size_t capacity, size_t *length);

#ifdef __cplusplus
}
#endif
#endif /* PCFIT_H */
