/* lio_c.h — C-ABI of the MI355X-native sliding-window LiDAR-inertial estimator.
 *
 * The reference (hyye/lio-mapping) has no FFI: estimator_node links the C++ classes directly.
 * This header flattens the public surface of those classes to POD so a host such as
 * estimator_node can bind it.  It is implemented twice, with identical symbols:
 *   - lio-mapping_amd/csrc  -> liblio_hip.so     (the product: HIP kernels for gfx950 + C++ host)
 *   - oracle/               -> liblio_oracle.so  (CPU restatement; test infrastructure only)
 *
 * Conventions: every function returns int (LIO_OK or a negative code) unless it returns a handle,
 * a count or void; nothing throws; no caller pointer is retained past the call; one caller thread
 * per handle (like the reference's estimator thread B, src/estimator_node.cc:153).
 * Points are float[4] = x,y,z,intensity (pcl::PointXYZI's useful 16 bytes).
 * Poses are double[7] = px,py,pz,qx,qy,qz,qw (Estimator.cc:2445-2452); speed-bias double[9] =
 * v,ba,bg (Estimator.cc:2454-2464); rotation matrices are row-major double[9].
 */
#ifndef LIO_C_H_
#define LIO_C_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIO_OK 0
#define LIO_ERR_ARG (-1)      /* null / out-of-range argument */
#define LIO_ERR_STATE (-2)    /* call order violated (e.g. solve before the window is full) */
#define LIO_ERR_DEVICE (-3)   /* HIP runtime failure or no GPU: the product never falls back to CPU */
#define LIO_ERR_CAPACITY (-4) /* a fixed-capacity device buffer would overflow */

/* Twist<float> (include/utils/Twist.h:39-97): rotation quaternion x,y,z,w + translation */
typedef struct {
  float q[4];
  float p[3];
} lio_transform_f;

/* Which implementation is behind the symbols: "hip-gfx950" or "oracle-cpu". */
const char *lio_backend(void);

/* ------------------------------------------------------------------------------------------------
 * PointProcessor (include/point_processor/PointProcessor.h:127-165; §8a a1-a5)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  double scan_period;          /* 0.1   PointProcessor.h:106 */
  int num_scan_subregions;     /* 8     :107 */
  int num_curvature_regions;   /* 5     :108 */
  float surf_curv_th;          /* 0.1f  :109 */
  int max_corner_sharp;        /* 2     :110 */
  int max_corner_less_sharp;   /* 20    :111 */
  int max_surf_flat;           /* 4     :112 — the product requires max_corner_less_sharp + max_surf_flat <= 64 and
                                  max_corner_sharp <= max_corner_less_sharp (lio_pp_create returns NULL otherwise) */
  float less_flat_filter_size; /* 0.2f  :113 */
  int infer_start_ori;         /* 0     :119 — lio_pp_process only: replace a jumping start azimuth by the one the last ten
                                  sweeps predict (PointProcessor.cc:348-387); the ten-sweep history lives in the lio_pp */
  double rad_diff;             /* 0.2   :117 — the jump (rad) that triggers the replacement */
} lio_pp_config;

typedef struct lio_pp lio_pp;

enum {
  LIO_PP_RINGS = 0,       /* laser_scans concatenated: intensity = ring + rel_time */
  LIO_PP_SHARP = 1,       /* corner_points_sharp_ */
  LIO_PP_LESS_SHARP = 2,  /* corner_points_less_sharp_ */
  LIO_PP_FLAT = 3,        /* surface_points_flat_ */
  LIO_PP_LESS_FLAT = 4    /* surface_points_less_flat_ (per-ring voxel-downsampled) */
};

void lio_pp_default_config(lio_pp_config *cfg);
/* PointProcessor(float lower, float upper, int rings, bool uneven) — PointProcessor.cc:75.  `uneven` is not a
 * constructor argument here: it only selects which PointToRing overload runs, i.e. which of the two process calls
 * below the caller uses (PointProcessor.cc:185-190). */
lio_pp *lio_pp_create(float lower_deg, float upper_deg, int rings, const lio_pp_config *cfg_or_null);
/* Why lio_pp_create would refuse these arguments: LIO_OK, LIO_ERR_ARG (out-of-range value), or LIO_ERR_CAPACITY — the product's
 * k_ring_pick keeps a subregion's picks one per lane of a wave, so max_corner_less_sharp + max_surf_flat <= 64 (the reference,
 * PointProcessor.cc:685-732, accepts any quota; its defaults are 20 + 4).  The oracle has no such limit.  No device work. */
int lio_pp_check_config(float lower_deg, float upper_deg, int rings, const lio_pp_config *config_or_null);
void lio_pp_destroy(lio_pp *);
/* SetInputCloud + PointToRing + ExtractFeaturePoints (PointProcessor.cc:96-100, test_point_processor.cc:103-106) */
int lio_pp_process(lio_pp *, const float *xyzi, size_t n);
/* lio_pp_process in two halves, for hosts that keep several sweeps in flight (one handle per sweep in flight, each with its
 * own stream): _async enqueues upload + all kernels + the copy of the counts and returns; _wait blocks until they are done.
 * xyzi must stay untouched between the two calls; every accessor below (and the next _async on the handle) waits first.
 * (The oracle processes inside _async; its _wait is a no-op.) */
int lio_pp_process_async(lio_pp *, const float *xyzi, size_t n);
int lio_pp_wait(lio_pp *);
/* B sweeps (B sensors of a fleet node, or the sweeps of a log) through B handles in one call.  Handles created with the same
 * arguments share ONE launch chain: one upload per sweep, then every kernel of the chain (ring split, PrepareRing / PrepareSubregion /
 * picks, per-ring VoxelGrid, packing — PointProcessor.cc:207-783) runs ONCE over all B sweeps (the sweep is a grid dimension; a single
 * HDL-64 sweep fills 64 of the 256 compute units in the pick stage), and one copy brings all counts back.  Every handle then answers
 * the accessors below for ITS sweep, bit for bit what lio_pp_process gives (the same kernels: one sweep is the B = 1 case); with
 * infer_start_ori a sweep's start azimuth goes through its own handle's ten-sweep history.  The results live in storage the handles of
 * the call share until a handle's next process call: read them from one thread at a time; a later batch call that reuses the storage for
 * OTHER handles (it is kept by the first handle of a call) invalidates them — the accessors of a handle left out then return
 * LIO_ERR_STATE (counts 0) instead of another sweep's data.  Handles that differ in their arguments (or
 * B = 1) run lio_pp_process_async on every handle, then lio_pp_wait on every handle; the first failing code is returned after all
 * handles have been waited for.  A handle may appear once. */
int lio_pp_process_batch(lio_pp *const *handles, const float *const *xyzi, const size_t *n, int n_sweeps);
/* lio_pp_process_batch with the sweeps already in DEVICE memory (a decoder or a replay buffer on the GPU; d_xyzi[k]: n[k] x 4 floats in
 * HBM of the handles' device): no transfer over PCIe, the sweeps are copied device to device into the chain's segments.  (The oracle
 * has no device: it reads the pointers as host memory.) */
int lio_pp_process_batch_device(lio_pp *const *handles, const float *const *d_xyzi, const size_t *n, int n_sweeps);
/* The same with the PointIR overload of PointToRing (uneven = true, sensor_type 320 of processor_node.cc:73;
 * PointProcessor.cc:428-536): the ring of each point comes from its `ring` field (points whose ring is outside
 * [0, rings) are dropped) and rel_time = scan_period * (unwrapped azimuth - start_ori) / (end_ori - start_ori). */
int lio_pp_process_rings(lio_pp *, const float *xyzi, const uint16_t *ring, size_t n);
/* lio_pp_process_batch with that overload: B sweeps of ring-field sensors (ring[k]: one ring per point of sweep k) through one launch
 * chain; same rules as lio_pp_process_batch. */
int lio_pp_process_rings_batch(lio_pp *const *handles, const float *const *xyzi, const uint16_t *const *ring, const size_t *n, int n_sweeps);
/* start_ori_ the last process call used (after the inference when infer_start_ori is set); NaN before the first call */
float lio_pp_start_ori(const lio_pp *);
size_t lio_pp_count(const lio_pp *, int which);
int lio_pp_get_cloud(const lio_pp *, int which, float *xyzi_out);
/* parity object of §8a a4: ordered (ring, in-ring index) per picked class; which in {SHARP,LESS_SHARP,FLAT} */
int lio_pp_get_indices(const lio_pp *, int which, int32_t *ring_out, int32_t *idx_out);
/* scan_ranges as rings+1 exclusive offsets into the LIO_PP_RINGS cloud (PointProcessor.cc:193-201) */
int lio_pp_get_ring_offsets(const lio_pp *, int32_t *offsets_out);
/* per-point curvature of the LIO_PP_RINGS cloud (0 outside [5, n-5) of a ring) and final pick mask */
int lio_pp_get_curvature(const lio_pp *, float *curv_out, int32_t *mask_out);
/* the intensity channel of the reference's public intensity_scans, concatenated in ring order (one float per LIO_PP_RINGS point):
 * int(input intensity) + rel_time (PointProcessor.cc:413, :524; the coordinates are those of the LIO_PP_RINGS cloud) */
int lio_pp_get_ring_intensity(const lio_pp *, float *intensity_out);

/* ------------------------------------------------------------------------------------------------
 * PointOdometry (include/point_processor/PointOdometry.h:102-147; §8a a6-a7): LOAM scan-to-scan step
 * ---------------------------------------------------------------------------------------------- */
typedef struct lio_odom lio_odom;
/* PointOdometry(float scan_period = 0.1, int io_ratio = 2, size_t max_iter = 25) + the no_deskew param
 * (PointOdometry.cc:66-73,109) */
lio_odom *lio_odom_create(float scan_period, int io_ratio, int num_max_iterations, int no_deskew);
void lio_odom_destroy(lio_odom *);
/* PointOdometry::Process (PointOdometry.cc:294-683) on the four feature clouds of one sweep (the topics of
 * PointProcessor::PublishResults; intensity = ring + rel_time).  The first call only stores the clouds
 * (system_inited_, :302-310).  Outputs (any may be null): transform_sum_ (accumulated odometry),
 * transform_es_ (sweep end -> start), iterations run, number of selected correspondences in the last round. */
int lio_odom_process(lio_odom *, const float *sharp_xyzi, size_t n_sharp, const float *less_sharp_xyzi, size_t n_less_sharp,
                     const float *flat_xyzi, size_t n_flat, const float *less_flat_xyzi, size_t n_less_flat,
                     lio_transform_f *transform_sum_out, lio_transform_f *transform_es_out, int *iterations_out,
                     int *num_selected_out);
/* The iterations of the last lio_odom_process, one record each (SURVEY.md §8(d) config 2 compares them one by one):
 * trace[k] = transform_es_ at the end of iteration k (PointOdometry.cc:337-652; an iteration with fewer than 10 selected
 * rows leaves it unchanged, :535).  Returns the number of iterations run (<= num_max_iterations; 0 for the first call, a
 * disabled odometry or too small previous clouds) and copies min(that, capacity) records.  kz_out (may be null): the number
 * of leading update components zeroed by the degeneracy test of iteration 0 (:584-615, eigenvalues of AtA below 10;
 * the reference's mat_P is diag(0..0,1..1)); 0 = not degenerate. */
int lio_odom_get_iteration_trace(const lio_odom *, lio_transform_f *trace_or_null, int capacity, int *kz_out);
/* /enable_odom service (PointOdometry.h:126-131): 0 turns the step into a packer (A.18) */
int lio_odom_enable(lio_odom *, int on);
/* last_corner_cloud_ / last_surf_cloud_ after the swap (:667-668), i.e. the TransformToEnd outputs.
 * which: 0 corner, 1 surf.  Returns the count; copies when out is non-null. */
size_t lio_odom_get_last_cloud(const lio_odom *, int which, float *xyzi_or_null);

/* /compact_data wire format between PointOdometry and the estimator (§8a a8): a cloud of xyzi points where
 * point[0] = (tx,ty,tz | 0), point[1] = (qx,qy,qz | qw) of transform_sum_, point[2] = (n_corner, n_surf, n_full | qw),
 * followed by the corner, surf and full clouds (encode: PointOdometry.cc:732-764; decode: PointMapping.cc:171-238).
 * encode writes 3+nc+ns+nf points into out and returns that count.  decode validates the header (>= 4 points and
 * 3+nc+ns+nf == n_points, else LIO_ERR_ARG) and returns the three sizes; the clouds start at point 3, 3+nc, 3+nc+ns. */
size_t lio_compact_encode(const lio_transform_f *transform_sum, const float *corner_xyzi, size_t n_corner, const float *surf_xyzi,
                          size_t n_surf, const float *full_xyzi, size_t n_full, float *out_xyzi);
int lio_compact_decode(const float *data_xyzi, size_t n_points, lio_transform_f *transform_sum_out, size_t *n_corner, size_t *n_surf,
                       size_t *n_full);

/* ------------------------------------------------------------------------------------------------
 * PointMapping — LOAM scan-to-map step and the 21x21x11 cube map of 50 m cubes (§8a a26, §8f 2).
 * Reference: src/point_processor/PointMapping.cc:325-753 (OptimizeTransformTobeMapped), :755-763,
 * :765-1110 (Process), :1112-1208 (UpdateMapDatabase); include/point_processor/PointMapping.h:112-253.
 * The estimator runs it for every sweep until the IMU is initialised (Estimator.cc:823-826).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float corner_filter_size;   /* 0.2  down_size_filter_corner_ (PointMapping.cc:122; Estimator.cc:189) */
  float surf_filter_size;     /* 0.4  down_size_filter_surf_   (:123; Estimator.cc:190)               */
  float min_match_sq_dis;     /* 1.0  PointMapping.h:245 */
  float min_plane_dis;        /* 0.2  :246               */
  int num_max_iterations;     /* 10   :171               */
  /* MapBuilder (src/map_builder/MapBuilder.cc, include/map_builder/MapBuilder.h:52-70; §8f 4): the same stage driven
   * as MapBuilder::ProcessMap — first call adopts transform_sum, Transform4DAssociateToMap (:55-75) keeps only the yaw
   * of the increment, OptimizeMap (:624-1014) solves with the rotation Jacobian in the map frame weighted
   * diag(5e-3, 5e-3, 1) and a left-multiplied update, every skip_count-th call optimises, the map is always updated. */
  int map_builder;            /* 0 = PointMapping::Process, 1 = MapBuilder::ProcessMap */
  int enable_4d;              /* 1    MapBuilder.h:66 */
  int skip_count;             /* 2    MapBuilder.h:67 */
} lio_map_config;

typedef struct lio_map lio_map;

enum {
  LIO_MAP_CORNER_STACK_DS = 0,  /* laser_cloud_corner_stack_downsampled_ (sensor frame) */
  LIO_MAP_SURF_STACK_DS = 1,    /* laser_cloud_surf_stack_downsampled_                  */
  LIO_MAP_CORNER_FROM_MAP = 2,  /* laser_cloud_corner_from_map_ of the last Process     */
  LIO_MAP_SURF_FROM_MAP = 3     /* laser_cloud_surf_from_map_                           */
};

void lio_map_default_config(lio_map_config *cfg);
/* PointMapping(float scan_period, size_t num_max_iterations) (PointMapping.cc:65-126) */
lio_map *lio_map_create(const lio_map_config *cfg_or_null);
void lio_map_destroy(lio_map *);
/* Handlers + PointMapping::Process (:765-1110) for one sweep: corner_last / surf_last are the odometry's
 * last_corner_cloud_/last_surf_cloud_, transform_sum the accumulated odometry.  Outputs (any may be null):
 * transform_aft_mapped_, iterations run, rows selected in the last round. */
int lio_map_process(lio_map *, const float *corner_last_xyzi, size_t n_corner, const float *surf_last_xyzi, size_t n_surf,
                    const lio_transform_f *transform_sum, lio_transform_f *transform_aft_mapped_out, int *iterations_out,
                    int *num_selected_out);
/* is_degenerate of the last Process's OptimizeTransformTobeMapped (PointMapping.cc:650-680; MapBuilder::OptimizeMap,
 * MapBuilder.cc:930-960): returns 1 / 0 (< 0 on error); kz_out (may be null) = leading eigenvalues of AtA below 100 in
 * round 0 = leading components of every round's update that were zeroed. */
int lio_map_get_degeneracy(const lio_map *, int *kz_out);
/* SetInitFlag (:299-301): once set, Process neither applies the odometry increment nor updates the map */
int lio_map_set_init_flag(lio_map *, int imu_inited);
/* transform_tobe_mapped_ accessors (the estimator overwrites it after init, Estimator.cc:796-797) */
int lio_map_set_transform_tobe_mapped(lio_map *, const lio_transform_f *);
int lio_map_get_transform_tobe_mapped(const lio_map *, lio_transform_f *out);
/* UpdateMapDatabase (:1112-1208): add the down-sampled stacks (sensor frame) at `transform`, then VoxelGrid every
 * cube of valid_idx (indices relative to cube_center, re-based on the current centre as :1171-1186 does). */
int lio_map_update_map_database(lio_map *, const float *corner_ds_xyzi, size_t n_corner, const float *surf_ds_xyzi, size_t n_surf,
                                const uint32_t *valid_idx, size_t n_valid, const lio_transform_f *transform,
                                const int cube_center[3]);
/* clouds of the last Process; returns the count, copies when out is non-null */
size_t lio_map_get_cloud(const lio_map *, int which, float *xyzi_or_null);
/* one cube of laser_cloud_corner_array_ (cls 0) / laser_cloud_surf_array_ (cls 1); cube_idx = ToIndex(i,j,k) */
size_t lio_map_get_cube(const lio_map *, int cls, uint32_t cube_idx, float *xyzi_or_null);
/* laser_cloud_cen_{length,width,height}_ and laser_cloud_valid_idx_ (<= 125 entries); returns the valid count */
size_t lio_map_get_cube_state(const lio_map *, int cube_center_out[3], uint32_t *valid_idx_or_null);
/* score_point_coeff_ (:725-750): descending score; point = p_ori (xyzi), coeff = abs_coeff (4 floats). Returns the count. */
size_t lio_map_get_score_point_coeff(const lio_map *, float *score_or_null, float *point_xyzi_or_null, float *coeff_or_null);

/* ------------------------------------------------------------------------------------------------
 * Batched keyframe refinement (BASELINE.json configs[4]).  No reference
 * counterpart as an API: the unit of work is the reference's scan-to-map Gauss-Newton loop —
 * MapBuilder::OptimizeMap (MapBuilder.cc:624-1014; 4-DoF, when cfg.map_builder && cfg.enable_4d) or
 * PointMapping::OptimizeTransformTobeMapped (PointMapping.cc:325-753; 6-DoF otherwise) — run once per keyframe
 * against that keyframe's local map.  Keyframes are independent; a handle refines all of its keyframes together.
 * Sharding over GPUs = sharding the keyframe list (one handle per rank); the only exchange is a gather of poses.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lio_kf_batch lio_kf_batch;
lio_kf_batch *lio_kf_batch_create(const lio_map_config *cfg_or_null);
void lio_kf_batch_destroy(lio_kf_batch *);
/* A local map = laser_cloud_corner_from_map_ / laser_cloud_surf_from_map_ (map frame).  Returns its index (>= 0) or
 * a negative error code.  The buffers are copied. */
int lio_kf_batch_add_map(lio_kf_batch *, const float *corner_xyzi, size_t n_corner, const float *surf_xyzi, size_t n_surf);
/* A keyframe = its down-sampled feature stacks in the lidar frame (laser_cloud_{corner,surf}_stack_downsampled_), the
 * local map it refines against, and its initial transform_tobe_mapped_.  Returns its index (>= 0) or an error code. */
int lio_kf_batch_add_keyframe(lio_kf_batch *, int map_index, const float *corner_xyzi, size_t n_corner, const float *surf_xyzi,
                              size_t n_surf, const lio_transform_f *T_init);
int lio_kf_batch_clear_keyframes(lio_kf_batch *);
/* Refines every keyframe from its T_init (repeatable).  Outputs are arrays of n_keyframes entries, each optional:
 * the refined transform, the iterations run (0 when the map is too small, :327-329) and the rows of the last round.
 * device_ms_or_null (HIP library only; 0 from the oracle) = device time of the round loop. */
int lio_kf_batch_refine(lio_kf_batch *, lio_transform_f *T_out_or_null, int32_t *iterations_or_null, int32_t *rows_or_null,
                        double *device_ms_or_null);
/* kz of every keyframe's last refine (see lio_map_get_degeneracy); n_keyframes entries.  LIO_ERR_STATE before the first refine. */
int lio_kf_batch_get_degeneracy(const lio_kf_batch *, int32_t *kz_out);
size_t lio_kf_batch_size(const lio_kf_batch *);

/* ------------------------------------------------------------------------------------------------
 * Stateless building blocks (third-party semantics restated)
 * ---------------------------------------------------------------------------------------------- */
/* pcl::VoxelGrid<PointXYZI> (B.1): centroids in ascending voxel index; out capacity n points.   */
int lio_voxel_grid(const float *xyzi, size_t n, float leaf, float *xyzi_out, size_t *n_out);
/* Measurement hook (SURVEY.md 8(d) ii): the same filter on a cloud already resident in HBM, `reps` times back to back on one
 * stream, timed with HIP events; avg_ms_out = one filter (keys, sort, run heads, centroids, the count's way back to the host).
 * The oracle returns LIO_ERR_DEVICE. */
int lio_bench_voxel_grid(const float *xyzi, size_t n, float leaf, int reps, double *avg_ms_out, size_t *n_out_or_null);
/* pcl::KdTreeFLANN::nearestKSearch (B.2): exact K-NN, ascending squared distance, index tiebreak.
 * idx_out / sqd_out are m*k.  The product restricts the search to radius_sq (entries beyond it
 * come back as idx -1 / sqd +inf); pass radius_sq <= 0 for an unbounded search (oracle only).  */
int lio_knn(const float *map_xyzi, size_t n_map, const float *query_xyzi, size_t m, int k,
            float radius_sq, int32_t *idx_out, float *sqd_out);
/* Estimator::CalculateFeatures, surf branch (Estimator.cc:1014-1097) for one stack against one
 * filtered local map.  Outputs per stack point: valid flag, coeff[4] (s*pa,s*pb,s*pc,s*pd), score. */
int lio_calculate_features(const float *map_xyzi, size_t n_map, const float *stack_xyzi, size_t m,
                           const lio_transform_f *local_transform, float min_match_sq_dis,
                           float min_plane_dis, uint8_t *valid_out, float *coeff_out, float *score_out);

/* ------------------------------------------------------------------------------------------------
 * IntegrationBase (include/imu_processor/IntegrationBase.h:77-357; §8a a9-a10)
 * ---------------------------------------------------------------------------------------------- */
typedef struct lio_pim lio_pim;
lio_pim *lio_pim_create(const double acc0[3], const double gyr0[3], const double ba[3],
                        const double bg[3], double acc_n, double gyr_n, double acc_w, double gyr_w,
                        double g_norm);
void lio_pim_destroy(lio_pim *);
int lio_pim_push_back(lio_pim *, double dt, const double acc[3], const double gyr[3]);
int lio_pim_repropagate(lio_pim *, const double ba[3], const double bg[3]);
/* any output pointer may be null; dq = x,y,z,w; jacobian / covariance 15x15 row-major */
int lio_pim_get(const lio_pim *, double *sum_dt, double *delta_p, double *delta_q, double *delta_v,
                double *jacobian, double *covariance);
int lio_pim_evaluate(const lio_pim *, const double *pose_i, const double *sb_i,
                     const double *pose_j, const double *sb_j, double *residual15);

/* ------------------------------------------------------------------------------------------------
 * ImuInitializer (src/imu_processor/ImuInitializer.cc:49-436; include/imu_processor/ImuInitializer.h:73-91) — host
 * math on <= window_size+1 frames (§8f 3).  `laser_transforms[i]` / `pims[i]` are all_laser_transforms_[i]'s
 * transform and pre_integration (pims[0] is never read for its motion, may be NULL).
 * ---------------------------------------------------------------------------------------------- */
/* EstimateExtrinsicRotation (:331-397): writes transform_lb->q; returns 1 when the second-smallest singular value
 * exceeds 0.25 (calibration accepted), 0 when not, < 0 on error. */
int lio_imu_estimate_extrinsic_rotation(size_t n, const lio_transform_f *laser_transforms, lio_pim *const *pims,
                                        lio_transform_f *transform_lb);
/* Initialization (:399-436) = EstimateGyroBias + ApproximateGravity + RefineGravityAccBias.  Bgs (3n doubles) is
 * updated in place, the pims are re-propagated with the new gyro bias (:86-89), Vs_out gets 3n doubles, g_out the
 * refined gravity in the laser world frame, R_WI_out (row-major 3x3) the inertial->laser-world rotation.
 * Returns 1 on success, 0 when the gravity estimate is rejected (:175) or n < 6, < 0 on error. */
int lio_imu_initialization(size_t n, const lio_transform_f *laser_transforms, lio_pim *const *pims,
                           const lio_transform_f *transform_lb, double *Vs_out, double *Bgs_inout, double g_out[3],
                           double R_WI_out[9]);

/* ------------------------------------------------------------------------------------------------
 * Factors (ceres::SizedCostFunction::Evaluate restated; §8a a11, a15, a24).  Jacobians are
 * row-major in the ambient (7/9 column) layout exactly as the reference writes them; null = skip.
 * ---------------------------------------------------------------------------------------------- */
/* ImuFactor::Evaluate (include/factor/ImuFactor.h:53-168): <15,7,9,7,9> */
int lio_factor_imu(const lio_pim *, const double *pose_i, const double *sb_i, const double *pose_j,
                   const double *sb_j, double *residual15, double *j_pose_i, double *j_sb_i,
                   double *j_pose_j, double *j_sb_j);
/* PivotPointPlaneFactor::Evaluate (src/factor/PivotPointPlaneFactor.cc:43-137): <1,7,7,7> */
int lio_factor_pivot_point_plane(const double point[3], const double coeff[4],
                                 const double *pose_pivot, const double *pose_i,
                                 const double *pose_ex, double *residual1, double *j_pivot,
                                 double *j_i, double *j_ex);
/* PriorFactor::Evaluate (src/factor/PriorFactor.cc:35-67): <6,7>; rot0 = x,y,z,w */
int lio_factor_prior(const double pos0[3], const double rot0[4], const double *pose,
                     double *residual6, double *j_pose);
/* PoseLocalParameterization::Plus (src/factor/PoseLocalParameterization.cc:35-50) */
int lio_pose_plus(const double *pose, const double *delta6, double *pose_out);

/* ------------------------------------------------------------------------------------------------
 * Estimator (include/imu_processor/Estimator.h:110-299; §8a a12-a27, §8b)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int window_size;            /* 15   Estimator.h:78  */
  int opt_window_size;        /* 5    :79             */
  float corner_filter_size;   /* 0.2  :83             */
  float surf_filter_size;     /* 0.4  :84             */
  float min_match_sq_dis;     /* 1.0  :87             */
  float min_plane_dis;        /* 0.2  :88             */
  lio_transform_f transform_lb; /* :89 */
  int opt_extrinsic;          /* :91 */
  int imu_factor;             /* :97 */
  int point_distance_factor;  /* :98 */
  int prior_factor;           /* :99 */
  int marginalization_factor; /* :100 */
  int enable_deskew;          /* :103 */
  int cutoff_deskew;          /* :104 */
  int keep_features;          /* :105 */
  double acc_n, gyr_n, acc_w, gyr_w, g_norm; /* IntegrationBaseConfig, IntegrationBase.h:64-70 */
  int max_num_iterations;     /* ceres options, Estimator.cc:1916 (10) */
  double max_solver_time;     /* Estimator.cc:1921 (0.10 s); <= 0 disables the cap (parity runs) */
  int extrinsic_stage;        /* estimate_extrinsic / extrinsic_stage_ (Estimator.h:81,174): 0 = fixed (SetParameterBlockConstant),
                                 1 = refine in the window, 2 = calibrate the rotation first (EstimateExtrinsicRotation) */
  int init_window_factor;     /* 3    Estimator.h:80: only every n-th frame enters the window until the IMU is initialised */
  /* ---- execution switches of the product (no reference counterpart; results are the same up to the documented tolerances, the
   * oracle ignores them).  0 = the shipped default.  An environment variable (named at each field), when set, overrides the
   * field at lio_est_create — for A/B runs of a host that cannot be rebuilt. */
  int device_solve;           /* 1: the handle solves as a batch of ONE window (lio_est_batch below: trust-region loop and marginalization on
                                 the device, the prior stays there between solves); LIO_DEVICE_SOLVE=1 or LIO_DEVICE_MARG=1 in the
                                 environment select it too.  Ignores max_solver_time */
  int inline_marg;            /* 1: marginalization inside lio_est_solve_optimization instead of the worker thread (LIO_ASYNC_MARG=0) */
  int stream_sync;            /* 1: hipStreamSynchronize + D2H copies instead of completion words in host memory (LIO_HOST_SIGNAL=0) */
  int moments_form;           /* 0: by launch size, 1: fp64 MFMA form, 2: structured fp64 VALU form (LIO_MOMENTS=mfma|valu) */
  int resident_moments;       /* 0: by default rule (on), 1: on, 2: off, 3: its partition with launches only — the lidar moments of a solve
                                 come from ONE resident kernel that waits for each linearisation point on a doorbell in host memory
                                 (LIO_RESIDENT_MOMENTS=0|1).  "On" is a permission: a solve takes the resident form while it is the only
                                 solve in flight in the process and its factor slots fit 256 co-resident blocks, and a launch pair
                                 per linearisation otherwise — over the SAME partition of the factor slots, so the moments do not
                                 depend on which of the two ran (3 pins the launch pairs: what a refused solve gets).  2 is round 2's
                                 launch pair with its own partition: the same sums in a different order, last-ulp differences */
} lio_est_config;

/* Named after the reference's TicToc stages so CPU/GPU tables line up. */
typedef struct {
  int iterations;             /* trust-region iterations taken (successful + unsuccessful) */
  int successful_steps;
  int termination;            /* 0 no-convergence(max it), 1 param tol, 2 function tol, 3 gradient tol, 4 time, 5 failure */
  int n_lidar_residuals;
  int n_local_map;            /* filtered local map points */
  int laser_odom_iterations;  /* CalculateLaserOdom rounds on the newest frame */
  int turn_off;               /* Estimator.cc:1655,1938-1942 */
  int convergence_flag;       /* Estimator.cc:1960-1962 */
  int marginalized;           /* 1 if a new prior was produced */
  double cost_pim_before, cost_ppp_before, cost_marg_before; /* Estimator.cc:1924-1954 */
  double initial_cost, final_cost;
  double cost_trace[32];      /* cost after iteration k (k = 0 initial) */
  double ms_build_map;        /* "t_build_map cost"  Estimator.cc:1531 */
  double ms_features;         /* sum of "feature cost" :1643 */
  double ms_prepare;          /* "prepare for ceres" :1907 (minus the two above) */
  double ms_opt;              /* "t_opt" :1993 */
  double ms_marg;             /* "whole marginalization costs" :2247 */
  double ms_total;            /* "tic_toc_opt" :2436 */
  int laser_odom_kz;          /* CalculateLaserOdom's degeneracy test (Estimator.cc:1308-1339): leading eigenvalues of the newest
                                 frame's 6x6 AtA below 100 in round 0 = leading update components zeroed in every round; 0 = none */
} lio_solve_report;

typedef struct lio_est lio_est;

void lio_est_default_config(lio_est_config *cfg);
lio_est *lio_est_create(const lio_est_config *cfg);
void lio_est_destroy(lio_est *);

/* Estimator::ProcessImu (Estimator.cc:338-427) */
int lio_est_process_imu(lio_est *, double dt, const double acc[3], const double gyr[3], double stamp);
/* The IMU loop of Estimator::ProcessEstimation (Estimator.cc:2700-2726): n samples of one laser interval in one call;
 * acc / gyr are n x 3, row-major.  Identical to n lio_est_process_imu calls. */
int lio_est_process_imu_batch(lio_est *, size_t n, const double *dt, const double *acc, const double *gyr, const double *stamp);
/* Estimator::ProcessLaserOdom (Estimator.cc:430-774).  Until the IMU is initialised (NOT_INITED, :490-618): every
 * init_window_factor-th frame is pushed (the clouds are then PointMapping's down-sampled stacks, :474-481); once
 * window_size+1 frames are held, EstimateExtrinsicRotation / RunInitialization (:858-958) are tried and on success
 * the first SolveOptimization + SlideWindow run.  INITED branch (:620-774): push the frame, (deskew), VoxelGrid the
 * surf/corner clouds, SolveOptimization, SlideWindow; the clouds are the implicit inputs the reference keeps in
 * laser_cloud_{surf,corner}_last_ (Estimator.cc:467-487).  lio_est_get_stage tells which of these happened. */
int lio_est_process_laser_odom(lio_est *, const lio_transform_f *transform_in, const float *surf_xyzi,
                               size_t n_surf, const float *corner_xyzi, size_t n_corner, double stamp,
                               lio_solve_report *report_or_null);
/* Estimator::ProcessCompactData (Estimator.cc:776-856): decode one /compact_data message (lio_compact_encode layout),
 * run the PointMapping base (scan-to-map + cube map) until the IMU is initialised or, afterwards, predict
 * transform_tobe_mapped_ from the IMU-propagated body motion (:780-803), then ProcessLaserOdom with
 * transform_aft_mapped_.  transform_to_init_out (may be null) receives that transform.  After initialisation the
 * caller switches the scan-to-scan odometry to its packer mode (lio_odom_enable(.., 0); the reference does it
 * through the /enable_odom service, :549-558).  Not reproduced: the post-init map-database refresh (:703-708),
 * which only feeds the published surround map. */
int lio_est_process_compact(lio_est *, const float *compact_xyzi, size_t n_points, double stamp,
                            lio_transform_f *transform_to_init_out, lio_solve_report *report_or_null);
/* stage_flag_ (0 NOT_INITED, 1 INITED), cir_buf_count_, extrinsic_stage_, what the last ProcessLaserOdom did
 * (0 frame skipped by init_window_factor, 1 window filling, 2 initialisation tried and failed, 3 initialised + first
 * solve, 4 regular solve), R_WI_ (row-major 3x3) and g_vec_.  Any output may be null. */
int lio_est_get_stage(const lio_est *, int *stage_flag, int *cir_buf_count, int *extrinsic_stage, int *last_event,
                      double *R_WI_or_null, double *g_vec_or_null);
/* First half of ProcessLaserOdom only (Estimator.cc:441-488,620-693): push the pre-integration, deskew,
 * VoxelGrid and push the clouds — everything up to, not including, SolveOptimization.
 * lio_est_process_laser_odom == lio_est_push_frame + lio_est_solve_optimization + lio_est_slide_window. */
int lio_est_push_frame(lio_est *, const lio_transform_f *transform_in, const float *surf_xyzi, size_t n_surf,
                       const float *corner_xyzi, size_t n_corner, double stamp);
/* Estimator::SolveOptimization (Estimator.cc:1648-2438) */
int lio_est_solve_optimization(lio_est *, lio_solve_report *report_or_null);
/* Estimator::SlideWindow (Estimator.cc:2570-2666) */
int lio_est_slide_window(lio_est *);
/* The product defers the marginalization of a solve (host-only work whose result, the prior, is first read by the next
 * solve) to a worker thread; every reader joins it first, so results are those of the synchronous sequence.  This call
 * waits for all deferred work of the handle (no reference counterpart; a no-op in the oracle).  LIO_ASYNC_MARG=0 in the
 * environment keeps the marginalization inside lio_est_solve_optimization. */
int lio_est_sync(lio_est *);

/* ---- test hooks (no reference counterpart): inject / read a window ---- */
/* n_frames must be window_size+1.  Marks the estimator INITED with a full window. */
int lio_est_set_window(lio_est *, int n_frames, const double *Ps, const double *Rs, const double *Vs,
                       const double *Bas, const double *Bgs, const double g_vec[3]);
int lio_est_get_window(const lio_est *, int n_frames, double *Ps, double *Rs, double *Vs, double *Bas,
                       double *Bgs, lio_transform_f *transform_lb_out);
/* surf_stack_[frame] (already voxel-filtered, lidar frame); also sets size_surf_stack_[frame] */
int lio_est_set_surf_stack(lio_est *, int frame, const float *xyzi, size_t n);
size_t lio_est_get_surf_stack(const lio_est *, int frame, float *xyzi_or_null);
/* pre_integrations_[frame] rebuilt from raw samples (frame >= 1) */
int lio_est_set_preintegration(lio_est *, int frame, const double acc0[3], const double gyr0[3],
                               const double ba[3], const double bg[3], const double *dt,
                               const double *acc, const double *gyr, size_t n_samples);
/* acc_last_/gyr_last_ + tmp_pre_integration_ restart, as after a ProcessLaserOdom push */
int lio_est_begin_frame(lio_est *, const double acc_last[3], const double gyr_last[3]);

/* Estimator::BuildLocalMap alone (Estimator.cc:1361-1646): local map + per-frame features */
int lio_est_build_local_map(lio_est *);
size_t lio_est_get_local_map(const lio_est *, float *xyzi_or_null);
size_t lio_est_get_features(const lio_est *, int frame, double *point3_or_null,
                            double *coeff4_or_null, double *score_or_null);
/* newest-frame transform after CalculateLaserOdom (Estimator.cc:1242-1359) */
int lio_est_get_laser_odom_transform(const lio_est *, lio_transform_f *out);

/* Marginalization prior (MarginalizationInfo after Marginalize, MarginalizationFactor.cc:185-311),
 * in the canonical kept order pose1,sb1,pose2..poseWo,ex.  Returns n (0 = none).
 * JtJ = linearized_jacobians^T linearized_jacobians (n*n), Jtr = linearized_jacobians^T
 * linearized_residuals (n), x0 = kept parameter blocks concatenated in ambient layout.           */
int lio_est_get_prior(const lio_est *, double *JtJ_or_null, double *Jtr_or_null, double *x0_or_null,
                      int *x0_len_or_null);

/* Test hooks (no reference counterpart, like lio_est_set_window): read / overwrite the NUMBERS of the current prior —
 * linearized_jacobians (n*n, row-major), linearized_residuals (n), keep_block_data (x0, ambient layout) — and the extrinsic
 * T_lb.  A chain of ProcessLaserOdom calls amplifies a 1e-8 difference of its inputs to 1e-3 m within a few steps through
 * the prior (measured on the CPU oracle against itself, tests/golden/README.md), so a step-by-step parity chain has to
 * hand BOTH implementations the same states, the same prior and the same extrinsic before every step.
 * get: returns n (0 = no prior).  set: the estimator must already hold a prior of the same n and x0 length (the block
 * structure is not transferable), else LIO_ERR_STATE. */
int lio_est_get_prior_factor(const lio_est *, double *lin_jac_or_null, double *lin_res_or_null, double *x0_or_null,
                             int *x0_len_or_null);
int lio_est_set_prior_factor(lio_est *, int n, const double *lin_jac, const double *lin_res, const double *x0, int x0_len);
int lio_est_set_extrinsic(lio_est *, const lio_transform_f *T_lb);

/* Test hook: x = A^-1 b for a symmetric positive definite A (n*n row-major, n <= 128) by the dense factorisation the solver
 * uses for its D x D system (Estimator.cc:1911 DENSE_SCHUR; the product: blocked L D L^T in LDS with fp64-MFMA trailing
 * updates, csrc/solve_step.h).  LIO_ERR_STATE when A is not positive definite. */
int lio_dense_spd_solve(const double *A, const double *b, int n, double *x_out);

/* Test hook: the dense tail of MarginalizationInfo::Marginalize (MarginalizationFactor.cc:271-302) on a caller-supplied
 * system.  A: (m + n)^2 row-major, b: m + n, the first m parameters are marginalised (m <= 15, n <= 80).  Out: lin_jac n x n
 * row-major = diag(sqrt(s_k > 1e-8 ? s_k : 0)) V^T, lin_res n = diag(1 / sqrt(s_k) or 0) V^T bs, evals n (ascending s_k of the
 * Schur complement).  The product runs it on the device (csrc/marg_kernels.hip: Jacobi eigensolver in LDS, Schur complement on
 * the fp64 matrix cores) — the path LIO_DEVICE_MARG=1 switches the estimator to. */
int lio_marginalize_schur(const double *A, const double *b, int m, int n, double *lin_jac, double *lin_res, double *evals);

/* In-memory snapshot / restore of the whole estimator state (bench + parity loops). */
int lio_est_snapshot(lio_est *);
int lio_est_restore(lio_est *);
/* Measurement hook: dst's snapshot becomes a copy of src's (same window sizes; the clouds are copied into buffers of dst's own), so
 * that B windows holding the same data at distinct addresses exist without replaying the sequence B times.  lio_est_restore(dst)
 * then puts dst into that state.  LIO_ERR_STATE when src has no snapshot or the window sizes differ. */
int lio_est_copy_snapshot(lio_est *dst, lio_est *src);
/* Measurement hook: `steps` times lio_est_restore + lio_est_solve_optimization, in one call (what bench.py times: a caller that
 * is compiled code, like estimator_node, pays no interpreter between two solves).  report_or_null receives the last solve's report.
 * Stops at the first failing step and returns its code. */
int lio_est_solve_restored(lio_est *, int steps, lio_solve_report *report_or_null);

/* ------------------------------------------------------------------------------------------------
 * Batched windows (no reference counterpart: the reference solves one window per process).
 * B independent estimators — different vehicles, logs, or the shards of an offline re-optimisation — solve together: every stage
 * of Estimator::SolveOptimization (Estimator.cc:1648-2438: BuildLocalMap :1361-1646, CalculateFeatures :970-1097,
 * CalculateLaserOdom :1242-1359, ceres::Solve :1909-1990, MarginalizationInfo::Marginalize MarginalizationFactor.cc:185-311)
 * is ONE launch over all windows, the trust-region loop (ImuFactor.h:53-168 and the prior included) and the Schur-complement
 * marginalization run on the device, one workgroup per window.  A window gives the same bits alone (a batch of one) and inside
 * any batch.  Windows the device loop does not take (convergence_flag_ still changing the problem, factor sharding, a local map
 * beyond the batched filter's key range) are solved by the single-window path inside the same call.  Every window must be
 * initialised: lio_est_batch_solve checks all of them first and returns LIO_ERR_STATE without solving any otherwise.
 * lio_est_batch_create ADOPTS the handles: their device work moves to the batch's stream; they stay usable one at a time from the
 * thread that drives the batch (push frames, slide, snapshot / restore, getters) and should outlive the batch — lio_est_destroy of
 * an adopted handle first dissolves its batch (the other members are released; the batch handle stays valid for
 * lio_est_batch_destroy only and every other call on it returns LIO_ERR_STATE).  NULL on bad arguments: no window, more than
 * 65535, a null handle, a handle twice, a handle that already belongs to a batch (lio_est_batch_destroy releases its members),
 * windows created on different devices. */
typedef struct lio_est_batch lio_est_batch;
lio_est_batch *lio_est_batch_create(lio_est *const *windows, int n_windows);
void lio_est_batch_destroy(lio_est_batch *);
int lio_est_batch_size(const lio_est_batch *);
/* SolveOptimization of every window; reports_or_null: n_windows reports (the ms_* fields of a window solved on the device are the
 * batch's stage times: ms_build_map = filter chain, ms_features = K-NN grids + features + newest-frame rounds, ms_opt = the
 * trust-region loop, ms_marg = write-back + marginalization enqueue).  LIO_ERR_STATE when a window is not initialised. */
int lio_est_batch_solve(lio_est_batch *, lio_solve_report *reports_or_null);
/* Measurement hook (what bench.py times): `steps` x (lio_est_restore of every window + lio_est_batch_solve), then a wait for the
 * last marginalizations. */
int lio_est_batch_solve_restored(lio_est_batch *, int steps, lio_solve_report *reports_or_null);
/* waits for everything the batch has enqueued (the marginalizations of the last solve run behind its return) */
int lio_est_batch_sync(lio_est_batch *);
/* host wall clock of the last lio_est_batch_solve, ms: [0] descriptors, [1] filter chain, [2] grids + features + rounds, [3] problem
 * packing (inside [2]), [4] trust-region loop, [5] write-back + marginalization enqueue, [6] single-window fallbacks, [7] total,
 * [8] windows solved on the device, [9] newest-frame rounds launched; DEVICE time of the stages (HIP events on the batch's stream;
 * the call waits for the stream): [10] filter chain, [11] K-NN grids, [12] features of the older frames, [13] newest-frame rounds
 * (without [16]), [14] trust-region loop, [15] marginalization, [16] the stream's wait for the PREVIOUS solve's marginalization
 * (it runs on a stream of its own beside this solve's first stages and is joined in front of the problems' upload); with the
 * option "time_kernels" (measurement runs: HIP events around every launch of the trust-region loop, on the stream it runs on)
 * [17..19] summed duration (ms) and [20..22] number of the last solve's launches of the aux row, the moments pass and the step
 * kernel, else 0; [23] reserved.  out: 24 doubles. */
int lio_est_batch_get_clock(const lio_est_batch *, double *out24);
/* Execution choices of a batch that its results do not depend on (bit for bit: tests/test_gpu_batch_scale.py), by name; value 0
 * (occupancy: -1) = chosen by the size of the launch, the default.  "lanes_per_query" 1 | 2 | 4 | 8 (search kernels of
 * CalculateFeatures / CalculateLaserOdom), "occupancy" 0 | 6 | 8 waves per SIMD of their one-lane-per-query forms, "loop_groups"
 * 1 .. 4 launch chains of the trust-region loop side by side, "aux_threads" 64 | 128 | 256 threads per block of the IMU / prior
 * row, "aux_stream" 0 | 1, "finish_threads" 1 .. 8 host threads of the write-back, "parts" 1 | 2 (a batch of at least 96 windows is solved
 * as two halves side by side from two host threads — one half's host phases and latency-bound stages fill with the other's kernels; the
 * clock then gives the longer half's host times and the SUM of the halves' device times), "time_kernels" 0 | 1 (lio_est_batch_get_clock).  The environment variables LIO_BW_LPQ,
 * LIO_BW_OCC, LIO_BW_GROUPS, LIO_BW_AUX_THREADS, LIO_BW_AUX_STREAM, LIO_BW_FINISH_THREADS set a new batch's defaults (read once
 * at lio_est_batch_create).  LIO_ERR_ARG: unknown name or value.  The oracle accepts and ignores them. */
int lio_est_batch_set_option(lio_est_batch *, const char *name, int value);
/* Test hook: the segmented stable radix sort of the batched BuildLocalMap (csrc/seg_sort.h; it orders a window's points by PCL's voxel
 * index, Estimator.cc:1518-1519, and by K-NN cell, :1544-1545): `passes` passes of `bits` (1 .. 9) bits from bit 0 over
 * (key, value) pairs inside n_segments ranges [seg_off[k], seg_off[k] + seg_n[k]) of arrays of n_total elements; values_or_null:
 * the values are the elements' positions.  Elements outside every segment are copied through.  LIO_ERR_ARG on bad arguments. */
int lio_seg_sort_pairs(const unsigned *keys, const unsigned *values_or_null, size_t n_total, const int *seg_off, const int *seg_n, int n_segments, int bits, int passes,
                       unsigned *keys_out, unsigned *values_out);
/* Test hook: one 64-bit digest per window of what a stage of the LAST lio_est_batch_solve left on the device — stage 0 the filtered
 * local map (Estimator.cc:1518-1519), 1 the K-NN grid (:1544-1545; points of a cell as a multiset), 2 the feature flags and
 * 3 the plane coefficients of CalculateFeatures / CalculateLaserOdom (:970-1097, :1242-1359), 4 the newest frame's Gauss-Newton state,
 * 5 the trust-region loop's final state (:1909-1990), 6 the final normal-equation moments, 7 the Jacobi scaling fixed at the first
 * linearisation, 8 the scaled Hessian at the accepted point, 9 the prior the
 * marginalization left on the device (MarginalizationFactor.cc:185-311).  Equal inputs must give equal digests
 * whatever the batch size (tests/test_gpu_batch_scale.py).  Waits for the batch.  The oracle returns zeros. */
int lio_est_batch_stage_digest(lio_est_batch *, int stage, unsigned long long *out_n_windows);

/* Multi-GPU factor sharding (the reference's own 4-thread split of ThreadsConstructA,
 * MarginalizationFactor.cc:245-269, extended across ranks): rank r of `world` evaluates only its contiguous share of
 * every frame's lidar factors; `allreduce` (in-place SUM over ranks of `count` doubles, returns 0) is called once per
 * linearisation on the per-shard normal-equation moments; every rank then takes the same trust-region step, so the
 * replicas stay in lockstep without a broadcast.  world = 1 or a null callback switches sharding off.
 * While sharding is on, max_solver_time is ignored (a per-rank wall clock must not decide how many collectives a rank
 * issues): the solve stops on max_num_iterations and the function / parameter / gradient tolerances only. */
typedef int (*lio_allreduce_fn)(double *inout, int count, void *user);
int lio_est_set_factor_sharding(lio_est *, int rank, int world, lio_allreduce_fn allreduce, void *user);

/* The same exchange done INSIDE the library with RCCL over xGMI: one communicator per process (= per GPU).  Rank 0 calls
 * lio_rccl_unique_id and hands the bytes to the other ranks by any side channel; every rank then calls lio_rccl_init on the
 * device it drives (ncclCommInitRank).  With a communicator set, the per-shard moments stay in HBM: the fold kernel writes
 * them to a device buffer, ncclAllReduce(ncclDouble, Wo x 260, SUM) runs on the estimator's stream, and only the reduced
 * moments cross PCIe.  rank / world come from the communicator; null switches back to the unsharded / callback form.
 * (The oracle is a CPU library: its lio_rccl_* return LIO_ERR_DEVICE / null.) */
#define LIO_RCCL_ID_BYTES 128
typedef struct lio_rccl lio_rccl;
int lio_rccl_unique_id(unsigned char id[LIO_RCCL_ID_BYTES]);
lio_rccl *lio_rccl_init(const unsigned char id[LIO_RCCL_ID_BYTES], int rank, int world);
void lio_rccl_destroy(lio_rccl *);
int lio_rccl_rank(const lio_rccl *);    /* ncclCommUserRank of the communicator (-1: null handle or RCCL error) */
int lio_rccl_world(const lio_rccl *);   /* ncclCommCount of the communicator (0: null handle or RCCL error) — what RCCL itself counts */
int lio_est_set_factor_sharding_rccl(lio_est *, lio_rccl *comm_or_null);
/* Measurement hook (collective: every rank of `comm` calls it with the same arguments): `reps` in-place SUM all-reduces of
 * `count` doubles from a device buffer, back to back on one stream; avg_us_out = mean microseconds per all-reduce by HIP
 * events on this rank.  bench.py reports it next to the factor-sharded solve rate (count = opt_window_size x 260). */
int lio_rccl_bench_all_reduce(lio_rccl *comm, int count, int reps, double *avg_us_out);
/* lio_kf_batch_refine followed by an all-gather of the results over `comm`: every rank receives `slots_per_rank` records of
 * 9 floats (q x,y,z,w; p x,y,z; iterations; rows) from every rank, rank r's records at [r * slots_per_rank, ...), records
 * beyond a rank's own keyframe count zero.  packed_all: world * slots_per_rank * 9 floats (host). */
int lio_kf_batch_refine_gather(lio_kf_batch *, lio_rccl *comm, int slots_per_rank, float *packed_all, double *device_ms_or_null);

/* Per-kernel timing with HIP events on the estimator's own stream (bench.py's roofline block).
 * Names: "features" (batched CalculateFeatures), "odom_features", "odom_rows", "odom_update",
 * "moments" (lidar normal-equation moments, MFMA), "voxel", "knn_grid", "concat"; and "moments_resident": the passes of the
 * resident moments kernel (lio_est_config.resident_moments), which one launch per solve serves — timed on the device's wall
 * clock from the doorbell seen to the sums posted, counted since the handle was created, independent of `on`; and
 * "moments_resident_launch": the dispatch-to-exit span of that kernel's launches, bracketed by HIP events while `on` = -1
 * (under `on` >= 1 every pass is a separate launch so that events can bracket it, and the resident kernel is not used).
 * `on` = 0 stops, 1 times every launch, N > 1 times every N-th launch of each kind (an event pair between two
 * kernels costs a few microseconds of dispatch overlap; sampling keeps the timed region honest).
 * get returns the number of launches TIMED since timing was enabled (0 for an unknown name or
 * for the oracle), total_ms their summed duration, algorithmic_bytes the summed SURVEY.md §8d bytes. */
int lio_est_enable_kernel_timing(lio_est *, int on);
/* Batched roofline probe (SURVEY.md §8d ii): the lidar factors of the CURRENT window (features as left by the last
 * BuildLocalMap / SolveOptimization) are replicated n_windows times at distinct addresses and evaluated by ONE launch
 * of the moments kernel (frame descriptors in device memory); `reps` launches are timed with HIP events on the
 * estimator's stream.  avg_ms_out = mean duration of one launch (moments + reduce), algorithmic_bytes_out = 60 B x
 * residuals x n_windows.  LIO_ERR_STATE when no features exist (and always for the CPU oracle). */
int lio_est_bench_batched_moments(lio_est *, int n_windows, int reps, double *avg_ms_out, double *algorithmic_bytes_out);
int lio_est_get_kernel_timing(lio_est *, const char *name, double *total_ms, double *algorithmic_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LIO_C_H_ */
