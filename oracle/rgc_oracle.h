/*
 * rgc_oracle.h -- CPU restatement of the RGC-SLAM scan-to-map registration path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle (and the timed CPU
 * baseline) for the HIP path; the product library (librgc_hip.so) never links,
 * loads or calls anything in this directory.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED: the reference (ROBOT-WSC/RGC-SLAM @2024_10_08) ships no tests,
 * golden vectors or fixtures for this path and cannot be compiled in this image
 * (needs ROS1, PCL, Eigen, Ceres, Boost -- none installed, no network).  The
 * oracle is therefore pinned only by (i) an independent numpy/scipy restatement
 * (oracle/py_oracle.py -> tests/golden/), (ii) analytic known-answer cases and
 * (iii) recovery of known SE(3) motions.  See DESIGN.md.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/rgc_slam/).
 */
#ifndef RGC_ORACLE_H
#define RGC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* enum order follows include/fast_gicp/gicp/gicp_settings.hpp:8 */
enum { ORC_DIRECT27 = 0, ORC_DIRECT7 = 1, ORC_DIRECT1 = 2 };

typedef struct {
  double voxel_res;             /* RGC_odometer.cpp:308,1000  (1.0)                  */
  int    max_iterations;        /* RGC_odometer.cpp:1001      (25)                   */
  int    lm_max_iterations;     /* lsq_registration_impl.hpp:17 (10)                 */
  double rotation_eps;          /* lsq_registration_impl.hpp:12 (2e-3)               */
  double translation_eps;       /* RGC_odometer.cpp:1003      (1e-6)                 */
  double lm_init_lambda_factor; /* lsq_registration_impl.hpp:18 (1e-9)               */
  int    k_correspondences;     /* fast_gicp_impl.hpp:16      (20)                   */
  int    neighbor_method;       /* fast_vgicp_impl.hpp:23     (ORC_DIRECT1)          */
  int    num_threads;           /* RGC_odometer.cpp:1006      (14; 0 = omp max)      */
  int    regularization;        /* fast_gicp_impl.hpp:20      (ORC_REG_PLANE); enum order = gicp_settings.hpp:6  */
  int    voxel_mode;            /* fast_vgicp_impl.hpp:24     (ORC_VOXEL_ADDITIVE); enum order = gicp_settings.hpp:10 */
} orc_params;
enum { ORC_REG_NONE = 0, ORC_REG_MIN_EIG = 1, ORC_REG_NORMALIZED_MIN_EIG = 2, ORC_REG_PLANE = 3, ORC_REG_FROBENIUS = 4 };
enum { ORC_VOXEL_ADDITIVE = 0, ORC_VOXEL_ADDITIVE_WEIGHTED = 1, ORC_VOXEL_MULTIPLICATIVE = 2 };

void orc_default_params(orc_params* p);

/* ---- C1/C2: exact kNN + PLANE-regularised covariance (fast_gicp_impl.hpp:241-298) ---- */
/* pts: AoS floats, point i at pts + i*stride (x,y,z first).  idx_out: n*k ints sorted by
 * (float squared distance, index); d2_out: n*k floats (may be NULL). returns 0 / <0 error. */
int orc_knn(const float* pts, int n, int stride, int k, int* idx_out, float* d2_out, int num_threads);
/* cov9_out: n*9 doubles row-major 3x3 (the block<3,3>(0,0) of the reference's Matrix4d);
 * normal_out: n*3 unit eigenvector of the least eigenvalue (may be NULL). */
int orc_covariances(const float* pts, int n, int stride, int k, double* cov9_out, double* normal_out, int num_threads);
/* the same under any RegularizationMethod (fast_gicp_impl.hpp:262-293); orc_regularize: one neighbourhood's sample covariance S (row-major
 * 3x3) -> the regularised covariance */
int orc_covariances_m(const float* pts, int n, int stride, int k, int method, double* cov9_out, int num_threads);
void orc_regularize(const double S[9], int method, double cov9[9]);
/* covariance of ONE neighbourhood given explicit indices (for known-answer tests) */
void orc_cov_from_neighbors(const float* pts, int stride, const int* idx, int k, double cov9[9], double normal[3]);
/* symmetric 3x3 eigen decomposition (cyclic Jacobi); evals descending, evecs columns row-major */
void orc_eig3(const double A[9], double evals[3], double evecs[9]);

/* ---- C3: Gaussian voxel map (fast_vgicp_voxel.hpp:105-182) ---- */
typedef struct orc_voxelmap orc_voxelmap;
orc_voxelmap* orc_voxelmap_create(const float* pts, int n, int stride, const double* cov9, double res);
/* multiplicative != 0: MultiplicativeGaussianVoxel (fast_vgicp_voxel.hpp:76-99) instead of AdditiveGaussianVoxel (:105-122) */
orc_voxelmap* orc_voxelmap_create_m(const float* pts, int n, int stride, const double* cov9, double res, int multiplicative);
void orc_voxelmap_free(orc_voxelmap*);
int  orc_voxelmap_size(const orc_voxelmap*);
/* dump sorted by (cx,cy,cz): coords 3V ints, num V ints, mean 3V doubles, cov 9V doubles */
void orc_voxelmap_dump(const orc_voxelmap*, int* coords, int* num, double* mean, double* cov9);
/* voxel_coord (fast_vgicp_voxel.hpp:158-160) */
void orc_voxel_coord(const double x[3], double res, int c[3]);

/* ---- C4-C9: registration object mirroring FastVGICP as called at RGC_odometer.cpp:998-1011 ---- */
typedef struct orc_reg orc_reg;
orc_reg* orc_reg_create(const orc_params* p);
void orc_reg_free(orc_reg*);
int  orc_reg_set_target(orc_reg*, const float* pts, int n, int stride);   /* fast_vgicp_impl.hpp:56-63 */
int  orc_reg_set_source(orc_reg*, const float* pts, int n, int stride);   /* fast_gicp_impl.hpp:72-80  */
/* force covariance + voxel map computation now (otherwise lazily in align/linearize) */
int  orc_reg_prepare(orc_reg*);
/* fast_vgicp_impl.hpp:119-180.  T row-major 4x4 double.  H(36 row-major)/b(6) may be NULL. */
double orc_reg_linearize(orc_reg*, const double T[16], double* H, double* b);
/* fast_vgicp_impl.hpp:183-204 -- frozen correspondences + Mahalanobis of the last linearize */
double orc_reg_compute_error(orc_reg*, const double T[16]);
int  orc_reg_num_correspondences(const orc_reg*);

typedef struct {
  int    outer;        /* outer iteration index                       */
  int    inner;        /* number of LM tries in this outer iteration  */
  int    n_corr;
  double y0, yi, rho, lambda_before, lambda_after;
  int    accepted;     /* 1 = x updated, 0 = rho<0 && converged exit  */
  double x[16];        /* pose after this outer iteration             */
} orc_lm_trace;

/* lsq_registration_impl.hpp:53-172.  guess/final row-major 4x4 float.  trace: up to
 * max_trace entries (may be NULL).  returns number of outer iterations executed. */
int orc_reg_align(orc_reg*, const float guess[16], float final_T[16], double final_H[36],
                  int* converged, int* lm_failed, orc_lm_trace* trace, int max_trace);
int orc_reg_num_linearize(const orc_reg*);   /* counters since last align start */
int orc_reg_num_error(const orc_reg*);
/* pcl::Registration::getFitnessScore (RGC_odometer.cpp:1010) */
double orc_reg_fitness(orc_reg*, const float final_T[16]);
/* access the lazily computed per-point results */
const double* orc_reg_source_cov(orc_reg*);   /* n_s*9 */
const double* orc_reg_target_cov(orc_reg*);   /* n_t*9 */
const orc_voxelmap* orc_reg_voxelmap(orc_reg*);

/* ---- B3: pcl::VoxelGrid<PointXYZI>::filter restatement (RGC_odometer.cpp:976-991) ---- */
/* pts: n*4 floats (x,y,z,intensity). out: caller buffer n*4. returns number of output points,
 * or -1 if the leaf grid overflows int (PCL then returns the input unfiltered). */
int orc_voxelgrid_filter(const float* xyzi, int n, float leaf, float* out_xyzi);

/* ---- B2: vg_ICP::adjustDistortion (src/RGC_odometer.cpp:1441-1481), in place; quaternions are x,y,z,w ---- */
void orc_deskew(float* xyzi, int n, int stride, const double q_last_curr_xyzw[4], const double t_last_curr[3]);
/* ---- B9: vg_ICP::transformPointCloud (src/RGC_odometer.cpp:1495-1514); out: n*4 floats ---- */
void orc_transform_cloud(const float* xyzi, int n, int stride, const double q_xyzw[4], const double t[3], float* out_xyzi);

/* ---- A1-A8: ScanRegistration::laserCloudHandler (src/scanRegistration.cpp:89-730) ---- */
typedef struct { int n_scans; double min_range, max_range; int use_intensity; } orc_fe_params;
typedef struct {
  /* caller-allocated, capacity n (input size) unless noted */
  float* cloud;            /* n*4: ring-major full cloud, intensity = ring + 0.1*relTime (/velodyne_cloud_2) */
  int n_cloud;
  int ring_count[64], scan_start[64], scan_end[64];
  float *curvature, *curvature2, *inten_curvature;   /* n each */
  int *label, *inten_label, *picked, *ground_marked; /* n each */
  float *sharp, *flat, *inten;                       /* feat_cap*5: x,y,z,intensity,normal_x weight */
  int feat_cap, n_sharp, n_sharp_own, n_flat, n_inten;
  float* ground_pts; int ground_cap, n_ground;       /* ground_cap*4, pushed with duplicates in reference order */
  double groundparam[11]; int ground_valid;          /* ground_msg/groundparam field order */
} orc_fe_out;
void orc_fe_default_params(orc_fe_params* p);
int orc_frontend(const float* xyzi, int n, int stride, const orc_fe_params* prm, orc_fe_out* out);

/* ---- exact kNN of arbitrary query points (pcl::KdTreeFLANN::nearestKSearch restated) ---- */
int orc_knn_query(const float* pts, int n, int stride, const float* queries, int nq, int qstride, int k, int* idx_out, float* d2_out,
                  int num_threads);

/* ---- f1: scan-to-map FEATURE registration of the mapping node (src/RGC_mapping.cpp:1069-1358, src/lidarFactor.hpp:9-51,91-121) ---- */
typedef struct { int valid; int pad; double a[3], b[3], var; } orc_edge_factor;   /* LidarEdgeFactor(curr, point_a, point_b, var) */
typedef struct { int valid; int pad; double n[3], d, var; } orc_plane_factor;     /* LidarPlaneNormFactor(curr, norm, negative_OA_dot_norm, var) */
typedef struct {
  double initial_cost, final_cost, radius;
  int iterations, successful, n_edge_cur, n_edge_last, n_plane_cur, n_plane_last, pad;
} orc_mapreg_trace;
/* one Ground_DeltaFactor_goable (src/lidarFactor.hpp:352-403, RGC_mapping.cpp:1314-1340); quaternions x,y,z,w */
typedef struct {
  double last_v1[3], last_v2[3], last_norm[3], last_distance; /* g_last: vector_1, vector_2, vector_norm, distance */
  double cur_norm[3], cur_distance;                          /* g_cur */
  double q_history[4], last_q[4], last_t[3], p_var;          /* q_w_curr_f, q_w_last, t_w_last, ground_cov (0.2) */
} orc_mapreg_ground;
/* the IMU block of RGC_mapping.cpp:1285-1312: RelativeRFactor(delta_q_imu, imu_cov) on (q_last, q_cur) and PitchRollFactor(pitch, roll,
 * pr_var) on each pose (src/lidarFactor.hpp:174-226, 434-468) */
typedef struct { double delta_q[4], imu_cov, pitch_cur, roll_cur, pitch_last, roll_last, pr_var; } orc_mapreg_imu;
/* features: nf x 4 floats (x, y, z, normal_x = the per-feature weight of scanRegistration.cpp:501,554,609); q = x,y,z,w */
int orc_mapreg_associate_edges(const float* feat, int nf, const double q_xyzw[4], const double t[3], const float* map_xyz, int nmap,
                               int mstride, orc_edge_factor* out, int num_threads);
int orc_mapreg_associate_planes(const float* feat, int nf, const double q_xyzw[4], const double t[3], const float* map_xyz, int nmap,
                                int mstride, orc_plane_factor* out, int num_threads);
/* poses: q_cur[4] t_cur[3] q_last[4] t_last[3], in/out */
int orc_mapreg_solve(const float* corner_cur, const orc_edge_factor* e_cur, int n_ccur, const float* surf_cur, const orc_plane_factor* p_cur,
                     int n_scur, const float* corner_last, const orc_edge_factor* e_last, int n_clast, const float* surf_last,
                     const orc_plane_factor* p_last, int n_slast, const orc_mapreg_ground* ground_cur /* nullable */,
                     const orc_mapreg_ground* ground_last /* nullable */, const orc_mapreg_imu* imu /* nullable */, double poses[14],
                     int max_iterations, orc_mapreg_trace* trace);
/* returns 1 if the gate of RGC_mapping.cpp:1069 is not met (poses untouched), 0 on success */
int orc_mapreg_optimize(const float* corner_cur, int n_ccur, const float* surf_cur, int n_scur, const float* corner_last, int n_clast,
                        const float* surf_last, int n_slast, const float* corner_map, int n_cmap, const float* surf_map, int n_smap,
                        int mstride, const orc_mapreg_ground* ground_cur, const orc_mapreg_ground* ground_last, const orc_mapreg_imu* imu,
                        double poses[14], orc_mapreg_trace trace[2], int num_threads);

/* ---- f4: loop-closure ICP = pcl::IterativeClosestPoint as used at src/RGC_mapping.cpp:2050-2069 (restated, see rgc_oracle.c) ---- */
enum { ORC_ICP_NOT_CONVERGED = 0, ORC_ICP_ITERATIONS = 1, ORC_ICP_TRANSFORM = 2, ORC_ICP_ABS_MSE = 3, ORC_ICP_REL_MSE = 4, ORC_ICP_NO_CORRESPONDENCES = 5 };
typedef struct { int max_iterations; int pad; double max_corr_dist, transformation_eps, fitness_eps; } orc_icp_params;
typedef struct { int iterations, converged, state, n_correspondences; double fitness; } orc_icp_result;
/* source -> target, identity guess; final_T row-major 4x4 float (icp.getFinalTransformation()); result.fitness = getFitnessScore() */
int orc_icp_align(const float* src, int ns, int sstride, const float* tgt, int nt, int tstride, const orc_icp_params* prm, float final_T[16],
                  orc_icp_result* res, int num_threads);
void orc_rigid_from_sums(double n, const double sp[3], const double sq[3], const double spq[9], double R[9], double t[3]);

/* ---- C7 helpers ---- */
void orc_so3_exp(const double omega[3], double q_wxyz[4]);                 /* so3/so3.hpp:58-77 */
int  orc_is_converged(const double delta[16], double rot_eps, double trans_eps); /* lsq_registration_impl.hpp:82-91 */

/* ---- point transforms ---- */
/* pcl::transformPointCloud restated in fp32 (lsq_registration_impl.hpp:78) */
void orc_transform_f32(const float* pts, int n, int stride, const float T[16], float* out_xyz /* n*3 */);

#ifdef __cplusplus
}
#endif
#endif
