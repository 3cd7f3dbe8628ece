/*
 * kt_oracle_host.c -- CPU restatement of the reference's host-side hot-path logic
 * (SURVEY.md 8a rows a7, a10, a16): ICPOdometry / RGBDOdometry Gauss-Newton loops and the
 * KintinuousTracker::processFrame state machine, with the Eigen / OpenCV calls restated in plain C.
 * TEST INFRASTRUCTURE ONLY (see kt_oracle.h).  PARITY UNPINNED for this file: Eigen / OpenCV / PCL are neither vendored with the
 * reference nor installed, so the host logic is pinned by known-answer tests only (the device kernels are pinned
 * against the reference's own sources, see kt_oracle.h).
 * Paths cited are relative to /root/reference/src/.
 *
 * Third-party arithmetic restated here (not vendored in the reference; versions README.md:14-31):
 *   Eigen 3.2.x  Matrix3f::inverse() (cofactor form), Matrix<double,6,6>::ldlt().solve() (pivoted LDL^T
 *                with pseudo-inverse of D), Isometry3f compose / inverse, Quaternionf(Matrix3f).
 *                Transform::rotation() is taken as the linear part (what Eigen >= 3.3 does for Isometry;
 *                3.2's SVD polar factor differs from it by float epsilon on a rotation matrix).
 *   OpenCV 2.4.9 cv::Rodrigues (vector -> matrix), Mat::inv(DECOMP_SVD) of a rigid 4x4 (analytic here),
 *                3x3 / 4x4 double products.
 * The reference's host code is built with -O3 -msse2 -msse3 (CMakeLists.txt:57): no FMA on the host,
 * so no fmaf() appears in this file.
 */
#include "kt_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define LEVELS 4 /* ICPOdometry.h:52 */

/* ---------------------------------------------------------------------------------------------- */
/* Eigen Matrix3f::inverse(): cofactors / determinant (Eigen/src/LU/Inverse.h compute_inverse_size3) */
void kto_mat33_inverse(const kto_mat33* in, kto_mat33* out)
{
    const float* m = in->m;
#define M(i, j) m[(i) * 3 + (j)]
#define COF(i, j) (M(((i) + 1) % 3, ((j) + 1) % 3) * M(((i) + 2) % 3, ((j) + 2) % 3) - M(((i) + 1) % 3, ((j) + 2) % 3) * M(((i) + 2) % 3, ((j) + 1) % 3))
    float c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
    float det = (c00 * M(0, 0) + c10 * M(1, 0)) + c20 * M(2, 0);
    float invdet = 1.0f / det;
    float r[9];
    r[0] = c00 * invdet; r[1] = c10 * invdet; r[2] = c20 * invdet;
    r[3] = COF(0, 1) * invdet; r[4] = COF(1, 1) * invdet; r[5] = COF(2, 1) * invdet;
    r[6] = COF(0, 2) * invdet; r[7] = COF(1, 2) * invdet; r[8] = COF(2, 2) * invdet;
#undef COF
#undef M
    memcpy(out->m, r, sizeof(r));
}

/* Eigen LDLT<Matrix<double,6,6>>: in-place pivoted LDL^T (largest remaining diagonal), then
 * solve with the pseudo-inverse of D (Eigen/src/Cholesky/LDLT.h).  ICPOdometry.cpp:127-131. */
void kto_ldlt_solve6(const double Ain[36], const double bin[6], double x[6])
{
    enum { n = 6 };
    double A[n][n];
    int tr[n];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[i][j] = Ain[i * n + j];
    for (int k = 0; k < n; ++k) {
        int p = k;
        double big = fabs(A[k][k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i][i]) > big) { big = fabs(A[i][i]); p = i; }
        tr[k] = p;
        if (p != k) { /* symmetric swap of rows/cols k and p (lower triangle is the one used) */
            for (int j = 0; j < n; ++j) { double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
            for (int i = 0; i < n; ++i) { double t = A[i][k]; A[i][k] = A[i][p]; A[i][p] = t; }
        }
        /* A[k][k] -= sum_j L[k][j]^2 D[j];  column k below the diagonal */
        double d = A[k][k];
        for (int j = 0; j < k; ++j) d -= A[k][j] * A[k][j] * A[j][j];
        A[k][k] = d;
        for (int i = k + 1; i < n; ++i) {
            double s = A[i][k];
            for (int j = 0; j < k; ++j) s -= A[i][j] * A[k][j] * A[j][j];
            A[i][k] = (d != 0.0) ? s / d : s;
        }
    }
    double y[n];
    for (int i = 0; i < n; ++i) y[i] = bin[i];
    for (int k = 0; k < n; ++k)
        if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
    double maxd = 0;
    for (int i = 0; i < n; ++i)
        if (fabs(A[i][i]) > maxd) maxd = fabs(A[i][i]);
    double tol = maxd * DBL_EPSILON;
    if (tol < 1.0 / DBL_MAX) tol = 1.0 / DBL_MAX;
    for (int i = 0; i < n; ++i) y[i] = (fabs(A[i][i]) > tol) ? y[i] / A[i][i] : 0.0;
    for (int i = n - 1; i >= 0; --i)
        for (int j = i + 1; j < n; ++j) y[i] -= A[j][i] * y[j];
    for (int k = n - 1; k >= 0; --k)
        if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < n; ++i) x[i] = y[i];
}

/* cv::Rodrigues, rotation vector -> matrix (OpenCV 2.4 calib3d cvRodrigues2); OdometryProvider.h:54-68 */
void kto_rodrigues(const double r[3], double R[9])
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; ++k) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
}

/* Eigen Quaternionf(Matrix3f) (Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl) */
void kto_quat_from_mat33(const kto_mat33* Rm, float q[4])
{
    const float* m = Rm->m;
#define M(i, j) m[(i) * 3 + (j)]
    float t = M(0, 0) + M(1, 1) + M(2, 2);
    float w, x, y, z;
    if (t > 0.0f) {
        t = sqrtf(t + 1.0f);
        w = 0.5f * t;
        t = 0.5f / t;
        x = (M(2, 1) - M(1, 2)) * t;
        y = (M(0, 2) - M(2, 0)) * t;
        z = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        float v[3];
        t = sqrtf(M(i, i) - M(j, j) - M(k, k) + 1.0f);
        v[i] = 0.5f * t;
        t = 0.5f / t;
        w = (M(k, j) - M(j, k)) * t;
        v[j] = (M(j, i) + M(i, j)) * t;
        v[k] = (M(k, i) + M(i, k)) * t;
        x = v[0]; y = v[1]; z = v[2];
    }
#undef M
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

/* ---------------------------------------------------------------------------------------------- */
static void mat4d_mul(const double a[16], const double b[16], double o[16])
{
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
            r[i * 4 + j] = s;
        }
    memcpy(o, r, sizeof(r));
}
static void mat3d_mul(const double a[9], const double b[9], double o[9])
{
    double r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
            r[i * 3 + j] = s;
        }
    memcpy(o, r, sizeof(r));
}
/* float 3x3 * 3x3 and 3x3 * vec, Eigen coefficient-based product: (a0*b0 + a1*b1) + a2*b2 */
static void mat3f_mul(const float a[9], const float b[9], float o[9])
{
    float r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[i * 3 + j] = (a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j]) + a[i * 3 + 2] * b[2 * 3 + j];
    memcpy(o, r, sizeof(r));
}
static void mat3f_vec(const float a[9], const float v[3], float o[3])
{
    float r[3];
    for (int i = 0; i < 3; ++i) r[i] = (a[i * 3 + 0] * v[0] + a[i * 3 + 1] * v[1]) + a[i * 3 + 2] * v[2];
    memcpy(o, r, sizeof(r));
}

/* The pose update shared by ICPOdometry.cpp:133-178 and RGBDOdometry.cpp:328-373:
 *   currRt = [Rodrigues(x[3:6]) | x[0:3]];  resultRt = currRt * resultRt;
 *   T_curr = T_prev * [R | t]^-1 in float Isometry arithmetic. */
static void pose_update(const double x[6], double resultRt[16], const float Rprev[9], const float tprev[3],
                        float Rcurr[9], float tcurr[3])
{
    double currRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double R[9];
    kto_rodrigues(&x[3], R);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) currRt[i * 4 + j] = R[i * 3 + j];
        currRt[i * 4 + 3] = x[i];
    }
    mat4d_mul(currRt, resultRt, resultRt);
    float rot[9], trans[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) rot[i * 3 + j] = (float)resultRt[i * 4 + j];
        trans[i] = (float)resultRt[i * 4 + 3];
    }
    /* rgbOdom.inverse() for an Isometry: linear^T, -linear^T * translation */
    float rinv[9], tinv[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rinv[i * 3 + j] = rot[j * 3 + i];
    mat3f_vec(rinv, trans, tinv);
    tinv[0] = -tinv[0]; tinv[1] = -tinv[1]; tinv[2] = -tinv[2];
    /* currentT * inverse: linear = Rprev * rinv; translation = Rprev * tinv + tprev */
    float tl[3];
    mat3f_mul(Rprev, rinv, Rcurr);
    mat3f_vec(Rprev, tinv, tl);
    tcurr[0] = tl[0] + tprev[0]; tcurr[1] = tl[1] + tprev[1]; tcurr[2] = tl[2] + tprev[2];
}

/* ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t ts; float pose[16]; int is_loop; } dense_pose;
typedef struct { kto_point* pts; size_t n; int dim; int pr; } slice_rec;
typedef struct { uint64_t utime; float trans[3], rot[9]; } pr_rec; /* PlaceRecognitionInput.h:30-56 without the frame bytes */

struct kto_tracker {
    kto_tracker_config cfg;
    kto_intr intr;
    int N;
    float volume_size[3];
    float voxel_size[3]; /* Volume::voxelSizeMeters / TsdfVolume::getVoxelSize */
    float tranc_dist;
    float volume_basis[3];
    float initial_rotation[9];
    int voxel_wrap[3], v_wrap_copy[3];
    int global_time;
    int parked;
    float Rlast[9], tlast[3]; /* rmats_.back(), tvecs_.back() */
    float current_global_camera[3];
    /* device buffers of the reference, host arrays here */
    int16_t* tsdf; uint8_t* color;
    uint16_t* depths_curr[LEVELS];
    float *vmaps_curr[LEVELS], *nmaps_curr[LEVELS], *vmaps_g_prev[LEVELS], *nmaps_g_prev[LEVELS];
    uint8_t* vmap_curr_color;
    float* depth_raw_scaled;
    kto_point* cloud_device; size_t cloud_cap;
    /* RGBDOdometry state */
    float *last_depth[LEVELS], *next_depth[LEVELS];
    uint8_t *last_image[LEVELS], *next_image[LEVELS];
    int16_t *next_dIdx[LEVELS], *next_dIdy[LEVELS];
    float* point_clouds[LEVELS];
    kto_dataterm* corres[LEVELS];
    /* outputs */
    dense_pose* poses; int n_poses, cap_poses;
    slice_rec* slices; int n_slices, cap_slices;
    double stage_s[6];
    long long last_U, last_S;
    /* ground-truth odometry (-p): camera_trajectory + current_utime, KintinuousTracker.h:237, KintinuousTracker.cpp:216-260 */
    int has_trajectory, n_traj, cap_traj;
    int* traj_key;          /* the map's comparator is std::less<int>: keys are the timestamps narrowed to int */
    float (*traj_T)[12];    /* Isometry3f: R row-major [0..8], t [9..11] */
    uint64_t current_utime;
    /* place recognition (KintinuousTracker.h:216, 248-249, 132-135) */
    float last_pr_trans[3], last_pr_rot[9];
    pr_rec* pr; int n_pr, cap_pr;
};

static int lvl_cols(const kto_tracker* t, int l) { return t->cfg.cols >> l; }
static int lvl_rows(const kto_tracker* t, int l) { return t->cfg.rows >> l; }
static kto_intr lvl_intr(kto_intr k, int l) /* Intr::operator() internal.h:255-259 */
{
    int div = 1 << l;
    kto_intr r = {k.fx / div, k.fy / div, k.cx / div, k.cy / div};
    return r;
}

kto_tracker* kto_tracker_create(const kto_tracker_config* cfg)
{
    kto_tracker* t = calloc(1, sizeof(*t));
    t->cfg = *cfg;
    t->N = cfg->N;
    /* KintinuousTracker ctor KintinuousTracker.cpp:71-182 */
    t->intr.fx = cfg->fx; t->intr.fy = cfg->fy; t->intr.cx = cfg->cx; t->intr.cy = cfg->cy;
    for (int k = 0; k < 3; ++k) {
        t->volume_size[k] = cfg->volume_size;
        t->voxel_size[k] = cfg->volume_size / (float)cfg->N;
    }
    const size_t nvox = (size_t)cfg->N * cfg->N * cfg->N;
    t->tsdf = malloc(nvox * sizeof(int16_t));
    t->color = malloc(nvox * 4);
    for (int k = 0; k < 9; ++k) t->initial_rotation[k] = (k % 4 == 0) ? 1.f : 0.f;
    for (int k = 0; k < 3; ++k) t->volume_basis[k] = t->volume_size[k] * 0.5f;
    if (cfg->static_mode || cfg->dynamic_cube) /* :101-110; the double z offset is narrowed to float by the Vector3f ctor, then subtracted */
        t->volume_basis[2] = t->volume_size[2] * 0.5f - (float)(((double)t->volume_size[2] * 0.5) + (cfg->static_mode ? 0.45 : 0));
    /* :112-113 and TSDFVolume.cpp:89-97 */
    float default_tranc = fmaxf(0.01f, t->volume_size[0] / 100.0f);
    float mc = fmaxf(t->voxel_size[0], fmaxf(t->voxel_size[1], t->voxel_size[2]));
    t->tranc_dist = fmaxf(default_tranc, 2.1f * mc);
    const int P = cfg->cols * cfg->rows;
    for (int l = 0; l < LEVELS; ++l) { /* allocateBuffers :356-382 */
        size_t p = (size_t)lvl_cols(t, l) * lvl_rows(t, l);
        t->depths_curr[l] = calloc(p, sizeof(uint16_t));
        t->vmaps_curr[l] = calloc(3 * p, sizeof(float));
        t->nmaps_curr[l] = calloc(3 * p, sizeof(float));
        t->vmaps_g_prev[l] = calloc(3 * p, sizeof(float));
        t->nmaps_g_prev[l] = calloc(3 * p, sizeof(float));
        t->last_depth[l] = calloc(p, sizeof(float));
        t->next_depth[l] = calloc(p, sizeof(float));
        t->last_image[l] = calloc(p, 1);
        t->next_image[l] = calloc(p, 1);
        t->next_dIdx[l] = calloc(p, sizeof(int16_t));
        t->next_dIdy[l] = calloc(p, sizeof(int16_t));
        t->point_clouds[l] = calloc(3 * p, sizeof(float));
        t->corres[l] = calloc(p, sizeof(kto_dataterm));
    }
    t->vmap_curr_color = calloc((size_t)P, 4);
    t->depth_raw_scaled = calloc((size_t)P, sizeof(float));
    t->cloud_cap = (size_t)P * 3; /* cloud_device_(numPixels * 3) :77 */
    t->cloud_device = malloc(t->cloud_cap * sizeof(kto_point));
    kto_tracker_reset(t);
    return t;
}

void kto_tracker_destroy(kto_tracker* t)
{
    if (!t) return;
    free(t->tsdf); free(t->color);
    for (int l = 0; l < LEVELS; ++l) {
        free(t->depths_curr[l]); free(t->vmaps_curr[l]); free(t->nmaps_curr[l]); free(t->vmaps_g_prev[l]); free(t->nmaps_g_prev[l]);
        free(t->last_depth[l]); free(t->next_depth[l]); free(t->last_image[l]); free(t->next_image[l]);
        free(t->next_dIdx[l]); free(t->next_dIdy[l]); free(t->point_clouds[l]); free(t->corres[l]);
    }
    free(t->vmap_curr_color); free(t->depth_raw_scaled); free(t->cloud_device);
    for (int i = 0; i < t->n_slices; ++i) free(t->slices[i].pts);
    free(t->slices); free(t->poses);
    free(t->traj_key); free(t->traj_T);
    free(t->pr);
    free(t);
}

static void compute_global_camera(kto_tracker* t, const float tcurr[3])
{
    /* KintinuousTracker.cpp:581-595 */
    float initial_trans[3];
    for (int k = 0; k < 3; ++k) {
        initial_trans[k] = (float)((double)t->volume_basis[k] - (double)t->cfg.volume_size * 0.5);
        t->current_global_camera[k] = initial_trans[k];
        t->current_global_camera[k] += (float)t->voxel_wrap[k] * t->voxel_size[k];
        if (tcurr) t->current_global_camera[k] += tcurr[k] - t->volume_basis[k];
    }
}

void kto_tracker_reset(kto_tracker* t)
{
    /* KintinuousTracker::reset :262-354 */
    t->global_time = 0;
    memcpy(t->Rlast, t->initial_rotation, sizeof(t->Rlast));
    memcpy(t->tlast, t->volume_basis, sizeof(t->tlast));
    compute_global_camera(t, NULL); /* uses voxelWrap before it is zeroed, as the reference does */
    t->voxel_wrap[0] = t->voxel_wrap[1] = t->voxel_wrap[2] = 0;
    memcpy(t->last_pr_trans, t->current_global_camera, sizeof(t->last_pr_trans)); /* :290-291 */
    memcpy(t->last_pr_rot, t->initial_rotation, sizeof(t->last_pr_rot));
    t->n_pr = 0;
    t->n_poses = 0;
    for (int i = 0; i < t->n_slices; ++i) free(t->slices[i].pts);
    t->n_slices = 0;
    t->parked = t->cfg.static_mode ? 1 : 0;
    kto_init_volume(t->tsdf, t->N);
    kto_init_color_volume(t->color, t->N);
    memset(t->stage_s, 0, sizeof(t->stage_s));
}

static void v_wrap_copy_update(kto_tracker* t)
{
    /* vWrapCopyUpdate :1075-1085 */
    for (int k = 0; k < 3; ++k) {
        t->v_wrap_copy[k] = t->voxel_wrap[k];
        if (t->v_wrap_copy[k] < 0) t->v_wrap_copy[k] = t->N - ((-t->v_wrap_copy[k]) % t->N);
    }
}

static void push_pose(kto_tracker* t, uint64_t ts, const float R[9], int is_loop)
{
    if (t->n_poses == t->cap_poses) {
        t->cap_poses = t->cap_poses ? 2 * t->cap_poses : 256;
        t->poses = realloc(t->poses, (size_t)t->cap_poses * sizeof(dense_pose));
    }
    dense_pose* p = &t->poses[t->n_poses++];
    p->ts = ts;
    p->is_loop = is_loop;
    for (int i = 0; i < 16; ++i) p->pose[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) p->pose[i * 4 + j] = R[i * 3 + j];
        p->pose[i * 4 + 3] = t->current_global_camera[i];
    }
}

static void push_slice(kto_tracker* t, size_t n, int dim)
{
    if (t->n_slices == t->cap_slices) {
        t->cap_slices = t->cap_slices ? 2 * t->cap_slices : 16;
        t->slices = realloc(t->slices, (size_t)t->cap_slices * sizeof(slice_rec));
    }
    slice_rec* s = &t->slices[t->n_slices++];
    s->n = n;
    s->dim = dim;
    s->pr = -1;
    s->pts = malloc((n ? n : 1) * sizeof(kto_point));
    memcpy(s->pts, t->cloud_device, n * sizeof(kto_point));
}

/* addToPlaceRecognition :917-958 (the copies of the frame's image and depth are the caller's business) */
static int add_to_place_recognition(kto_tracker* t)
{
    if (t->n_pr == t->cap_pr) {
        t->cap_pr = t->cap_pr ? 2 * t->cap_pr : 64;
        t->pr = realloc(t->pr, (size_t)t->cap_pr * sizeof(pr_rec));
    }
    pr_rec* r = &t->pr[t->n_pr];
    r->utime = t->current_utime;
    memcpy(r->trans, t->last_pr_trans, sizeof(r->trans));
    memcpy(r->rot, t->last_pr_rot, sizeof(r->rot));
    return t->n_pr++;
}

/* mutexOutCloudBuffer :1156-1208 */
static void mutex_out_cloud_buffer(kto_tracker* t, size_t cloud_n, float device_tcurr[3], const int trans[3], int pr_frame)
{
    float voxel_trans_size[3];
    for (int k = 0; k < 3; ++k) voxel_trans_size[k] = t->voxel_size[k] * (float)trans[k];
    for (int k = 0; k < 3; ++k) t->tlast[k] -= voxel_trans_size[k];
    int dim = trans[0] > 0 ? 0 : trans[0] < 0 ? 1 : trans[1] > 0 ? 2 : trans[1] < 0 ? 3 : trans[2] > 0 ? 4 : 5; /* CloudSlice.h:33-36 */
    push_slice(t, cloud_n, dim);
    t->slices[t->n_slices - 1].pr = pr_frame;
    for (int k = 0; k < 3; ++k) t->voxel_wrap[k] += trans[k];
    for (int k = 0; k < 3; ++k) device_tcurr[k] -= voxel_trans_size[k];
}

/* ICPOdometry::getIncrementalTransformation  ICPOdometry.cpp:68-186 */
static void icp_odometry(kto_tracker* t, float tcurr[3], float Rcurr[9])
{
    int iters[LEVELS] = {10, 5, 4, 0};
    if (t->cfg.fast_odometry) { iters[0] = 0; iters[1] = 10; iters[2] = 5; iters[3] = 0; }
    const float dist_thres = 0.10f;
    const float angle_thres = (float)sin(20.f * 3.14159254f / 180.f); /* ICPOdometry.h:35-36 */
    float Rprev[9], tprev[3];
    memcpy(Rprev, t->Rlast, sizeof(Rprev));
    memcpy(tprev, t->tlast, sizeof(tprev));
    memcpy(Rcurr, Rprev, sizeof(Rprev));
    memcpy(tcurr, tprev, sizeof(tprev));
    kto_mat33 Rprev_m, Rprev_inv;
    memcpy(Rprev_m.m, Rprev, sizeof(Rprev));
    kto_mat33_inverse(&Rprev_m, &Rprev_inv);
    double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int l = LEVELS - 1; l >= 0; --l)
        for (int it = 0; it < iters[l]; ++it) {
            float A[36], b[6], res[2];
            kto_mat33 Rc;
            memcpy(Rc.m, Rcurr, sizeof(Rc.m));
            kto_icp_step(&Rc, tcurr, t->vmaps_curr[l], t->nmaps_curr[l], &Rprev_inv, tprev, lvl_intr(t->intr, l),
                         t->vmaps_g_prev[l], t->nmaps_g_prev[l], lvl_cols(t, l), lvl_rows(t, l), dist_thres, angle_thres,
                         t->cfg.reduce_order, A, b, res);
            double dA[36], db[6], x[6];
            for (int k = 0; k < 36; ++k) dA[k] = A[k];
            for (int k = 0; k < 6; ++k) db[k] = b[k];
            kto_ldlt_solve6(dA, db, x);
            pose_update(x, resultRt, Rprev, tprev, Rcurr, tcurr);
        }
}

/* RGBDOdometry::populateRGBDData  RGBDOdometry.cpp:140-158 */
static void populate_rgbd(kto_tracker* t, const uint16_t* depth, const uint8_t* rgb, float** dd, uint8_t** di)
{
    kto_depth_to_metres(depth, dd[0], t->cfg.cols, t->cfg.rows, (int)(6.0 * 1000));
    for (int l = 0; l + 1 < LEVELS; ++l) kto_pyr_down_gauss_f32(dd[l], lvl_cols(t, l), lvl_rows(t, l), dd[l + 1]);
    kto_bgr_to_intensity(rgb, di[0], t->cfg.cols, t->cfg.rows);
    for (int l = 0; l + 1 < LEVELS; ++l) kto_pyr_down_gauss_u8(di[l], lvl_cols(t, l), lvl_rows(t, l), di[l + 1]);
}

/* RGBDOdometry::getIncrementalTransformation  RGBDOdometry.cpp:165-393 */
static void rgbd_odometry(kto_tracker* t, const uint16_t* depth, const uint8_t* rgb, float tcurr[3], float Rcurr[9])
{
    int iters[LEVELS];
    if (!t->cfg.use_rgbd_icp) {
        iters[0] = 10; iters[1] = 7; iters[2] = 7; iters[3] = 7;
        if (t->cfg.fast_odometry) { iters[0] = 0; iters[1] = 10; iters[2] = 7; iters[3] = 0; }
    } else {
        iters[0] = 10; iters[1] = 5; iters[2] = 4; iters[3] = 0;
        if (t->cfg.fast_odometry) { iters[0] = 0; iters[1] = 10; iters[2] = 7; iters[3] = 0; }
    }
    const float min_grad[LEVELS] = {12, 5, 3, 1};
    const double SOBEL_SCALE = 1.0 / pow(2.0, 3), MAX_DEPTH_DELTA = 0.07;
    const float dist_thres = 0.10f;
    const float angle_thres = (float)sin(20.f * 3.14159254f / 180.f);
    float Rprev[9], tprev[3];
    memcpy(Rprev, t->Rlast, sizeof(Rprev));
    memcpy(tprev, t->tlast, sizeof(tprev));
    memcpy(Rcurr, Rprev, sizeof(Rprev));
    memcpy(tcurr, tprev, sizeof(tprev));
    kto_mat33 Rprev_m, Rprev_inv;
    memcpy(Rprev_m.m, Rprev, sizeof(Rprev));
    kto_mat33_inverse(&Rprev_m, &Rprev_inv);

    populate_rgbd(t, depth, rgb, t->next_depth, t->next_image);
    for (int l = 0; l < LEVELS; ++l) kto_derivative_images(t->next_image[l], lvl_cols(t, l), lvl_rows(t, l), t->next_dIdx[l], t->next_dIdy[l]);

    double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const double ifx = t->intr.fx, ify = t->intr.fy, icx = t->intr.cx, icy = t->intr.cy; /* IntrDoublePrecision from the float Intr :72-75 */
    for (int l = LEVELS - 1; l >= 0; --l) {
        const int cols = lvl_cols(t, l), rows = lvl_rows(t, l);
        kto_project_to_cloud(t->last_depth[l], cols, rows, t->point_clouds[l], ifx, ify, icx, icy, l);
        const int div = 1 << l;
        const double K[9] = {ifx / div, 0, icx / div, 0, ify / div, icy / div, 0, 0, 1};
        /* K.inv(): analytic inverse of the upper-triangular K */
        const double Kinv[9] = {1.0 / K[0], 0, -K[2] / K[0], 0, 1.0 / K[4], -K[5] / K[4], 0, 0, 1};
        for (int j = 0; j < iters[l]; ++j) {
            /* Rt = resultRt.inv(DECOMP_SVD): rigid inverse */
            double Rinv[9], tinv[3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) Rinv[a * 3 + b] = resultRt[b * 4 + a];
            for (int a = 0; a < 3; ++a) tinv[a] = -(Rinv[a * 3 + 0] * resultRt[3] + Rinv[a * 3 + 1] * resultRt[7] + Rinv[a * 3 + 2] * resultRt[11]);
            double KR[9], KRK[9], Kt[3];
            mat3d_mul(K, Rinv, KR);
            mat3d_mul(KR, Kinv, KRK);
            for (int a = 0; a < 3; ++a) Kt[a] = K[a * 3 + 0] * tinv[0] + K[a * 3 + 1] * tinv[1] + K[a * 3 + 2] * tinv[2];
            float kt[3] = {(float)Kt[0], (float)Kt[1], (float)Kt[2]};
            kto_mat33 krkinv;
            for (int n = 0; n < 9; ++n) krkinv.m[n] = (float)KRK[n];
            int sigma = 0, rgb_size = 0;
            float min_scale = (float)(pow(min_grad[l], 2.0) / pow(SOBEL_SCALE, 2.0));
            kto_rgb_residual(min_scale, t->next_dIdx[l], t->next_dIdy[l], t->last_depth[l], t->next_depth[l], t->last_image[l],
                             t->next_image[l], cols, rows, t->corres[l], (float)MAX_DEPTH_DELTA, kt, &krkinv, &sigma, &rgb_size);
            /* sigma quirk RGBDOdometry.cpp:253: sqrt(count) unless sigma/count == 0 */
            float sigma_val = sqrtf(((float)sigma / rgb_size == 0) ? 1 : rgb_size);
            float A_icp[36], b_icp[6], res[2];
            memset(A_icp, 0, sizeof(A_icp));
            memset(b_icp, 0, sizeof(b_icp));
            if (t->cfg.use_rgbd_icp) {
                kto_mat33 Rc;
                memcpy(Rc.m, Rcurr, sizeof(Rc.m));
                kto_icp_step(&Rc, tcurr, t->vmaps_curr[l], t->nmaps_curr[l], &Rprev_inv, tprev, lvl_intr(t->intr, l),
                             t->vmaps_g_prev[l], t->nmaps_g_prev[l], cols, rows, dist_thres, angle_thres, t->cfg.reduce_order,
                             A_icp, b_icp, res);
            }
            float A_rgbd[36], b_rgbd[6];
            kto_intr li = lvl_intr(t->intr, l);
            kto_rgb_step(t->corres[l], sigma_val, t->point_clouds[l], li.fx, li.fy, t->next_dIdx[l], t->next_dIdy[l],
                         (float)SOBEL_SCALE, cols, rows, t->cfg.reduce_order, A_rgbd, b_rgbd);
            double dA[36], db[6], x[6];
            if (t->cfg.use_rgbd_icp) {
                const double w = 10;
                for (int k = 0; k < 36; ++k) dA[k] = (double)A_rgbd[k] + w * w * (double)A_icp[k];
                for (int k = 0; k < 6; ++k) db[k] = (double)b_rgbd[k] + w * (double)b_icp[k];
            } else {
                for (int k = 0; k < 36; ++k) dA[k] = A_rgbd[k];
                for (int k = 0; k < 6; ++k) db[k] = b_rgbd[k];
            }
            kto_ldlt_solve6(dA, db, x);
            pose_update(x, resultRt, Rprev, tprev, Rcurr, tcurr);
        }
    }
    for (int l = 0; l < LEVELS; ++l) { /* swap last/next :377-381 */
        float* fd = t->last_depth[l]; t->last_depth[l] = t->next_depth[l]; t->next_depth[l] = fd;
        uint8_t* ui = t->last_image[l]; t->last_image[l] = t->next_image[l]; t->next_image[l] = ui;
    }
    float d[3] = {tcurr[0] - tprev[0], tcurr[1] - tprev[1], tcurr[2] - tprev[2]};
    if (sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > 0.3) { /* :383-387 */
        memcpy(Rcurr, Rprev, sizeof(Rprev));
        memcpy(tcurr, tprev, sizeof(tprev));
    }
}

/* KintinuousTracker::loadTrajectory KintinuousTracker.cpp:216-260 (the text parsing stays with the caller).
 * pose7 = n x {x y z qx qy qz qw}.  T.setIdentity(); T.pretranslate(t).rotate(q): linear = Quaternionf::toRotationMatrix(),
 * translation = t; a repeated key (as the int comparator sees it) overwrites the earlier entry. */
void kto_tracker_load_trajectory(kto_tracker* t, int n, const uint64_t* utimes, const float* pose7)
{
    t->has_trajectory = 1;
    for (int i = 0; i < n; ++i) {
        const float* p = pose7 + (size_t)i * 7;
        const float x = p[3], y = p[4], z = p[5], w = p[6];
        /* Eigen QuaternionBase::toRotationMatrix */
        const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
        const float twx = tx * w, twy = ty * w, twz = tz * w;
        const float txx = tx * x, txy = ty * x, txz = tz * x;
        const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
        float T[12];
        T[0] = 1.f - (tyy + tzz); T[1] = txy - twz; T[2] = txz + twy;
        T[3] = txy + twz; T[4] = 1.f - (txx + tzz); T[5] = tyz - twx;
        T[6] = txz - twy; T[7] = tyz + twx; T[8] = 1.f - (txx + tyy);
        T[9] = p[0]; T[10] = p[1]; T[11] = p[2];
        const int key = (int)(uint32_t)utimes[i];
        int at = -1;
        for (int k = 0; k < t->n_traj; ++k)
            if (t->traj_key[k] == key) { at = k; break; }
        if (at < 0) {
            if (t->n_traj == t->cap_traj) {
                t->cap_traj = t->cap_traj ? 2 * t->cap_traj : 256;
                t->traj_key = realloc(t->traj_key, (size_t)t->cap_traj * sizeof(int));
                t->traj_T = realloc(t->traj_T, (size_t)t->cap_traj * sizeof(*t->traj_T));
            }
            at = t->n_traj++;
            t->traj_key[at] = key;
        }
        memcpy(t->traj_T[at], T, sizeof(T));
    }
    t->current_utime = 0; /* :259 */
}

static const float* traj_find(const kto_tracker* t, uint64_t utime)
{
    const int key = (int)(uint32_t)utime;
    for (int k = 0; k < t->n_traj; ++k)
        if (t->traj_key[k] == key) return t->traj_T[k];
    return NULL;
}

/* GroundTruthOdometry::getIncrementalTransformation GroundTruthOdometry.cpp:42-74.  last_utime there is a reference to the
 * tracker's current_utime, i.e. the previous frame's timestamp; a previous timestamp of 0 means "no motion" (:50).
 * currentTsdf * M.inverse() * delta * M is evaluated left to right as 4x4 float products; M is a signed permutation, so the
 * outer two products only move and negate columns and the one rounding product is (currentTsdf*M^-1) * delta, whose sums run
 * over the PERMUTED columns in order k = 0..3. */
static void ground_truth_odometry(kto_tracker* t, uint64_t timestamp, float tcurr[3], float Rcurr[9])
{
    memcpy(Rcurr, t->Rlast, 9 * sizeof(float));
    memcpy(tcurr, t->tlast, 3 * sizeof(float));
    if (t->current_utime == 0 || t->n_traj == 0) return;
    const float* Ta = traj_find(t, t->current_utime);
    const float* Tb = traj_find(t, timestamp);
    float ident[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    if (!Ta) Ta = ident; /* operator[] default-constructs; cannot happen: the previous frame passed preRun */
    if (!Tb) Tb = ident;
    /* delta = Ta.inverse() * Tb.  Isometry inverse: linear^T, -(linear^T) * translation */
    float Ai[9], ai[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Ai[r * 3 + c] = Ta[c * 3 + r];
    for (int r = 0; r < 3; ++r) ai[r] = ((-Ai[r * 3 + 0]) * Ta[9] + (-Ai[r * 3 + 1]) * Ta[10]) + (-Ai[r * 3 + 2]) * Ta[11];
    float D[16]; /* delta as a 4x4, row-major */
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) D[r * 4 + c] = (Ai[r * 3 + 0] * Tb[0 * 3 + c] + Ai[r * 3 + 1] * Tb[1 * 3 + c]) + Ai[r * 3 + 2] * Tb[2 * 3 + c];
        D[r * 4 + 3] = ((Ai[r * 3 + 0] * Tb[9] + Ai[r * 3 + 1] * Tb[10]) + Ai[r * 3 + 2] * Tb[11]) + ai[r];
    }
    D[12] = 0.f; D[13] = 0.f; D[14] = 0.f; D[15] = 1.f;
    /* P = currentTsdf * M^-1 (M^-1 = M^T): columns (T2, -T0, -T1, T3) of [Rprev | tprev; 0 0 0 1] */
    float P[16];
    for (int r = 0; r < 3; ++r) {
        P[r * 4 + 0] = t->Rlast[r * 3 + 2];
        P[r * 4 + 1] = -t->Rlast[r * 3 + 0];
        P[r * 4 + 2] = -t->Rlast[r * 3 + 1];
        P[r * 4 + 3] = t->tlast[r];
    }
    P[12] = 0.f; P[13] = 0.f; P[14] = 0.f; P[15] = 1.f;
    float Q[12];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
            Q[r * 4 + c] = ((P[r * 4 + 0] * D[0 * 4 + c] + P[r * 4 + 1] * D[1 * 4 + c]) + P[r * 4 + 2] * D[2 * 4 + c]) + P[r * 4 + 3] * D[3 * 4 + c];
    /* X = Q * M: columns (-Q1, -Q2, Q0, Q3); rot = linear part (Isometry::rotation()), trans = translation */
    for (int r = 0; r < 3; ++r) {
        Rcurr[r * 3 + 0] = -Q[r * 4 + 1];
        Rcurr[r * 3 + 1] = -Q[r * 4 + 2];
        Rcurr[r * 3 + 2] = Q[r * 4 + 0];
        tcurr[r] = Q[r * 4 + 3];
    }
}

static int voxel_trans(float translation, float voxel, int thresh);

/* KintinuousTracker::rodrigues2 (KintinuousTracker.cpp:1210-1255): rotation matrix -> axis-angle, cv::Rodrigues' inverse branch in
 * double.  The reference first re-orthonormalises the matrix (JacobiSVD, R = U V^T); for the rotation matrices this is called on,
 * that polar factor differs from the input by float rounding only and the SVD is not restated: R is used as is. */
static void rodrigues2(const float R[9], float out[3])
{
    double rx = (double)(R[7] - R[5]), ry = (double)(R[2] - R[6]), rz = (double)(R[3] - R[1]); /* float differences, widened */
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (double)((R[0] + R[4] + R[8]) - 1) * 0.5; /* R.trace() is a float */
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) rx = ry = rz = 0;
        else {
            double tt = (R[0] + 1) * 0.5;
            rx = sqrt(tt > 0.0 ? tt : 0.0);
            tt = (R[4] + 1) * 0.5;
            ry = sqrt(tt > 0.0 ? tt : 0.0) * (R[1] < 0 ? -1.0 : 1.0);
            tt = (R[8] + 1) * 0.5;
            rz = sqrt(tt > 0.0 ? tt : 0.0) * (R[2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    out[0] = (float)rx; out[1] = (float)ry; out[2] = (float)rz;
}

/* KintinuousTracker::repositionCube :384-442.  cos / sin are the C library's double versions (global ::cos on a float argument). */
void kto_reposition_cube(const float R[9], const float tlast[3], float volume_size, const float voxel_size[3], int thresh, float basis[3])
{
    float rot[3];
    rodrigues2(R, rot);
    const float yRot = rot[1];
    const float PI = 3.14159265359f;
    const float radius = (float)(volume_size * 0.5);
    float np_[3] = {basis[0], basis[1], basis[2]};
    np_[0] = (float)(radius * (cos(yRot + (PI / 2)) + 1.0f));
    np_[2] = (float)(radius * (sin(yRot - (PI / 2)) + 1.0f));
    int moved = 0;
    for (int k = 0; k < 3; ++k) {
        const int v = voxel_trans(tlast[k] - np_[k], voxel_size[k], thresh);
        moved = moved || v >= thresh || v <= -thresh;
    }
    if (moved) { basis[0] = np_[0]; basis[1] = np_[1]; basis[2] = np_[2]; }
}

static int voxel_trans(float translation, float voxel, int thresh)
{
    /* KintinuousTracker.cpp:640-667 */
    int f = (int)floorf(translation / voxel);
    if (f < 0) return (-thresh > f) ? -thresh : f;
    return thresh < f ? thresh : f;
}

void kto_tracker_process_frame(kto_tracker* t, const uint16_t* depth_raw, const uint8_t* colors, uint64_t timestamp)
{
    const int cols = t->cfg.cols, rows = t->cfg.rows, N = t->N;
    /* the odometry provider objects of the ctor :128-178: ground truth wins over the RGB-D flags */
    const int gt = t->has_trajectory;
    const int icp = !gt && !(t->cfg.use_rgbd || t->cfg.use_rgbd_icp);
    const int rgbd = !gt && !icp;
    const int angle_color = !t->cfg.disable_color_angle;
    double t0 = omp_get_wtime(), t1;

    if (gt && !traj_find(t, timestamp)) return; /* GroundTruthOdometry::preRun :89-111, KintinuousTracker.cpp:460-463 */

    if (icp || t->cfg.use_rgbd_icp || !t->cfg.disable_color_angle) { /* [A] :465-479 */
        kto_bilateral_filter(depth_raw, t->depths_curr[0], cols, rows);
        for (int l = 1; l < LEVELS; ++l) kto_pyr_down(t->depths_curr[l - 1], lvl_cols(t, l - 1), lvl_rows(t, l - 1), t->depths_curr[l]);
        for (int l = 0; l < LEVELS; ++l) {
            kto_create_vmap(lvl_intr(t->intr, l), t->depths_curr[l], lvl_cols(t, l), lvl_rows(t, l), t->vmaps_curr[l]);
            kto_create_nmap(t->vmaps_curr[l], lvl_cols(t, l), lvl_rows(t, l), t->nmaps_curr[l]);
        }
    }
    t1 = omp_get_wtime(); t->stage_s[0] += t1 - t0; t0 = t1;

    if (t->global_time == 0) { /* [B] :481-557 */
        kto_mat33 Rcam, Rcam_inv;
        memcpy(Rcam.m, t->Rlast, sizeof(Rcam.m));
        kto_mat33_inverse(&Rcam, &Rcam_inv);
        const int empty[3] = {0, 0, 0};
        if (rgbd) populate_rgbd(t, depth_raw, colors, t->last_depth, t->last_image); /* firstRun */
        t->last_U = kto_integrate_tsdf(depth_raw, cols, rows, t->intr, t->volume_size, &Rcam_inv, t->tlast, t->tranc_dist, t->tsdf,
                                       t->depth_raw_scaled, empty, t->color, colors, t->nmaps_curr[0], angle_color, N);
        t1 = omp_get_wtime(); t->stage_s[3] += t1 - t0; t0 = t1;
        for (int l = 0; l < LEVELS; ++l)
            kto_transform_maps(t->vmaps_curr[l], t->nmaps_curr[l], lvl_cols(t, l), lvl_rows(t, l), &Rcam, t->tlast, t->vmaps_g_prev[l], t->nmaps_g_prev[l]);
        ++t->global_time;
        t->current_utime = timestamp; /* :527-528 */
        push_pose(t, timestamp, t->Rlast, 1);
        if (t->cfg.place_recognition) add_to_place_recognition(t); /* :546-549 */
        return;
    }

    float Rcurr[9], tcurr[3];
    if (gt) ground_truth_odometry(t, timestamp, tcurr, Rcurr);
    else if (icp) icp_odometry(t, tcurr, Rcurr); /* [C] :564-572 */
    else rgbd_odometry(t, depth_raw, colors, tcurr, Rcurr);
    t->current_utime = timestamp; /* :574-575 */
    t1 = omp_get_wtime(); t->stage_s[1] += t1 - t0; t0 = t1;

    memcpy(t->Rlast, Rcurr, sizeof(Rcurr)); /* [D] rmats_/tvecs_ push */
    memcpy(t->tlast, tcurr, sizeof(tcurr));
    compute_global_camera(t, tcurr);
    if (t->cfg.dynamic_cube) /* :597-600; its threshold while parked is VOLUME_X (x 3 with -sm), :403 */
        kto_reposition_cube(Rcurr, t->tlast, t->cfg.volume_size, t->voxel_size,
                            t->parked ? (t->cfg.static_mode ? t->N * 3 : t->N) : t->cfg.voxel_shift, t->volume_basis);

    kto_mat33 Rc, Rc_inv;
    memcpy(Rc.m, Rcurr, sizeof(Rc.m));
    kto_mat33_inverse(&Rc, &Rc_inv);

    /* [E] place-recognition sampling :601-624 */
    int shift_send = 0, is_loop_pose = 0;
    if (t->cfg.place_recognition) {
        float rel[9], rvec[3];
        mat3f_mul(Rc_inv.m, t->last_pr_rot, rel);            /* Rcurr.inverse() * lastPlaceRecognitionRot */
        rodrigues2(rel, rvec);
        const float rnorm = sqrtf((rvec[0] * rvec[0] + rvec[1] * rvec[1]) + rvec[2] * rvec[2]);
        const float dx = t->current_global_camera[0] - t->last_pr_trans[0], dy = t->current_global_camera[1] - t->last_pr_trans[1],
                    dz = t->current_global_camera[2] - t->last_pr_trans[2];
        const float tnorm = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float alpha = 1.f, place_recognition_movement = 0.15f; /* :76 */
        if ((rnorm + alpha * tnorm) / 2 >= place_recognition_movement) {
            memcpy(t->last_pr_rot, Rcurr, sizeof(t->last_pr_rot));
            memcpy(t->last_pr_trans, t->current_global_camera, sizeof(t->last_pr_trans));
            add_to_place_recognition(t);
            is_loop_pose = 1;
        } else
            shift_send = 1;
    }

    /* [F] shift decision :627-667 */
    float current_translation[3];
    for (int k = 0; k < 3; ++k) current_translation[k] = t->tlast[k] - t->volume_basis[k];
    const int thresh = t->parked ? INT_MAX : t->cfg.voxel_shift;
    int vt[3];
    for (int k = 0; k < 3; ++k) vt[k] = voxel_trans(current_translation[k], t->voxel_size[k], thresh);
    const int ov = t->cfg.overlap;
    for (int axis = 0; axis < 3; ++axis) { /* :669-833 */
        v_wrap_copy_update(t);
        int cycled = 0;
        size_t cloud_n = 0;
        int lo[3] = {0, 0, 0}, hi[3] = {N, N, N};
        if (vt[axis] >= thresh) {
            lo[axis] = 0; hi[axis] = vt[axis] + 1 + ov;
            cloud_n = kto_extract_cloud_slice(t->tsdf, t->volume_size, t->cloud_device, t->cloud_cap, t->v_wrap_copy, t->color,
                                              lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], 1, t->voxel_wrap, N);
            kto_clear_volume(t->tsdf, 2, N, axis, 0, t->voxel_wrap[axis], t->voxel_wrap[axis] + vt[axis]);
            kto_clear_volume(t->color, 4, N, axis, 0, t->voxel_wrap[axis], t->voxel_wrap[axis] + vt[axis]);
            cycled = 1;
        } else if (vt[axis] <= -thresh) {
            if (axis == 2) { lo[2] = N + (vt[2] - ov) - 1; hi[2] = N - 1; } /* z-minus off by one :805 */
            else { lo[axis] = N + (vt[axis] - ov); hi[axis] = N; }
            cloud_n = kto_extract_cloud_slice(t->tsdf, t->volume_size, t->cloud_device, t->cloud_cap, t->v_wrap_copy, t->color,
                                              lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], 1, t->voxel_wrap, N);
            kto_clear_volume(t->tsdf, 2, N, axis, 1, t->voxel_wrap[axis], t->voxel_wrap[axis] + vt[axis]);
            kto_clear_volume(t->color, 4, N, axis, 1, t->voxel_wrap[axis], t->voxel_wrap[axis] + vt[axis]);
            cycled = 1;
        }
        if (cycled) {
            int next_pr_frame = -1;
            if (shift_send) { /* :706-717, :762-771, :816-825 */
                memcpy(t->last_pr_rot, Rcurr, sizeof(t->last_pr_rot));
                memcpy(t->last_pr_trans, t->current_global_camera, sizeof(t->last_pr_trans));
                next_pr_frame = add_to_place_recognition(t);
                is_loop_pose = 1;
                shift_send = 0;
            }
            int trans[3] = {0, 0, 0};
            trans[axis] = vt[axis];
            mutex_out_cloud_buffer(t, cloud_n, tcurr, trans, next_pr_frame);
        }
    }
    v_wrap_copy_update(t);
    t1 = omp_get_wtime(); t->stage_s[2] += t1 - t0; t0 = t1;

    /* [H] integrate with raw depth, current-frame level-0 normals :864-876 */
    t->last_U = kto_integrate_tsdf(depth_raw, cols, rows, t->intr, t->volume_size, &Rc_inv, tcurr, t->tranc_dist, t->tsdf,
                                   t->depth_raw_scaled, t->v_wrap_copy, t->color, colors, t->nmaps_curr[0], angle_color, N);
    t1 = omp_get_wtime(); t->stage_s[3] += t1 - t0; t0 = t1;
    v_wrap_copy_update(t);
    /* [I] raycast :880-890 */
    t->last_S = kto_raycast(t->intr, &Rc, tcurr, t->tranc_dist, t->volume_size, t->tsdf, t->vmaps_g_prev[0], t->nmaps_g_prev[0],
                            cols, rows, t->v_wrap_copy, t->vmap_curr_color, t->color, N);
    t1 = omp_get_wtime(); t->stage_s[4] += t1 - t0; t0 = t1;
    if (icp || t->cfg.use_rgbd_icp) /* [J] :892-899 */
        for (int l = 1; l < LEVELS; ++l) {
            kto_resize_vmap(t->vmaps_g_prev[l - 1], lvl_cols(t, l - 1), lvl_rows(t, l - 1), t->vmaps_g_prev[l]);
            kto_resize_nmap(t->nmaps_g_prev[l - 1], lvl_cols(t, l - 1), lvl_rows(t, l - 1), t->nmaps_g_prev[l]);
        }
    t1 = omp_get_wtime(); t->stage_s[5] += t1 - t0;
    ++t->global_time;
    push_pose(t, timestamp, Rcurr, is_loop_pose); /* [K] :903-909 */
}

void kto_tracker_finalise(kto_tracker* t)
{
    /* finalise :1003-1048 */
    v_wrap_copy_update(t);
    size_t n = kto_extract_cloud_slice(t->tsdf, t->volume_size, t->cloud_device, t->cloud_cap, t->v_wrap_copy, t->color, 0, t->N, 0,
                                       t->N, 0, t->N, 1, t->voxel_wrap, t->N);
    push_slice(t, n, 7 /* CloudSlice::FINAL */);
    if (t->cfg.place_recognition) { /* :1035-1045 */
        memcpy(t->last_pr_rot, t->Rlast, sizeof(t->last_pr_rot));
        memcpy(t->last_pr_trans, t->current_global_camera, sizeof(t->last_pr_trans));
        t->slices[t->n_slices - 1].pr = add_to_place_recognition(t);
    }
}

int kto_tracker_num_pr_samples(const kto_tracker* t) { return t->n_pr; }
void kto_tracker_pr_sample(const kto_tracker* t, int i, uint64_t* utime, float trans[3], float rot[9])
{
    *utime = t->pr[i].utime;
    memcpy(trans, t->pr[i].trans, sizeof(t->pr[i].trans));
    memcpy(rot, t->pr[i].rot, sizeof(t->pr[i].rot));
}
int kto_tracker_slice_pr_id(const kto_tracker* t, int i) { return t->slices[i].pr; }

void kto_tracker_get_pose(const kto_tracker* t, float R[9], float tvec[3], float global_cam[3])
{
    memcpy(R, t->Rlast, sizeof(t->Rlast));
    memcpy(tvec, t->tlast, sizeof(t->tlast));
    memcpy(global_cam, t->current_global_camera, sizeof(t->current_global_camera));
}
int kto_tracker_num_poses(const kto_tracker* t) { return t->n_poses; }
void kto_tracker_get_dense_pose(const kto_tracker* t, int i, uint64_t* ts, float pose16[16], int* is_loop)
{
    *ts = t->poses[i].ts;
    memcpy(pose16, t->poses[i].pose, sizeof(t->poses[i].pose));
    *is_loop = t->poses[i].is_loop;
}
void kto_tracker_get_volume_basis(const kto_tracker* t, float basis[3]) { memcpy(basis, t->volume_basis, 3 * sizeof(float)); }
void kto_tracker_get_voxel_wrap(const kto_tracker* t, int wrap[3]) { memcpy(wrap, t->voxel_wrap, sizeof(t->voxel_wrap)); }
int kto_tracker_num_slices(const kto_tracker* t) { return t->n_slices; }
size_t kto_tracker_slice_size(const kto_tracker* t, int i) { return t->slices[i].n; }
int kto_tracker_slice_dimension(const kto_tracker* t, int i) { return t->slices[i].dim; }
const kto_point* kto_tracker_slice_points(const kto_tracker* t, int i) { return t->slices[i].pts; }
const int16_t* kto_tracker_volume(const kto_tracker* t) { return t->tsdf; }
const uint8_t* kto_tracker_color_volume(const kto_tracker* t) { return t->color; }
const float* kto_tracker_vmap_g_prev(const kto_tracker* t, int l) { return t->vmaps_g_prev[l]; }
const float* kto_tracker_nmap_g_prev(const kto_tracker* t, int l) { return t->nmaps_g_prev[l]; }
float kto_tracker_trunc_dist(const kto_tracker* t) { return t->tranc_dist; }
void kto_tracker_stage_seconds(const kto_tracker* t, double out[6]) { memcpy(out, t->stage_s, sizeof(t->stage_s)); }
void kto_tracker_last_counts(const kto_tracker* t, long long* U, long long* S) { *U = t->last_U; *S = t->last_S; }
