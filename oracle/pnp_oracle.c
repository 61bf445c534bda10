/*
 * pnp_oracle.c -- CPU restatement of cv::solvePnP(SOLVEPNP_ITERATIVE) + cv::projectPoints as the
 * reference calls them (aruco_detect/src/aruco_detect.cpp:247 and :210; stag common.hpp:34-46).
 * TEST INFRASTRUCTURE ONLY (see aruco_oracle.h).
 *
 * Follows OpenCV 4.2.0 calib3d: solvepnp.cpp solvePnPGeneric(ITERATIVE) ->
 * calibration.cpp cvFindExtrinsicCameraParams2 (planar branch: cvUndistortPoints, homography init,
 * CvLevMarq refinement through cvProjectPoints2 with analytic Jacobians), fundam.cpp
 * HomographyEstimatorCallback::runKernel, calibration.cpp cvRodrigues2, compat_ptsetreg.cpp CvLevMarq.
 * Where OpenCV calls a LAPACK-style kernel whose internal operation order is not part of the
 * algorithm (cv::eigen on the 9x9 DLT matrix, cv::SVD on 3x3/6x6), a cyclic Jacobi solver is used;
 * those only feed the initial guess / a damped linear solve, the converged Levenberg-Marquardt
 * minimum is what is compared (tolerances in tests/).
 */
#include "aruco_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* ---- small dense helpers --------------------------------------------------------------------- */
/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9).
 * A is destroyed; w = eigenvalues (descending), V rows = eigenvectors (like cv::eigen). */
static void jacobi_eigen(double *A, int n, double *w, double *V)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = i == j ? 1. : 0.;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                double theta = (aqq - app) / (2. * apq);
                double t = (theta >= 0 ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
                double c = 1. / sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vpk = V[p * n + k], vqk = V[q * n + k];
                    V[p * n + k] = c * vpk - s * vqk;
                    V[q * n + k] = s * vpk + c * vqk;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    /* sort descending (selection sort keeps it deterministic) */
    for (int i = 0; i < n - 1; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++)
            if (w[j] > w[m]) m = j;
        if (m != i) {
            double t = w[i];
            w[i] = w[m];
            w[m] = t;
            for (int k = 0; k < n; k++) {
                t = V[i * n + k];
                V[i * n + k] = V[m * n + k];
                V[m * n + k] = t;
            }
        }
    }
}

static void mat3_mul(const double *A, const double *B, double *C)
{
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    memcpy(C, T, sizeof(T));
}

/* nearest rotation in the sense of SVD: R <- U*Vt where R = U*S*Vt  ( = R*(RtR)^(-1/2) ) */
static void orthonormalize3(double *R)
{
    double RtR[9], w[3], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) RtR[i * 3 + j] = R[i] * R[j] + R[3 + i] * R[3 + j] + R[6 + i] * R[6 + j];
    jacobi_eigen(RtR, 3, w, V);
    double P[9] = {0};
    for (int k = 0; k < 3; k++) {
        double is = w[k] > 1e-300 ? 1. / sqrt(w[k]) : 0.;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) P[i * 3 + j] += V[k * 3 + i] * V[k * 3 + j] * is;
    }
    mat3_mul(R, P, R);
}

/* calibration.cpp cvRodrigues2: vector -> matrix, optional jacobian J (3 x 9, dR/dr) */
static void rodrigues_v2m(const double r_in[3], double R[9], double *J)
{
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
        if (J) {
            memset(J, 0, 27 * sizeof(double));
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1. : 0.) + c1 * rrt[k] + s * r_x[k];
    if (J) {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double drrt[27] = {rx + rx, ry, rz, ry, 0,       0,  rz, 0,  0,       0, rx, 0, rx, ry + ry,
                           rz,      0,  rz, 0,  0,       0,  rx, 0,  0,       ry, rx, ry, rz + rz};
        const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

/* cvRodrigues2: matrix -> vector */
static void rodrigues_m2v(const double Rin[9], double r[3])
{
    double R[9];
    memcpy(R, Rin, sizeof(R));
    orthonormalize3(R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0)
            rx = ry = rz = 0;
        else {
            t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    r[0] = rx;
    r[1] = ry;
    r[2] = rz;
}

/* calibration.cpp cvProjectPoints2Internal (k = D[0..4] plumb-bob; no rational/prism/tilt terms).
 * M: n x 3 doubles; m out: n x 2; dpdr/dpdt: (2n) x 3 each (row stride 3) or NULL. */
static void project_points(const double *M, int n, const double rv[3], const double tv[3], const double K[9],
                           const double Din[5], double *m, double *dpdr, double *dpdt)
{
    double R[9], dRdr[27];
    double k[5] = {0, 0, 0, 0, 0};
    if (Din) memcpy(k, Din, sizeof(k));
    rodrigues_v2m(rv, R, dpdr ? dRdr : 0);
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int i = 0; i < n; i++) {
        double X = M[i * 3], Y = M[i * 3 + 1], Z = M[i * 3 + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tv[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tv[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tv[2];
        double r2, r4, r6, a1, a2, a3, cdist, icdist2;
        double xd, yd;
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        r2 = x * x + y * y;
        r4 = r2 * r2;
        r6 = r4 * r2;
        a1 = 2 * x * y;
        a2 = r2 + 2 * x * x;
        a3 = r2 + 2 * y * y;
        cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        icdist2 = 1.;
        xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2;
        yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1;
        m[i * 2] = xd * fx + cx;
        m[i * 2 + 1] = yd * fy + cy;
        if (dpdt) {
            double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            for (int j = 0; j < 3; j++) {
                double dr2dt = 2 * x * dxdt[j] + 2 * y * dydt[j];
                double dcdist_dt = k[0] * dr2dt + 2 * k[1] * r2 * dr2dt + 3 * k[4] * r4 * dr2dt;
                double da1dt = 2 * (x * dydt[j] + y * dxdt[j]);
                double dmxdt = (dxdt[j] * cdist * icdist2 + x * dcdist_dt * icdist2 + k[2] * da1dt +
                                k[3] * (dr2dt + 4 * x * dxdt[j]));
                double dmydt = (dydt[j] * cdist * icdist2 + y * dcdist_dt * icdist2 +
                                k[2] * (dr2dt + 4 * y * dydt[j]) + k[3] * da1dt);
                dpdt[(2 * i) * 3 + j] = fx * dmxdt;
                dpdt[(2 * i + 1) * 3 + j] = fy * dmydt;
            }
        }
        if (dpdr) {
            double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                               X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
            double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                               X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
            double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                               X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
            for (int j = 0; j < 3; j++) {
                double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                double dr2dr = 2 * x * dxdr + 2 * y * dydr;
                double dcdist_dr = (k[0] + 2 * k[1] * r2 + 3 * k[4] * r4) * dr2dr;
                double da1dr = 2 * (x * dydr + y * dxdr);
                double dmxdr = (dxdr * cdist * icdist2 + x * dcdist_dr * icdist2 + k[2] * da1dr +
                                k[3] * (dr2dr + 4 * x * dxdr));
                double dmydr = (dydr * cdist * icdist2 + y * dcdist_dr * icdist2 +
                                k[2] * (dr2dr + 4 * y * dydr) + k[3] * da1dr);
                dpdr[(2 * i) * 3 + j] = fx * dmxdr;
                dpdr[(2 * i + 1) * 3 + j] = fy * dmydr;
            }
        }
    }
}

int ora_project_points(const double K[9], const double D[5], const double rvec[3], const double tvec[3],
                       const float *obj3, int n, double *img2)
{
    if (n > 64) return -1;
    double M[64 * 3];
    for (int i = 0; i < 3 * n; i++) M[i] = obj3[i];
    project_points(M, n, rvec, tvec, K, D, img2, 0, 0);
    return 0;
}

/* undistort.cpp cvUndistortPointsInternal with TermCriteria(COUNT, 5), R = I, P = none */
static void undistort_points(const double *src, int n, const double K[9], const double Din[5], double *dst)
{
    double k[5] = {0, 0, 0, 0, 0};
    if (Din) memcpy(k, Din, sizeof(k));
    double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    for (int i = 0; i < n; i++) {
        double x = src[2 * i], y = src[2 * i + 1], x0, y0, u = x, v = y;
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        if (Din) {
            x0 = x;
            y0 = y;
            for (int j = 0; j < 5; j++) {
                double r2 = x * x + y * y;
                double icdist = (1) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
                if (icdist < 0) {
                    x = (u - cx) * ifx;
                    y = (v - cy) * ify;
                    break;
                }
                double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
                double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
                x = (x0 - deltaX) * icdist;
                y = (y0 - deltaY) * icdist;
            }
        }
        dst[2 * i] = x;
        dst[2 * i + 1] = y;
    }
}

/* fundam.cpp HomographyEstimatorCallback::runKernel (inputs are converted to float by findHomography) */
static int homography_dlt(const double *Mxy, const double *mn, int count, double H[9])
{
    float Mf[64 * 2], mf[64 * 2];
    for (int i = 0; i < 2 * count; i++) {
        Mf[i] = (float)Mxy[i];
        mf[i] = (float)mn[i];
    }
    double LtL[81], W[9], V[81];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) {
        cmx += mf[2 * i];
        cmy += mf[2 * i + 1];
        cMx += Mf[2 * i];
        cMy += Mf[2 * i + 1];
    }
    cmx /= count;
    cmy /= count;
    cMx /= count;
    cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += fabs(mf[2 * i] - cmx);
        smy += fabs(mf[2 * i + 1] - cmy);
        sMx += fabs(Mf[2 * i] - cMx);
        sMy += fabs(Mf[2 * i + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = count / smx;
    smy = count / smy;
    sMx = count / sMx;
    sMy = count / sMy;
    double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < count; i++) {
        double x = (mf[2 * i] - cmx) * smx, y = (mf[2 * i + 1] - cmy) * smy;
        double X = (Mf[2 * i] - cMx) * sMx, Y = (Mf[2 * i + 1] - cMy) * sMy;
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    jacobi_eigen(LtL, 9, W, V);
    double H0[9], T[9];
    memcpy(H0, V + 8 * 9, sizeof(H0));
    mat3_mul(invHnorm, H0, T);
    mat3_mul(T, Hnorm2, H0);
    if (H0[8] == 0) return 0;
    double sc = 1. / H0[8];
    for (int i = 0; i < 9; i++) H[i] = H0[i] * sc;
    return 1;
}

/* compat_ptsetreg.cpp CvLevMarq (6 params, 2n errors, TermCriteria(EPS+ITER, 20, FLT_EPSILON)) */
typedef struct {
    double param[6], prevParam[6];
    double JtJ[36], JtErr[6];
    double prevErrNorm, errNorm;
    int lambdaLg10, iters, state; /* 0 DONE, 1 STARTED, 2 CALC_J, 3 CHECK_ERR */
} levmarq;

static void lm_step(levmarq *s)
{
    const double LOG10 = log(10.);
    double lambda = exp(s->lambdaLg10 * LOG10);
    double A[36], w[6], V[36];
    memcpy(A, s->JtJ, sizeof(A));
    for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
    /* solve(JtJN, JtErr, x, DECOMP_SVD): symmetric PSD -> eigen-decomposition, SVBkSb threshold */
    jacobi_eigen(A, 6, w, V);
    double thr = 0;
    for (int i = 0; i < 6; i++) thr += fabs(w[i]);
    thr *= DBL_EPSILON * 2;
    double x[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 6; k++) {
        if (fabs(w[k]) <= thr) continue;
        double d = 0;
        for (int i = 0; i < 6; i++) d += V[k * 6 + i] * s->JtErr[i];
        d /= w[k];
        for (int i = 0; i < 6; i++) x[i] += V[k * 6 + i] * d;
    }
    for (int i = 0; i < 6; i++) s->param[i] = s->prevParam[i] - x[i];
}

static double norm_l2(const double *v, int n)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

/* calibration.cpp cvFindExtrinsicCameraParams2, useExtrinsicGuess = 0 */
static int solve_pnp_core(const double K[9], const double D[5], const double *M, const double *m, int count, double rvec[3], double tvec[3]);

int ora_solve_pnp(const double K[9], const double D[5], const float *obj3, const float *img2, int count,
                  double rvec[3], double tvec[3])
{
    if (count < 4 || count > 32) return -1;
    double M[32 * 3], m[32 * 2];
    for (int i = 0; i < 3 * count; i++) M[i] = obj3[i];
    for (int i = 0; i < 2 * count; i++) m[i] = img2[i];
    return solve_pnp_core(K, D, M, m, count, rvec, tvec);
}

/* the same with double-precision points: Common::solvePnpSingle of stag_detect hands solvePnP vector<Point3d> / vector<Point2d>
 * (stag_ros/common.hpp:34-46).  For count > 4 OpenCV's findHomography additionally polishes the DLT homography with a few
 * Levenberg-Marquardt steps before it is decomposed; that polish of the STARTING point is not restated here ("parity unpinned"
 * for count > 4) -- the extrinsic refinement below starts next to the same minimum either way. */
int ora_solve_pnp_d(const double K[9], const double D[5], const double *obj3, const double *img2, int count, double rvec[3],
                    double tvec[3])
{
    if (count < 4 || count > 32) return -1;
    return solve_pnp_core(K, D, obj3, img2, count, rvec, tvec);
}

static int solve_pnp_core(const double K[9], const double D[5], const double *M, const double *m, int count, double rvec[3], double tvec[3])
{
    const int max_iter = 20;
    double mn[32 * 2], Mxy[32 * 2];
    undistort_points(m, count, K, D, mn);

    double Mc[3] = {0, 0, 0}, MM[9] = {0}, W[3], V[9], R[9], param[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < count; i++)
        for (int j = 0; j < 3; j++) Mc[j] += M[i * 3 + j];
    for (int j = 0; j < 3; j++) Mc[j] /= count; /* cvAvg */
    for (int i = 0; i < count; i++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) MM[a * 3 + b] += (M[i * 3 + a] - Mc[a]) * (M[i * 3 + b] - Mc[b]);
    {
        double A[9];
        memcpy(A, MM, sizeof(A));
        jacobi_eigen(A, 3, W, V); /* SVD of a symmetric PSD matrix; V rows = Vt */
    }
    if (!(W[2] / W[1] < 1e-3)) return -4; /* non-planar DLT branch: not on this path */
    {
        double tt[3], h[9], h1_norm, h2_norm;
        double *Rt = V;
        if (V[2] * V[2] + V[5] * V[5] < 1e-10) {
            for (int i = 0; i < 9; i++) Rt[i] = (i % 4 == 0) ? 1. : 0.;
        }
        double det = Rt[0] * (Rt[4] * Rt[8] - Rt[5] * Rt[7]) - Rt[1] * (Rt[3] * Rt[8] - Rt[5] * Rt[6]) +
                     Rt[2] * (Rt[3] * Rt[7] - Rt[4] * Rt[6]);
        if (det < 0)
            for (int i = 0; i < 9; i++) Rt[i] = -Rt[i];
        for (int i = 0; i < 3; i++) tt[i] = -(Rt[i * 3] * Mc[0] + Rt[i * 3 + 1] * Mc[1] + Rt[i * 3 + 2] * Mc[2]);
        for (int i = 0; i < count; i++) {
            const double *src = M + i * 3;
            Mxy[2 * i] = Rt[0] * src[0] + Rt[1] * src[1] + Rt[2] * src[2] + tt[0];
            Mxy[2 * i + 1] = Rt[3] * src[0] + Rt[4] * src[1] + Rt[5] * src[2] + tt[1];
        }
        if (homography_dlt(Mxy, mn, count, h)) {
            h1_norm = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
            h2_norm = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            double s1 = 1. / fmax(h1_norm, DBL_EPSILON), s2 = 1. / fmax(h2_norm, DBL_EPSILON);
            double st = 2. / fmax(h1_norm + h2_norm, DBL_EPSILON);
            double t3[3] = {h[2] * st, h[5] * st, h[8] * st};
            h[0] *= s1;
            h[3] *= s1;
            h[6] *= s1;
            h[1] *= s2;
            h[4] *= s2;
            h[7] *= s2;
            h[2] = h[3] * h[7] - h[6] * h[4];
            h[5] = h[6] * h[1] - h[0] * h[7];
            h[8] = h[0] * h[4] - h[3] * h[1];
            double rtmp[3];
            rodrigues_m2v(h, rtmp);
            rodrigues_v2m(rtmp, h, 0);
            for (int i = 0; i < 3; i++)
                param[3 + i] = h[i * 3] * tt[0] + h[i * 3 + 1] * tt[1] + h[i * 3 + 2] * tt[2] + t3[i];
            mat3_mul(h, Rt, R);
        } else {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
            param[3] = param[4] = param[5] = 0;
        }
        rodrigues_m2v(R, param);
    }

    /* refine extrinsic parameters using the iterative algorithm */
    levmarq s;
    memset(&s, 0, sizeof(s));
    memcpy(s.param, param, sizeof(param));
    s.lambdaLg10 = -3;
    s.state = 1;
    s.iters = 0;
    double J[64 * 6], err[64], dpdr[64 * 3], dpdt[64 * 3], proj[64];
    const int nerr = 2 * count;
    for (;;) {
        int needJ = 0, needErr = 0, proceed = 1;
        /* CvLevMarq::update */
        if (s.state == 0) {
            proceed = 0;
        } else if (s.state == 1) {
            needJ = needErr = 1;
            s.state = 2;
        } else if (s.state == 2) {
            for (int a = 0; a < 6; a++) {
                for (int b = 0; b < 6; b++) {
                    double acc = 0;
                    for (int k = 0; k < nerr; k++) acc += J[k * 6 + a] * J[k * 6 + b];
                    s.JtJ[a * 6 + b] = acc;
                }
                double acc = 0;
                for (int k = 0; k < nerr; k++) acc += J[k * 6 + a] * err[k];
                s.JtErr[a] = acc;
            }
            memcpy(s.prevParam, s.param, sizeof(s.param));
            lm_step(&s);
            if (s.iters == 0) s.prevErrNorm = norm_l2(err, nerr);
            needErr = 1;
            s.state = 3;
        } else {
            s.errNorm = norm_l2(err, nerr);
            int retry = 0;
            if (s.errNorm > s.prevErrNorm) {
                if (++s.lambdaLg10 <= 16) {
                    lm_step(&s);
                    needErr = 1;
                    s.state = 3;
                    retry = 1;
                }
            }
            if (!retry) {
                s.lambdaLg10 = s.lambdaLg10 - 1 > -16 ? s.lambdaLg10 - 1 : -16;
                double dn = 0, pn = 0;
                for (int i = 0; i < 6; i++) {
                    dn += (s.param[i] - s.prevParam[i]) * (s.param[i] - s.prevParam[i]);
                    pn += s.prevParam[i] * s.prevParam[i];
                }
                /* cvNorm(param, prevParam, CV_RELATIVE_L2) = |param - prev| / |prev| */
                double rel = sqrt(dn) / (sqrt(pn) + DBL_EPSILON);
                if (++s.iters >= max_iter || rel < FLT_EPSILON) {
                    s.state = 0;
                    /* update() returns true with _err = 0 -> caller breaks */
                    break;
                }
                s.prevErrNorm = s.errNorm;
                needJ = needErr = 1;
                s.state = 2;
            }
        }
        if (!proceed || !needErr) break;
        project_points(M, count, s.param, s.param + 3, K, D, proj, needJ ? dpdr : 0, needJ ? dpdt : 0);
        for (int k = 0; k < nerr; k++) err[k] = proj[k] - m[k];
        if (needJ)
            for (int k = 0; k < nerr; k++)
                for (int j = 0; j < 3; j++) {
                    J[k * 6 + j] = dpdr[k * 3 + j];
                    J[k * 6 + 3 + j] = dpdt[k * 3 + j];
                }
    }
    for (int i = 0; i < 3; i++) {
        rvec[i] = s.param[i];
        tvec[i] = s.param[3 + i];
    }
    return 0;
}

/* aruco_detect.cpp:151-161 getSingleMarkerObjectPoints + :247 solvePnP + :203-221 getReprojectionError */
int ora_solve_pnp_square(const double K[9], const double D[5], const float corners[8], double marker_len,
                         double rvec[3], double tvec[3], double *reproj_err)
{
    float ml = (float)marker_len; /* estimatePoseSingleMarkers takes float markerLength... */
    /* ...but fiducialSize is a double and getSingleMarkerObjectPoints(float) narrows it again */
    float obj[12] = {-ml / 2.f, ml / 2.f, 0, ml / 2.f, ml / 2.f, 0, ml / 2.f, -ml / 2.f, 0, -ml / 2.f, -ml / 2.f, 0};
    int rc = ora_solve_pnp(K, D, obj, corners, 4, rvec, tvec);
    if (rc) return rc;
    if (reproj_err) {
        double pr[8];
        ora_project_points(K, D, rvec, tvec, obj, 4, pr);
        double total = 0;
        for (int i = 0; i < 4; i++) {
            /* projectedPoints is vector<Point2f>: projections are rounded to float */
            double x1 = corners[2 * i], y1 = corners[2 * i + 1];
            double x2 = (float)pr[2 * i], y2 = (float)pr[2 * i + 1];
            double dx = x1 - x2, dy = y1 - y2;
            double e = sqrt(dx * dx + dy * dy);
            total += e * e;
        }
        *reproj_err = total / 4.0;
    }
    return 0;
}
