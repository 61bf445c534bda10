// fid_stag_pose.hip -- STag rows s9, s10: pose refinement (ellipse fit + simplex search) and the 5-point marker pose.
// Part of the fid_stag.hip translation unit (included there; not compiled on its own).
// ------------------------------------------------------------------------------------------------ K16: pose refinement
// PoseRefiner::refineMarkerPose (PoseRefiner.cpp:12-190), one wave per marker:
//   (1) pick the edge segment that is the image of the marker's circular border: a closed loop of >= 20 pixels inside the
//       quad whose back-projection stays within 0.1 of the circle of radius 0.4 and whose distances to 36 points of that circle
//       sum to < 1.8 (lanes take pixels; minima are order-free, the sum over the 36 points runs in order);
//   (2) fit an ellipse to it: customEllipse(pix*, n) (Ellipse.cpp:296-473) = Fitzgibbon's direct least squares through the
//       reference's own small linear algebra (scatter matrix summed in pixel order -- one lane per matrix entry --, choldc,
//       Gauss-Jordan inverse, Jacobi eigenvalues, all 1-based like the original);
//   (3) move the 9 entries of H with Nelder-Mead so that H^T C H is the circle (0.5, 0.5, r 0.4): cv::DownhillSolver with its
//       defaults, restated (see oracle/stag_ref.cpp for the same restatement on the checker's side); cost Refine::calc (:224-258);
//   (4) corners and centre from the new H.
// The reference takes atan / sin / cos of the ellipse's rotation from glibc; here the sine and cosine come from the tangent
// algebraically (sr_conic_to_ellipse): the results agree to rounding, the optimiser then follows a path that can differ in the
// last bits -- this row's parity bar is a tolerance (corners to 1e-3 px), not equality.
struct SrEllipse {
    double A1, B1, C1, D1, E1, F1, cX, cY, a, b;
};

// the conic -> ellipse conversion shared by both customEllipse constructors (Ellipse.cpp:394-454, :668-728); coefficients
// come in unnormalised
__device__ void sr_conic_to_ellipse(double A1, double B1, double C1, double D1, double E1, double F1, SrEllipse *e)
{
    B1 /= A1; C1 /= A1; D1 /= A1; E1 /= A1; F1 /= A1; A1 /= A1;
    double A2, C2, D2, E2, F2, sr = 0, cr = 1;  // (the reference leaves rotation unset when B1 == 0)
    bool rotated = false;
    if (B1 == 0) {
        A2 = A1; C2 = C1; D2 = D1; E2 = E1; F2 = F1;
    } else {
        // rotation = atan(t) / 2 with t = B1 / (A1 - C1); the reference takes cos / sin of 2 * rotation and of rotation from libm.
        // Here they come from t itself: cos(atan t) = 1 / sqrt(1 + t^2), sin(atan t) = t / sqrt(1 + t^2), and the half angle
        // (|rotation| <= pi / 4: its cosine is positive) cos r = sqrt((1 + cos 2r) / 2), sin r = sin 2r / (2 cos r) -- two square
        // roots and three divisions instead of an arctangent and two sincos calls (~ 380 of the ~ 950 dependent f64
        // instructions of one cost evaluation; the simplex search is a chain of a few hundred of them).  Equal to the libm road
        // to rounding, like the device's atan / sin / cos were: this row's parity bar is the tolerance stated above.
        const double t = B1 / (A1 - C1);
        double s2, c2;
        if (fabs(t) > 1e150) {  // A1 == C1 (t infinite: 2 r = +- pi / 2), or t * t would overflow
            c2 = 0.0;
            s2 = copysign(1.0, t);
        } else {
            c2 = 1.0 / sqrt(1.0 + t * t);
            s2 = t * c2;
        }
        cr = sqrt(0.5 * (1.0 + c2));
        sr = s2 / (2.0 * cr);
        rotated = t != 0.0;  // (rotation != 0)
        A2 = 0.5 * (A1 * (1 + c2 + B1 * s2 + C1 * (1 - c2)));
        C2 = 0.5 * (A1 * (1 - c2 - B1 * s2 + C1 * (1 + c2)));
        D2 = D1 * cr + E1 * sr;
        E2 = -D1 * sr + E1 * cr;
        F2 = F1;
    }
    const double D3 = D2 / A2, E3 = E2 / C2;
    double cX = -(D3 / 2), cY = -(E3 / 2);
    const double F3 = A2 * (cX * cX) + C2 * (cY * cY) - F2;
    e->a = sqrt(F3 / A2);
    e->b = sqrt(F3 / C2);
    if (rotated) {
        const double tx = cX, ty = cY;
        cX = tx * cr - ty * sr;
        cY = tx * sr + ty * cr;
    }
    e->cX = cX; e->cY = cY;
    e->A1 = A1; e->B1 = B1; e->C1 = C1; e->D1 = D1; e->E1 = E1; e->F1 = F1;
}

// 1-based 7 x 7 scratch matrices as in the reference
typedef double SrM[7][7];

__device__ void sr_jacobi(SrM a, double d[7], SrM v)
{
    const int n = 6;
    double b[7], z[7];
    for (int ip = 1; ip <= n; ip++) {
        for (int iq = 1; iq <= n; iq++) v[ip][iq] = 0.0;
        v[ip][ip] = 1.0;
    }
    for (int ip = 1; ip <= n; ip++) {
        b[ip] = d[ip] = a[ip][ip];
        z[ip] = 0.0;
    }
    auto rot = [](SrM m, int i, int j, int k, int l, double tau, double s) {
        const double g = m[i][j], h = m[k][l];
        m[i][j] = g - s * (h + g * tau);
        m[k][l] = h + s * (g - h * tau);
    };
    for (int i = 1; i <= 50; i++) {
        double sm = 0.0;
        for (int ip = 1; ip <= n - 1; ip++)
            for (int iq = ip + 1; iq <= n; iq++) sm += fabs(a[ip][iq]);
        if (sm == 0.0) return;
        const double tresh = i < 4 ? 0.2 * sm / (n * n) : 0.0;
        for (int ip = 1; ip <= n - 1; ip++) {
            for (int iq = ip + 1; iq <= n; iq++) {
                const double g = 100.0 * fabs(a[ip][iq]);
                if (i > 4 && g == 0.0) a[ip][iq] = 0.0;
                else if (fabs(a[ip][iq]) > tresh) {
                    double h = d[iq] - d[ip], t;
                    if (g == 0.0) t = (a[ip][iq]) / h;
                    else {
                        const double theta = 0.5 * h / (a[ip][iq]);
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    const double c = 1.0 / sqrt(1 + t * t), sn = t * c, tau = sn / (1.0 + c);
                    h = t * a[ip][iq];
                    z[ip] -= h; z[iq] += h; d[ip] -= h; d[iq] += h;
                    a[ip][iq] = 0.0;
                    for (int j = 1; j <= ip - 1; j++) rot(a, j, ip, j, iq, tau, sn);
                    for (int j = ip + 1; j <= iq - 1; j++) rot(a, ip, j, j, iq, tau, sn);
                    for (int j = iq + 1; j <= n; j++) rot(a, ip, j, iq, j, tau, sn);
                    for (int j = 1; j <= n; j++) rot(v, j, ip, j, iq, tau, sn);
                }
            }
        }
        for (int ip = 1; ip <= n; ip++) {
            b[ip] += z[ip];
            d[ip] = b[ip];
            z[ip] = 0.0;
        }
    }
}

// customEllipse(pix*, n) from the scatter matrix S (1-based, full) on: returns false if the inverse fails
__device__ bool sr_fit_from_scatter(SrM S, SrEllipse *e)
{
    const int n = 6;
    SrM L, invL, temp, C, V, sol, Const;
    double d[7], p[7];
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) L[i][j] = invL[i][j] = temp[i][j] = C[i][j] = V[i][j] = sol[i][j] = Const[i][j] = 0.0;
    for (int i = 0; i < 7; i++) d[i] = p[i] = 0.0;
    Const[1][3] = -2; Const[2][2] = 1; Const[3][1] = -2;  // FPF mode
    // choldc
    for (int i = 1; i <= n; i++) {
        for (int j = i; j <= n; j++) {
            double sum = S[i][j];
            for (int k = i - 1; k >= 1; k--) sum -= S[i][k] * S[j][k];
            if (i == j) {
                if (sum > 0.0) p[i] = sqrt(sum);
            } else
                S[j][i] = sum / p[i];
        }
    }
    for (int i = 1; i <= n; i++)
        for (int j = i; j <= n; j++) {
            if (i == j) L[i][i] = p[i];
            else {
                L[j][i] = S[j][i];
                L[i][j] = 0.0;
            }
        }
    // inverse(L): Gauss-Jordan with row pivoting on [L | I]
    {
        double A[7][14];
        for (int k = 0; k < 7; k++)
            for (int j = 0; j < 14; j++) A[k][j] = 0.0;
        for (int k = 1; k <= n; k++) {
            for (int j = 1; j <= n; j++) A[k][j] = L[k][j];  // (column n + 1 stays 0 as in the reference)
            A[k][k - 1 + n + 2] = 1;
        }
        for (int k = 1; k <= n; k++) {
            double maxpivot = fabs(A[k][k]);
            int npivot = k;
            for (int i = k; i <= n; i++)
                if (maxpivot < fabs(A[i][k])) {
                    maxpivot = fabs(A[i][k]);
                    npivot = i;
                }
            if (!(maxpivot >= 10e-20)) return false;
            if (npivot != k)
                for (int j = k; j <= 2 * n + 1; j++) {
                    const double t = A[npivot][j];
                    A[npivot][j] = A[k][j];
                    A[k][j] = t;
                }
            const double Dv = A[k][k];
            for (int j = 2 * n + 1; j >= k; j--) A[k][j] = A[k][j] / Dv;
            for (int i = 1; i <= n; i++)
                if (i != k) {
                    const double mult = A[i][k];
                    for (int j = 2 * n + 1; j >= k; j--) A[i][j] = A[i][j] - mult * A[k][j];
                }
        }
        for (int k = 1; k <= n; k++)
            for (int j = n + 2, q = 1; j <= 2 * n + 1; j++, q++) invL[k][q] = A[k][j];
    }
    // temp = Const * invL^T, C = invL * temp
    for (int pp = 1; pp <= n; pp++)
        for (int q = 1; q <= n; q++) {
            temp[pp][q] = 0.0;
            for (int l = 1; l <= n; l++) temp[pp][q] = temp[pp][q] + Const[pp][l] * invL[q][l];
        }
    for (int pp = 1; pp <= n; pp++)
        for (int q = 1; q <= n; q++) {
            C[pp][q] = 0.0;
            for (int l = 1; l <= n; l++) C[pp][q] = C[pp][q] + invL[pp][l] * temp[l][q];
        }
    sr_jacobi(C, d, V);
    // sol = invL^T * V
    for (int pp = 1; pp <= n; pp++)
        for (int q = 1; q <= n; q++) {
            sol[pp][q] = 0.0;
            for (int l = 1; l <= n; l++) sol[pp][q] = sol[pp][q] + invL[l][pp] * V[l][q];
        }
    for (int j = 1; j <= n; j++) {
        double mod = 0.0;
        for (int i = 1; i <= n; i++) mod += sol[i][j] * sol[i][j];
        for (int i = 1; i <= n; i++) sol[i][j] /= sqrt(mod);
    }
    int solind = 0;
    for (int i = 1; i <= n; i++)
        if (d[i] < 0 && fabs(d[i]) > 10e-20) solind = i;
    sr_conic_to_ellipse(sol[1][solind], sol[2][solind], sol[3][solind], sol[4][solind], sol[5][solind], sol[6][solind], e);
    return true;
}

__device__ void sr_mul3(const double a[9], const double b[9], double d[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) d[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

// cv::Mat::inv() of a 3 x 3 (closed form, as OpenCV's invert() does for n <= 3)
__device__ void sr_inv3(const double S[9], double D[9])
{
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    for (int k = 0; k < 9; k++) D[k] = 0.0;
    if (d != 0.) {
        d = 1. / d;
        D[0] = (S[4] * S[8] - S[5] * S[7]) * d; D[1] = (S[2] * S[7] - S[1] * S[8]) * d; D[2] = (S[1] * S[5] - S[2] * S[4]) * d;
        D[3] = (S[5] * S[6] - S[3] * S[8]) * d; D[4] = (S[0] * S[8] - S[2] * S[6]) * d; D[5] = (S[2] * S[3] - S[0] * S[5]) * d;
        D[6] = (S[3] * S[7] - S[4] * S[6]) * d; D[7] = (S[1] * S[6] - S[0] * S[7]) * d; D[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    }
}

// Refine::calc (PoseRefiner.cpp:224-258): x[i + 3 j] = H(i, j)
__device__ double sr_cost(const double x[9], const double Cm[9])
{
    double H[9], HT[9], T[9], P[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            H[3 * i + j] = x[i + j * 3];
            HT[3 * j + i] = x[i + j * 3];
        }
    sr_mul3(HT, Cm, T);
    sr_mul3(T, H, P);
    SrEllipse e;
    sr_conic_to_ellipse(P[0], -P[1] * 2, P[4], P[2] * 2, -P[5] * 2, P[8], &e);
    double acc = 0;
    acc += fabs(e.a - 0.4);
    acc += fabs(e.b - 0.4);
    acc += fabs(e.cX - 0.5);
    acc += fabs(-e.cY - 0.5);
    return acc;
}

// Refine::calc for FOUR points at once, one per 16-lane row of the wave (round 4).  One evaluation is ~900 dependent f64
// instructions when a lane does it alone, and the solver's three candidates of an iteration kept three lanes of 64 busy: the
// wave issued the whole stream for them (4 300 cycles per iteration, 4.8 cycles per instruction -- the single-wave issue rate
// tools/valu_calib.hip measures).  Here a ROW works on one point:
//   lane (a, b) = 4 a + b of the row (a, b < 3) forms T[a][b] = sum_k H[k][a] C[k][b], takes T[a][0..2] from its quad by DPP
//   quad_perm and forms P[a][b] = sum_k T[a][k] H[k][b] -- the two 3 x 3 products in 2 x 5 instructions instead of 2 x 45;
//   the lanes that hold P00, P01, P11, P02, P12, P22 make the conic coefficients and divide by A1 in ONE division; the rest of
//   Ellipse.cpp's conversion runs on lane pairs: one lane the A / D / x / a side, its neighbour the C / E / y / b side (the two
//   sides are the same expressions with other operands and signs: D3 = D2 / A2 beside E3 = E2 / C2, sqrt(F3 / A2) beside
//   sqrt(F3 / C2)), exchanging by quad_perm.
// Every quantity is formed by the same operations on the same operands, in the same order, as sr_cost (a - b is taken as
// a + (-b), which is the same IEEE result); the cost comes back in every lane of the row.  -DSR_COST_SCALAR keeps sr_cost.
__device__ __forceinline__ double sr_row_from(double v, int lane_in_row, int lane)
{
    return shfl_f64(v, (lane & 48) | lane_in_row);
}
__device__ double sr_cost_rows(const double xa[3], const double xb[3], const double cmb[3], int lane)
{
    const int r = lane & 15, a = r >> 2, b = r & 3;
    // T[a][b], then the row a of T from the quad, then P[a][b]  (sr_mul3's expression: a0 b0 + a1 b1 + a2 b2)
    const double T = xa[0] * cmb[0] + xa[1] * cmb[1] + xa[2] * cmb[2];
    const double t0 = dpp_f64<0x00>(T), t1 = dpp_f64<0x55>(T), t2 = dpp_f64<0xAA>(T);
    const double Pab = t0 * xb[0] + t1 * xb[1] + t2 * xb[2];
    // the conic of sr_conic_to_ellipse(P0, -2 P1, P4, 2 P2, -2 P5, P8): this lane's coefficient, unnormalised
    double coef = Pab;                       // A1 (lane 0), C1 (lane 5), F1 (lane 10)
    if (r == 1 || r == 6) coef = -Pab * 2;   // B1 = -P[1] * 2 (lane 1), E1 = -P[5] * 2 (lane 6)
    if (r == 2) coef = Pab * 2;              // D1 = P[2] * 2
    const double A1raw = sr_row_from(coef, 0, lane);
    const double q = coef / A1raw;           // B1 /= A1 ... F1 /= A1, A1 /= A1: one division for all six
    const double A1 = sr_row_from(q, 0, lane), B1 = sr_row_from(q, 1, lane), C1 = sr_row_from(q, 5, lane), D1 = sr_row_from(q, 2, lane),
                 E1 = sr_row_from(q, 6, lane), F1 = sr_row_from(q, 10, lane);
    const bool side = (r & 1) != 0;          // false: the A2 / D / x / a side, true: the C2 / E / y / b side
    double V2, W2, F2, sr = 0, cr = 1;       // V2 = A2 | C2, W2 = D2 | E2
    bool rotated = false;
    if (B1 == 0) {
        V2 = side ? C1 : A1;
        W2 = side ? E1 : D1;
        F2 = F1;
    } else {
        const double t = B1 / (A1 - C1);
        double s2, c2;
        if (fabs(t) > 1e150) {
            c2 = 0.0;
            s2 = copysign(1.0, t);
        } else {
            c2 = 1.0 / sqrt(1.0 + t * t);
            s2 = t * c2;
        }
        cr = sqrt(0.5 * (1.0 + c2));
        sr = s2 / (2.0 * cr);
        rotated = t != 0.0;
        // A2 = 0.5 (A1 (1 + c2 + B1 s2 + C1 (1 - c2))), C2 = 0.5 (A1 (1 - c2 - B1 s2 + C1 (1 + c2)))
        const double sc2 = side ? -c2 : c2, bs = B1 * s2, sbs = side ? -bs : bs;
        V2 = 0.5 * (A1 * (1 + sc2 + sbs + C1 * (1 - sc2)));
        // D2 = D1 cr + E1 sr, E2 = -D1 sr + E1 cr
        W2 = (side ? -D1 : D1) * (side ? sr : cr) + E1 * (side ? cr : sr);
        F2 = F1;
    }
    const double W3 = W2 / V2;               // D3 = D2 / A2 beside E3 = E2 / C2
    double cc = -(W3 / 2);                   // cX | cY
    const double term = V2 * (cc * cc);      // A2 cX^2 | C2 cY^2
    const double oterm = dpp_f64<0xB1>(term);
    const double F3 = (side ? oterm + term : term + oterm) - F2;
    const double ax = sqrt(F3 / V2);         // a | b
    if (rotated) {
        const double occ = dpp_f64<0xB1>(cc);
        // cX = tx cr - ty sr (this lane holds tx), cY = tx sr + ty cr (this lane holds ty)
        cc = side ? occ * sr + cc * cr : cc * cr - occ * sr;
    }
    const double e_ax = fabs(ax - 0.4);                               // |a - 0.4| | |b - 0.4|
    const double e_c = side ? fabs(-cc - 0.5) : fabs(cc - 0.5);       // |cX - 0.5| | |-cY - 0.5|
    const double o_ax = dpp_f64<0xB1>(e_ax), o_c = dpp_f64<0xB1>(e_c);
    double acc = 0;
    acc += side ? o_ax : e_ax;
    acc += side ? e_ax : o_ax;
    acc += side ? o_c : e_c;
    acc += side ? e_c : o_c;
    return sr_row_from(acc, 0, lane);
}

// cv::DownhillSolver::minimize with the default TermCriteria(MAX_ITER + EPS, 5000, 1e-6), 9 dimensions, run by one wave with
// the simplex in LDS.  The three candidate points of an iteration -- reflection (-1), expansion (-2), contraction (0.5) --
// depend only on the current simplex: lanes 0, 1, 2 evaluate them side by side and the solver's decision sequence then
// picks what it would have evaluated one after the other (the evaluation counter advances as in the sequential solver).
// The ten vertices of the start simplex and of a shrink step are evaluated by ten lanes.  Same arithmetic per point as the
// sequential form (column sums in vertex order, the same alpha / beta expressions).
// wave-wide min / max of doubles on DPP row operations (fid_kernels.hip: FID_DPP_SCAN_STEPS), the result in every lane
__device__ __forceinline__ double sr_wave_min_f64(double v)
{
    unsigned long long u = __double_as_longlong(v);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
#define S_(C, M)                                                                                              \
    {                                                                                                         \
        const unsigned ol = (unsigned)FID_DPP(0, (int)lo, C, M), oh = (unsigned)FID_DPP(0x7ff00000, (int)hi, C, M); \
        const double o = __longlong_as_double(((unsigned long long)oh << 32) | ol);                           \
        const double m = fmin(__longlong_as_double(((unsigned long long)hi << 32) | lo), o);                  \
        const unsigned long long mu = __double_as_longlong(m);                                                \
        lo = (unsigned)mu;                                                                                    \
        hi = (unsigned)(mu >> 32);                                                                            \
    }
    FID_DPP_SCAN_STEPS(S_)  // (+inf fills the lanes a step does not reach)
#undef S_
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double sr_wave_max_nonneg_f64(double v)  // v >= 0 in every lane
{
    unsigned long long u = __double_as_longlong(v);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
#define S_(C, M)                                                                                    \
    {                                                                                               \
        const unsigned ol = (unsigned)FID_DPP(0, (int)lo, C, M), oh = (unsigned)FID_DPP(0, (int)hi, C, M); \
        const double o = __longlong_as_double(((unsigned long long)oh << 32) | ol);                 \
        const double m = fmax(__longlong_as_double(((unsigned long long)hi << 32) | lo), o);        \
        const unsigned long long mu = __double_as_longlong(m);                                      \
        lo = (unsigned)mu;                                                                          \
        hi = (unsigned)(mu >> 32);                                                                  \
    }
    FID_DPP_SCAN_STEPS(S_)
#undef S_
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

struct SrSimplex {
    double p[10][9], y[10], sum[9];
};

#define SR_LDS_SYNC()                                          \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)

__device__ void sr_downhill(double x[9], const double step[9], const double Cm[9], int lane, SrSimplex *S, int *evals = nullptr)
{
    const int nd = 9;
    if (lane <= nd) {
        for (int j = 0; j < nd; j++) {
            double v = x[j];
            if (lane == 0) v -= 0.5 * step[j];
            else if (lane - 1 == j) v += 0.5 * step[j];
            S->p[lane][j] = v;
        }
    }
    SR_LDS_SYNC();
#ifdef SR_COST_SCALAR
    auto eval_rows = [&](int skip) {  // y[i] = f(p[i]) for every vertex but `skip`, one lane per vertex
        const int i = lane <= nd ? lane : nd;
        double row[9];
        for (int j = 0; j < nd; j++) row[j] = S->p[i][j];
        const double v = sr_cost(row, Cm);
        if (lane <= nd && lane != skip) S->y[lane] = v;
        SR_LDS_SYNC();
    };
#else
    // the lane's place in its row's two 3 x 3 products (sr_cost_rows) and the column b of C it multiplies with
    const int ra = ((lane & 15) >> 2) < 3 ? (lane & 15) >> 2 : 2, rb = (lane & 3) < 3 ? lane & 3 : 2;
    const double cmb[3] = {rb == 0 ? Cm[0] : rb == 1 ? Cm[1] : Cm[2], rb == 0 ? Cm[3] : rb == 1 ? Cm[4] : Cm[5],
                           rb == 0 ? Cm[6] : rb == 1 ? Cm[7] : Cm[8]};
    auto eval_rows = [&](int skip) {  // y[i] = f(p[i]) for every vertex but `skip`: four vertices at a time, a 16-lane row each
        for (int v0 = 0; v0 <= nd; v0 += 4) {
            const int i = v0 + (lane >> 4) <= nd ? v0 + (lane >> 4) : nd;
            double xa[3], xb[3];
            for (int k = 0; k < 3; k++) {
                xa[k] = S->p[i][3 * ra + k];
                xb[k] = S->p[i][3 * rb + k];
            }
            const double v = sr_cost_rows(xa, xb, cmb, lane);
            if ((lane & 15) == 0 && v0 + (lane >> 4) <= nd && i != skip) S->y[i] = v;
        }
        SR_LDS_SYNC();
    };
#endif
    // column sums in vertex order, and in the same pass over the column its extent |max - min| (the termination test's range,
    // round 5: it read the ten vertices a second time at the top of every iteration)
    double col_range = 0;  // (lanes >= nd carry 0)
    auto update_sum = [&]() {
        if (lane < nd) {
            double acc = 0., mn = INFINITY, mx = -INFINITY;
            for (int i = 0; i <= nd; i++) {
                const double v = S->p[i][lane];
                acc += v;
                mn = fmin(mn, v);
                mx = fmax(mx, v);
            }
            S->sum[lane] = acc;
            col_range = fabs(mx - mn);
        }
        SR_LDS_SYNC();
    };
    auto replace_point = [&](int ihi, double alpha_, double ytry) {
        // (alpha_ is one of -1, -2, 0.5: the quotient is one of three constants, no division per iteration)
        const double alpha = alpha_ == -1.0 ? (1.0 - -1.0) / 9 : alpha_ == -2.0 ? (1.0 - -2.0) / 9 : (1.0 - alpha_) / nd, beta = alpha - alpha_;
        // (a lane reads back only its own column: what it has just written is there without a synchronisation in between)
        if (lane < nd) S->p[ihi][lane] = S->sum[lane] * alpha - S->p[ihi][lane] * beta;
        if (lane == 0) S->y[ihi] = ytry;
        update_sum();
    };
    // the candidate a lane's row works on: reflection (-1), expansion (-2), contraction (0.5) -- its two coefficients, once
    const double row_a = (lane >> 4) == 0 ? -1.0 : (lane >> 4) == 1 ? -2.0 : 0.5;
    const double row_alpha = (1.0 - row_a) / nd, row_beta = row_alpha - row_a;
    int fcount = nd + 1;
    eval_rows(-1);
    update_sum();
    for (;;) {
        // (round 5) lowest, highest and next-highest vertex by RANK: lane i < 10 holds y[i] and counts the vertices above it; when
        // the highest, the second highest and the lowest value each occur once -- ten function values of ten different points:
        // always, bar a degenerate start -- the solver's scan below has exactly these three answers (strict comparisons find the
        // one maximum and the one second maximum whatever the order, `<=` the one minimum), and they cost ten compares in all
        // lanes at once instead of the scan's thirty compare-and-select chains (150 of an iteration's 1 000 instructions).
        int ilo = 0, ihi = 0, inhi = 0;
        double v_lo = 0, v_hi = 0, v_nhi = 0;
        bool ranked;
        {
            const double yl = S->y[lane < 10 ? lane : 9];
            int above = 0;
#pragma unroll
            for (int j = 0; j <= 9; j++) above += bcast_f64(yl, j) > yl ? 1 : 0;
            const unsigned m0 = (unsigned)__ballot(lane < 10 && above == 0), m1 = (unsigned)__ballot(lane < 10 && above == 1),
                           m9 = (unsigned)__ballot(lane < 10 && above == 9);
            ranked = __builtin_popcount(m0) == 1 && __builtin_popcount(m1) == 1 && __builtin_popcount(m9) == 1;
            if (ranked) {
                ihi = __builtin_ctz(m0); inhi = __builtin_ctz(m1); ilo = __builtin_ctz(m9);
                v_hi = bcast_f64(yl, ihi); v_nhi = bcast_f64(yl, inhi); v_lo = bcast_f64(yl, ilo);
            }
        }
        if (!ranked) {
        double y[10];
#pragma unroll
        for (int i = 0; i <= 9; i++) y[i] = S->y[i];
        // the solver's scan for the lowest, highest and next-highest vertex -- the same comparisons in the same order, with the
        // VALUES y[ilo], y[ihi], y[inhi] carried beside the indices: indexing the register array y[] with a run-time index costs
        // a ten-way select chain per read, thirty of them per scan (the scan was about half of an iteration's 5 700 cycles)
        ilo = 0;
        v_lo = y[0];
        if (y[0] > y[1]) { ihi = 0; inhi = 1; v_hi = y[0]; v_nhi = y[1]; } else { ihi = 1; inhi = 0; v_hi = y[1]; v_nhi = y[0]; }
#pragma unroll
        for (int i = 0; i <= 9; i++) {
            const double yv = y[i];
            if (yv <= v_lo) { ilo = i; v_lo = yv; }
            if (yv > v_hi) { inhi = ihi; v_nhi = v_hi; ihi = i; v_hi = yv; }
            else if (yv > v_nhi && i != ihi) { inhi = i; v_nhi = yv; }
        }
        if (ilo == inhi || ilo == ihi) {
            bool found = false;
#pragma unroll
            for (int i = 0; i <= 9; i++)
                if (!found && y[i] == v_lo && i != ihi && i != inhi) { ilo = i; found = true; }
        }
        }
        const double error = fabs(v_hi - v_lo);
        double range = 0;
        {
            const double r = col_range;
            // (lanes >= nd carry 0: the maximum of the first 16-lane row is the wave's -- four DPP steps instead of six)
            {
                unsigned long long u = __double_as_longlong(r);
                unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
#define S_(C)                                                                                       \
    {                                                                                               \
        const unsigned ol = (unsigned)FID_DPP(0, (int)lo, C, 0xf), oh = (unsigned)FID_DPP(0, (int)hi, C, 0xf); \
        const double o = __longlong_as_double(((unsigned long long)oh << 32) | ol);                 \
        const double m = fmax(__longlong_as_double(((unsigned long long)hi << 32) | lo), o);        \
        const unsigned long long mu = __double_as_longlong(m);                                      \
        lo = (unsigned)mu;                                                                          \
        hi = (unsigned)(mu >> 32);                                                                  \
    }
                S_(0x111) S_(0x112) S_(0x114) S_(0x118)  // row_shr:1, 2, 4, 8: lane 15 holds the maximum of lanes 0..15
#undef S_
                lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 15);
                hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 15);
                range = __longlong_as_double(((unsigned long long)hi << 32) | lo);
            }
        }
        if (range <= 0.000001 || error <= 0.000001 || fcount >= 5000) {
            for (int j = 0; j < nd; j++) x[j] = S->p[ilo][j];
            if (evals) *evals = fcount;
            return;
        }
        const double y_lo = v_lo, y_nhi = v_nhi, y_hi = v_hi;
#ifdef SR_COST_SCALAR
        double buf[9];
        {
            const double a_ = lane == 0 ? -1.0 : lane == 1 ? -2.0 : 0.5;
            const double alpha = (1.0 - a_) / nd, beta = alpha - a_;
            for (int j = 0; j < nd; j++) buf[j] = S->sum[j] * alpha - S->p[ihi][j] * beta;
        }
        const double yl = sr_cost(buf, Cm);
        const double y_refl = bcast_f64(yl, 0), y_exp = bcast_f64(yl, 1), y_con = bcast_f64(yl, 2);
#else
        // row 0: reflection (-1), row 1: expansion (-2), rows 2 and 3: contraction (0.5); a lane forms the six entries it multiplies
        double xa[3], xb[3];
        {
            const double alpha = row_alpha, beta = row_beta;
            for (int k = 0; k < 3; k++) {
                xa[k] = S->sum[3 * ra + k] * alpha - S->p[ihi][3 * ra + k] * beta;
                xb[k] = S->sum[3 * rb + k] * alpha - S->p[ihi][3 * rb + k] * beta;
            }
        }
        const double yl = sr_cost_rows(xa, xb, cmb, lane);
        const double y_refl = bcast_f64(yl, 0), y_exp = bcast_f64(yl, 16), y_con = bcast_f64(yl, 32);
#endif
        fcount++;
        double alpha = -1.0, y_alpha = y_refl;
        if (y_alpha < y_nhi) {
            if (y_alpha < y_lo) {
                fcount++;
                if (y_exp < y_alpha) { alpha = -2.0; y_alpha = y_exp; }
            }
            replace_point(ihi, alpha, y_alpha);
        } else {
            fcount++;
            if (y_con < y_hi) replace_point(ihi, 0.5, y_con);
            else {
                if (lane <= nd && lane != ilo)
                    for (int j = 0; j < nd; j++) S->p[lane][j] = 0.5 * (S->p[lane][j] + S->p[ilo][j]);
                SR_LDS_SYNC();
                eval_rows(ilo);
                fcount += nd;
                update_sum();
            }
        }
    }
}

// PoseRefiner::checkIfPointInQuad (PoseRefiner.cpp:200-222)
__device__ bool sr_in_quad(const double c[8], double px, double py)
{
    const double c1c2x = c[2] - c[0], c1c2y = c[3] - c[1], c1c4x = c[6] - c[0], c1c4y = c[7] - c[1];
    const double c3c2x = c[2] - c[4], c3c2y = c[3] - c[5], c3c4x = c[6] - c[4], c3c4y = c[7] - c[5];
    const double c1px = px - c[0], c1py = py - c[1], c3px = px - c[4], c3py = py - c[5];
    if (sq_cross(c1px, c1py, c1c2x, c1c2y) * sq_cross(c1px, c1py, c1c4x, c1c4y) >= 0) return false;
    if (sq_cross(c1c2x, c1c2y, c1px, c1py) * sq_cross(c1c2x, c1c2y, c1c4x, c1c4y) <= 0) return false;
    if (sq_cross(c3px, c3py, c3c2x, c3c2y) * sq_cross(c3px, c3py, c3c4x, c3c4y) >= 0) return false;
    if (sq_cross(c3c2x, c3c2y, c3px, c3py) * sq_cross(c3c2x, c3c2y, c3c4x, c3c4y) <= 0) return false;
    return true;
}

__device__ __forceinline__ void k_stag_refine_impl(fid_stag_marker *__restrict__ markers, const int *__restrict__ nmarkers,
                                                    const int2 *__restrict__ vsegs, const int *__restrict__ nsegs, const int2 *__restrict__ pix,
                                                    int *__restrict__ chosen_out)
{
    const int m = blockIdx.x, lane = threadIdx.x;
    if (m >= *nmarkers) return;
    fid_stag_marker M = markers[m];
    const double sinVals[36] = {0.000000,  0.173648,  0.342020,  0.500000,  0.642788,  0.766044,  0.866025,  0.939693,  0.984808,
                                1.000000,  0.984808,  0.939693,  0.866025,  0.766044,  0.642788,  0.500000,  0.342020,  0.173648,
                                0.000000,  -0.173648, -0.342020, -0.500000, -0.642788, -0.766044, -0.866025, -0.939693, -0.984808,
                                -1.000000, -0.984808, -0.939693, -0.866025, -0.766044, -0.642788, -0.500000, -0.342020, -0.173648};
    double Hinv[9];
    sr_inv3(M.H, Hinv);
#ifdef SR_TIMING
    const unsigned long long t_0 = __builtin_readcyclecounter();
    int d_cands = 0;
#endif
    // ---- (1) the edge segment of the circular border
    int chosen = -1;
    double minAcc = INFINITY;
    const int ns = *nsegs;
    for (int sg0 = 0; sg0 < ns; sg0 += 64) {
      // the cheap tests (length, closed loop, sampled pixels inside the quad) for 64 segments at once, one per lane; the
      // survivors are then examined one after the other, in segment order, by the whole wave
      unsigned long long cand;
      {
        const int sgl = sg0 + lane;
        bool ok = false;
        if (sgl < ns) {
            const int first = vsegs[sgl].x, n = vsegs[sgl].y;
            if (n >= 20) {
                const int2 *p = pix + first;
                if (!(sq_dist2((double)p[0].y, (double)p[0].x, (double)p[n - 1].y, (double)p[n - 1].x) > 7.0 * 7.0)) {
                    ok = true;
                    for (int k = 0; k < n && ok; k += 20)
                        if (!sr_in_quad(M.corners, (double)p[k].y, (double)p[k].x)) ok = false;
                }
            }
        }
        cand = __ballot(ok);
      }
      while (cand) {
        const int sg = sg0 + __builtin_ctzll(cand);
        cand &= cand - 1;
#ifdef SR_TIMING
        d_cands++;
#endif
        const int first = vsegs[sg].x, n = vsegs[sg].y;
        const int2 *p = pix + first;
        // back-projection; per sample point the minimum over the pixels, per pixel the minimum over the sample points
        bool bad = false;
        // (round 3) first only the per-pixel test that rejects: 22 of the 23 closed loops inside a marker are code dots whose
        // pixels lie nowhere near the circle -- they end here, after the first 64 pixels, without the 36 wave-wide minima per
        // chunk that only the accepted segment's error sum needs (same distances, same comparisons, same decision)
        for (int k0 = 0; k0 < n && !bad; k0 += 64) {
            const int k = k0 + lane;
            const bool act = k < n;
            double pixErr = INFINITY;
            if (act) {
                const double ex = p[k].y, ey = p[k].x;
                const double a0 = Hinv[0] * ex + Hinv[1] * ey + Hinv[2] * 1, a1 = Hinv[3] * ex + Hinv[4] * ey + Hinv[5] * 1;
                const double a2 = Hinv[6] * ex + Hinv[7] * ey + Hinv[8] * 1;
                const double qx = a0 / a2, qy = a1 / a2;
                // (round 4) min_s sqrt(x_s) == sqrt(min_s x_s) bit for bit -- the correctly rounded square root is monotone -- so the
                // rejecting pass takes ONE square root per pixel instead of 36 (23 code-dot loops per marker end in this pass)
                double m2 = INFINITY;
                for (int s = 0; s < 36; s++) {
                    const double sx = 0.5 + 0.4 * sinVals[(s + 9) % 36], sy = 0.5 + 0.4 * sinVals[s];
                    const double d2 = (qx - sx) * (qx - sx) + (qy - sy) * (qy - sy);
                    if (d2 < m2) m2 = d2;
                }
                pixErr = sqrt(m2);
            }
            if (__ballot(act && pixErr > 0.1)) bad = true;
        }
        if (bad) continue;
        double sampleErr[36];
        for (int s = 0; s < 36; s++) sampleErr[s] = INFINITY;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            double qx = 0, qy = 0;
            const bool act = k < n;
            if (act) {
                const double ex = p[k].y, ey = p[k].x;
                const double a0 = Hinv[0] * ex + Hinv[1] * ey + Hinv[2] * 1, a1 = Hinv[3] * ex + Hinv[4] * ey + Hinv[5] * 1;
                const double a2 = Hinv[6] * ex + Hinv[7] * ey + Hinv[8] * 1;
                qx = a0 / a2;
                qy = a1 / a2;
            }
            double pixErr = INFINITY;
            for (int s = 0; s < 36; s++) {
                const double sx = 0.5 + 0.4 * sinVals[(s + 9) % 36], sy = 0.5 + 0.4 * sinVals[s];
                const double d = act ? sqrt((qx - sx) * (qx - sx) + (qy - sy) * (qy - sy)) : INFINITY;
                if (d < pixErr) pixErr = d;
                const double mn = sr_wave_min_f64(d);
                if (mn < sampleErr[s]) sampleErr[s] = mn;
            }
            if (__ballot(act && pixErr > 0.1)) bad = true;
        }
        if (bad) continue;
        double errSum = 0;
        for (int s = 0; s < 36; s++) errSum += sampleErr[s];
        if (errSum < minAcc && errSum < 36 * 0.05) {
            minAcc = errSum;
            chosen = sg;
        }
      }
    }
    if (lane == 0) chosen_out[m] = chosen;
    if (chosen < 0) return;
#ifdef SR_TIMING
    const unsigned long long t_1 = __builtin_readcyclecounter();
#endif
    // ---- (2) ellipse through the chosen segment: scatter matrix, one lane per entry (p <= q), summed in pixel order
    __shared__ double s_S[7][7];
    {
        const int first = vsegs[chosen].x, n = vsegs[chosen].y;
        const int2 *p = pix + first;
        if (lane < 49) s_S[lane / 7][lane % 7] = 0.0;
        __builtin_amdgcn_wave_barrier();
        // (round 3) the pixels across the lanes, every lane the 21 running sums of its pixels, then 21 wave sums: the terms are
        // integers (pixel coordinates), so up to ~600 pixels every partial sum is exact in a double whatever the order; beyond
        // that the order changes the last bit of a sum that the ellipse fit and the simplex search, a tolerance row anyway
        // (1e-3 px against the reference), do not resolve.  (One lane per matrix entry walked all n pixels: 0.15 ms per marker.)
        double acc[21];
#pragma unroll
        for (int e = 0; e < 21; e++) acc[e] = 0.0;
        for (int l = lane; l < n; l += 64) {
            const double tx = (double)p[l].y, ty = (double)(-p[l].x);
            const double Dl[6] = {tx * tx, tx * ty, ty * ty, tx, ty, 1.0};
            int e = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = a; b < 6; b++) acc[e++] += Dl[a] * Dl[b];
        }
        {
            int e = 0;
#pragma unroll
            for (int a = 1; a <= 6; a++)
#pragma unroll
                for (int b = a; b <= 6; b++) {
                    double v = acc[e++];
                    v += shfl_xor_f64(v, 1);
                    v += shfl_xor_f64(v, 2);
                    v += shfl_xor_f64(v, 4);
                    v += shfl_xor_f64(v, 8);
                    v += shfl_xor_f64(v, 16);
                    v += shfl_xor_f64(v, 32);
                    if (lane == 0) {
                        s_S[a][b] = v;
                        s_S[b][a] = v;
                    }
                }
        }
        __builtin_amdgcn_wave_barrier();
        if (n < 6) return;
    }
    // from here on every lane carries the same values (the fit is small scalar work; the simplex search spreads its function
    // evaluations over the lanes)
    __shared__ SrSimplex s_simplex;
    SrM S;
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) S[i][j] = s_S[i][j];
    SrEllipse E;
    if (!sr_fit_from_scatter(S, &E)) return;
    double Cm[9];
    Cm[0] = E.A1; Cm[1] = Cm[3] = -E.B1 / 2; Cm[4] = E.C1; Cm[2] = Cm[6] = E.D1 / 2; Cm[5] = Cm[7] = -E.E1 / 2; Cm[8] = E.F1;
    // ---- (3) Nelder-Mead over the entries of H
    double x[9], step[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) x[i + j * 3] = M.H[3 * i + j];
    for (int k = 0; k < 9; k++) step[k] = fabs(0.001 * x[k]);
#ifdef SR_TIMING
    const unsigned long long t_2 = __builtin_readcyclecounter();
    int d_evals = 0;
    sr_downhill(x, step, Cm, lane, &s_simplex, &d_evals);
    if (lane == 0)
        printf("refine marker %d: segments %d, candidates examined %d, chosen length %d; ticks: search %llu, scatter + fit %llu, simplex %llu (%d evaluations)\n", m, ns,
               d_cands, vsegs[chosen].y, t_1 - t_0, t_2 - t_1, (unsigned long long)__builtin_readcyclecounter() - t_2, d_evals);
#else
    sr_downhill(x, step, Cm, lane, &s_simplex);
#endif
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M.H[3 * i + j] = x[i + j * 3];
    // ---- (4) points from the refined H
    auto project = [&](double px, double py, double *ox, double *oy) {
        const double a0 = M.H[0] * px + M.H[1] * py + M.H[2] * 1, a1 = M.H[3] * px + M.H[4] * py + M.H[5] * 1, a2 = M.H[6] * px + M.H[7] * py + M.H[8] * 1;
        *ox = a0 / a2;
        *oy = a1 / a2;
    };
    project(0.5, 0.5, &M.center[0], &M.center[1]);
    project(0, 0, &M.corners[0], &M.corners[1]);
    project(1, 0, &M.corners[2], &M.corners[3]);
    project(1, 1, &M.corners[4], &M.corners[5]);
    project(0, 1, &M.corners[6], &M.corners[7]);
    if (lane == 0) markers[m] = M;
}
__global__ __launch_bounds__(64) void k_stag_refine(fid_stag_marker *__restrict__ markers, const int *__restrict__ nmarkers, const int2 *__restrict__ vsegs, const int *__restrict__ nsegs, const int2 *__restrict__ pix, int *__restrict__ chosen_out)
{
    k_stag_refine_impl(markers, nmarkers, vsegs, nsegs, pix, chosen_out);
}
struct k_stag_refine_fn {
    static constexpr int kBounds = 64;
    __device__ __forceinline__ void operator()(fid_stag_marker *__restrict__ markers, const int *__restrict__ nmarkers, const int2 *__restrict__ vsegs, const int *__restrict__ nsegs, const int2 *__restrict__ pix, int *__restrict__ chosen_out) const { k_stag_refine_impl(markers, nmarkers, vsegs, nsegs, pix, chosen_out); }
};

// ------------------------------------------------------------------------------------------------ K17: marker pose
// StagNode::imageCallback -> Common::solvePnpSingle (stag_detect.cpp:140-165, common.hpp:34-46): cv::solvePnP (ITERATIVE) on
// FIVE coplanar points, the marker centre (0, 0, 0) and the four corners (-h, h) (h, h) (h, -h) (-h, -h), h = float(marker_size /
// 2).  Same scheme as the aruco pose kernel (fid_kernels.hip K8): closed-form start from the four corners, then the reference's
// Levenberg-Marquardt (CvLevMarq: <= 20 iterations, lambda 1e-3 x 10^k, same accept / reject rule) on the reprojection error
// of all five points with distortion; a 16-lane group per marker, lane g < 10 owns residual g.  Tolerance row (the reference
// starts from a 5-point DLT + refinement; both land on the same minimum).
__device__ __forceinline__ double grp_sum16(double v)
{
    v += shfl_xor_f64(v, 1);
    v += shfl_xor_f64(v, 2);
    v += shfl_xor_f64(v, 4);
    v += shfl_xor_f64(v, 8);
    return v;
}

__device__ void sp_undistort(const double K[9], const double kd[5], double u, double v, double *ox, double *oy)
{
    const double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1) / (1 + ((kd[4] * r2 + kd[1]) * r2 + kd[0]) * r2);
        if (icdist < 0) {
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        const double deltaX = 2 * kd[2] * x * y + kd[3] * (r2 + 2 * x * x);
        const double deltaY = kd[2] * (r2 + 2 * y * y) + 2 * kd[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *ox = x;
    *oy = y;
}

__device__ __forceinline__ void k_stag_pose_impl(const fid_stag_marker *__restrict__ markers, const int *__restrict__ nmarkers, PoseCam cam,
                                                  double marker_size, fid_stag_pose_out *__restrict__ out)
{
    const int item = blockIdx.x * 4 + (threadIdx.x >> 4), g = threadIdx.x & 15;
    if (item >= *nmarkers) return;  // group-uniform
    const fid_stag_marker mk = markers[item];
    const double *K = cam.K, *kd = cam.D;
    const float halff = (float)(marker_size / 2.0);
    const double hx = (double)halff;
    const bool act = g < 10;
    const int pi = act ? g >> 1 : 0, sel = g & 1;
    // object point of this lane: 0 centre, 1..4 corners
    double M[3] = {0., 0., 0.};
    if (pi >= 1) {
        M[0] = (pi == 2 || pi == 3) ? hx : -hx;
        M[1] = (pi <= 2) ? hx : -hx;
    }
    const double mobs = pi == 0 ? mk.center[sel] : mk.corners[2 * (pi - 1) + sel];
    double param[6];
    {
        double mnx[4], mny[4];
        for (int i = 0; i < 4; i++) {
            double x, y;
            sp_undistort(K, kd, mk.corners[2 * i], mk.corners[2 * i + 1], &x, &y);
            mnx[i] = x;
            mny[i] = y;
        }
        // homography marker plane -> normalised image through the four corners (unit square -> quad, composed with
        // (X, Y) -> ((X + h) / 2h, (h - Y) / 2h)), then R, t from its columns
        const double x0 = mnx[0], y0 = mny[0], x1 = mnx[1], y1 = mny[1], x2 = mnx[2], y2 = mny[2], x3 = mnx[3], y3 = mny[3];
        const double dx1 = x1 - x2, dx2 = x3 - x2, sx = x0 - x1 + x2 - x3;
        const double dy1 = y1 - y2, dy2 = y3 - y2, sy = y0 - y1 + y2 - y3;
        const double den = dx1 * dy2 - dy1 * dx2;
        double h[9];
        bool okh = den != 0.;
        if (okh) {
            const double gg = (sx * dy2 - sy * dx2) / den, hh = (dx1 * sy - dy1 * sx) / den;
            const double a = x1 - x0 + gg * x1, b = x3 - x0 + hh * x3, c = x0;
            const double d = y1 - y0 + gg * y1, e = y3 - y0 + hh * y3, ff = y0;
            const double sc0 = 1. / (2. * hx);
            h[0] = a * sc0;  h[1] = -b * sc0;  h[2] = 0.5 * a + 0.5 * b + c;
            h[3] = d * sc0;  h[4] = -e * sc0;  h[5] = 0.5 * d + 0.5 * e + ff;
            h[6] = gg * sc0; h[7] = -hh * sc0; h[8] = 0.5 * gg + 0.5 * hh + 1.;
            okh = h[8] != 0.;
            if (okh) {
                const double sc = 1. / h[8];
                for (int i = 0; i < 9; i++) h[i] *= sc;
            }
        }
        double R[9];
        param[3] = param[4] = param[5] = 0.;
        if (okh) {
            const double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            const double s1 = 1. / fmax(h1n, DBL_EPSILON), s2 = 1. / fmax(h2n, DBL_EPSILON), stt = 2. / fmax(h1n + h2n, DBL_EPSILON);
            param[3] = h[2] * stt; param[4] = h[5] * stt; param[5] = h[8] * stt;
            h[0] *= s1; h[3] *= s1; h[6] *= s1;
            h[1] *= s2; h[4] *= s2; h[7] *= s2;
            h[2] = h[3] * h[7] - h[6] * h[4];
            h[5] = h[6] * h[1] - h[0] * h[7];
            h[8] = h[0] * h[4] - h[3] * h[1];
            double rtmp[3], dummy[27];
            rodrigues_m2v(h, rtmp);
            rodrigues_v2m(rtmp, R, dummy, false);
        } else {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
        }
        rodrigues_m2v(R, param);
    }
    // ---- CvLevMarq over the 10 residuals
    double prevParam[6], S[21], gJ[6], Jrow[6] = {0, 0, 0, 0, 0, 0};
    double err = 0, prevErrNorm = 0, errNorm = 0;
    int lambdaLg10 = -3, iters = 0, state = 1;
    // (CvLevMarq: lambda = exp(lambdaLg10 * log(10.)): lm_lambda)
    for (int i = 0; i < 6; i++) prevParam[i] = param[i];
    for (;;) {
        bool needJ = false, needErr = false;
        if (state == 1) {
            needJ = needErr = true;
            state = 2;
        } else if (state == 2) {
            int idx = 0;
            for (int a = 0; a < 6; a++) {
                for (int b = a; b < 6; b++) S[idx++] = grp_sum16(Jrow[a] * Jrow[b]);
                gJ[a] = grp_sum16(Jrow[a] * err);
            }
            for (int i = 0; i < 6; i++) prevParam[i] = param[i];
            double xs[6];
            solve6_spd(S, gJ, lm_lambda(lambdaLg10), xs);
            for (int i = 0; i < 6; i++) param[i] = prevParam[i] - xs[i];
            if (iters == 0) prevErrNorm = sqrt(grp_sum16(err * err));
            needErr = true;
            state = 3;
        } else {
            errNorm = sqrt(grp_sum16(err * err));
            bool retry = false;
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) {
                    double xs[6];
                    solve6_spd(S, gJ, lm_lambda(lambdaLg10), xs);
                    for (int i = 0; i < 6; i++) param[i] = prevParam[i] - xs[i];
                    needErr = true;
                    state = 3;
                    retry = true;
                }
            }
            if (!retry) {
                lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
                double dn = 0, pn = 0;
                for (int i = 0; i < 6; i++) {
                    dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]);
                    pn += prevParam[i] * prevParam[i];
                }
                const double rel = sqrt(dn) / (sqrt(pn) + DBL_EPSILON);
                if (++iters >= 20 || rel < FLT_EPSILON) break;
                prevErrNorm = errNorm;
                needJ = needErr = true;
                state = 2;
            }
        }
        if (!needErr) break;
        const double pr = project_one(M, param, K, kd, sel, Jrow, needJ);
        err = act ? pr - mobs : 0.;
        if (!act)
            for (int i = 0; i < 6; i++) Jrow[i] = 0.;
    }
    if (g == 0) {
        fid_stag_pose_out o;
        o.id = mk.id;
        for (int i = 0; i < 3; i++) {
            o.rvec[i] = param[i];
            o.tvec[i] = param[3 + i];
        }
        double dummy[27];
        rodrigues_v2m(param, o.R, dummy, false);  // cv::Rodrigues(rVec, rMat) of solvePnpSingle
        out[item] = o;
    }
}
__global__ __launch_bounds__(64) void k_stag_pose(const fid_stag_marker *__restrict__ markers, const int *__restrict__ nmarkers, PoseCam cam, double marker_size, fid_stag_pose_out *__restrict__ out)
{
    k_stag_pose_impl(markers, nmarkers, cam, marker_size, out);
}
struct k_stag_pose_fn {
    static constexpr int kBounds = 64;
    __device__ __forceinline__ void operator()(const fid_stag_marker *__restrict__ markers, const int *__restrict__ nmarkers, PoseCam cam, double marker_size, fid_stag_pose_out *__restrict__ out) const { k_stag_pose_impl(markers, nmarkers, cam, marker_size, out); }
};
