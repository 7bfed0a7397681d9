// Pose evaluation metrics on the device (a25 is the path's parity metric).
// Replaces evaluate_pose of pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:317-333, i.e.
//   compute_mpjpe_batch / compute_pck_3d_batch / compute_auc_3d_batch   (utils/loss.py:9-48)
//   batch_compute_similarity_transform_numpy -> compute_similarity_transform (models/utils/pose_metric.py:104-167),
// which the reference runs on the host with one numpy SVD per sample after a device-to-host copy.
// One thread per sample: 16 joints, a 3x3 cross-covariance, its SVD by cyclic Jacobi on K^T K in fp64 (tiny work;
// fp64 keeps the rotation accurate to fp32 round-off whatever the conditioning).
#include "egr_common.h"

namespace {

constexpr int MAXJ = 32;

__device__ void jacobi_eig3(double a[3][3], double v[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(a[p][q]) < 1e-300) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {  // V <- V J
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

__device__ double det3(const double m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

__global__ __launch_bounds__(64) void pose_metrics_kernel(const float* pred, const float* gt, int B, int J, float pck_thr_mm,
                                                          int n_auc, float* out, float* aligned) {
    int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* P = pred + (int64_t)b * J * 3;
    const float* G = gt + (int64_t)b * J * 3;
    // ---- MPJPE, PCK, AUC on the raw prediction (cm -> mm by x10 on each operand, as the reference does)
    float sum_err = 0.f;
    float dmm[MAXJ];
    for (int j = 0; j < J; ++j) {
        // no FMA contraction here: each cm->mm product is rounded to fp32 before the subtraction, as in the reference,
        // so a perfect prediction has distance exactly 0 and passes the threshold-0 bin of the AUC
#pragma clang fp contract(off)
        float dx = P[3 * j] - G[3 * j], dy = P[3 * j + 1] - G[3 * j + 1], dz = P[3 * j + 2] - G[3 * j + 2];
        sum_err += sqrtf(dx * dx + dy * dy + dz * dz);
        float gx = G[3 * j] * 10.f, gy = G[3 * j + 1] * 10.f, gz = G[3 * j + 2] * 10.f;
        float px = P[3 * j] * 10.f, py = P[3 * j + 1] * 10.f, pz = P[3 * j + 2] * 10.f;
        float ex = gx - px, ey = gy - py, ez = gz - pz;
        dmm[j] = sqrtf(ex * ex + ey * ey + ez * ez);
    }
    float mpjpe_mm = sum_err / (float)J * 10.f;
    int correct = 0;
    for (int j = 0; j < J; ++j) correct += dmm[j] <= pck_thr_mm;
    float pck = (float)correct / (float)J * 100.f;
    float auc = 0.f;
    for (int t = 0; t < n_auc; ++t) {  // thresholds linspace(0, pck_thr, n_auc)
        float thr = (n_auc > 1) ? pck_thr_mm * (float)t / (float)(n_auc - 1) : pck_thr_mm;
        int c = 0;
        for (int j = 0; j < J; ++j) c += dmm[j] <= thr;
        auc += (float)c / (float)J;
    }
    auc = auc / (float)n_auc * 100.f;
    // ---- similarity (Procrustes) alignment of pred onto gt: compute_similarity_transform
    double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
    for (int j = 0; j < J; ++j)
        for (int c = 0; c < 3; ++c) {
            mu1[c] += P[3 * j + c];
            mu2[c] += G[3 * j + c];
        }
    for (int c = 0; c < 3; ++c) { mu1[c] /= J; mu2[c] /= J; }
    double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
    for (int j = 0; j < J; ++j) {
        double x1[3], x2[3];
        for (int c = 0; c < 3; ++c) { x1[c] = P[3 * j + c] - mu1[c]; x2[c] = G[3 * j + c] - mu2[c]; var1 += x1[c] * x1[c]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) K[r][c] += x1[r] * x2[c];  // K = X1 X2^T
    }
    // SVD of K via eigen-decomposition of K^T K = V S^2 V^T;  U = K V S^-1
    double A[3][3], V[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = K[0][r] * K[0][c] + K[1][r] * K[1][c] + K[2][r] * K[2][c];
    jacobi_eig3(A, V);
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j2 = i + 1; j2 < 3; ++j2)
            if (A[ord[j2]][ord[j2]] > A[ord[i]][ord[i]]) { int t = ord[i]; ord[i] = ord[j2]; ord[j2] = t; }
    double Vs[3][3], U[3][3], sv[3];
    for (int i = 0; i < 3; ++i) {
        sv[i] = sqrt(fmax(A[ord[i]][ord[i]], 0.0));
        for (int r = 0; r < 3; ++r) Vs[r][i] = V[r][ord[i]];
    }
    for (int i = 0; i < 3; ++i) {
        double u[3] = {K[0][0] * Vs[0][i] + K[0][1] * Vs[1][i] + K[0][2] * Vs[2][i],
                       K[1][0] * Vs[0][i] + K[1][1] * Vs[1][i] + K[1][2] * Vs[2][i],
                       K[2][0] * Vs[0][i] + K[2][1] * Vs[1][i] + K[2][2] * Vs[2][i]};
        double nrm = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (nrm > 1e-12 * (sv[0] + 1e-300) && i < 2) {
            for (int r = 0; r < 3; ++r) U[r][i] = u[r] / nrm;
        } else if (i == 2) {
            // third left singular vector: use K v / s when well defined, else complete the basis
            if (nrm > 1e-9 * (sv[0] + 1e-300)) for (int r = 0; r < 3; ++r) U[r][2] = u[r] / nrm;
            else {
                U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
                U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
                U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
            }
        } else {
            for (int r = 0; r < 3; ++r) U[r][i] = (r == i) ? 1.0 : 0.0;
        }
    }
    // Z fixes det(R) = +1: Z = diag(1, 1, sign(det(U V^T)));  R = V Z U^T
    double UVt[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) UVt[r][c] = U[r][0] * Vs[c][0] + U[r][1] * Vs[c][1] + U[r][2] * Vs[c][2];
    double dt = det3(UVt);
    double z = dt > 0 ? 1.0 : (dt < 0 ? -1.0 : 0.0);
    double R[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r][c] = Vs[r][0] * U[c][0] + Vs[r][1] * U[c][1] + z * Vs[r][2] * U[c][2];
    double trRK = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) trRK += R[r][c] * K[c][r];
    double scale = trRK / var1;
    double tvec[3];
    for (int r = 0; r < 3; ++r) tvec[r] = mu2[r] - scale * (R[r][0] * mu1[0] + R[r][1] * mu1[1] + R[r][2] * mu1[2]);
    float pa_sum = 0.f;
    for (int j = 0; j < J; ++j) {
        float h[3];
        for (int r = 0; r < 3; ++r)
            h[r] = (float)(scale * (R[r][0] * P[3 * j] + R[r][1] * P[3 * j + 1] + R[r][2] * P[3 * j + 2]) + tvec[r]);
        if (aligned) { aligned[((int64_t)b * J + j) * 3 + 0] = h[0]; aligned[((int64_t)b * J + j) * 3 + 1] = h[1]; aligned[((int64_t)b * J + j) * 3 + 2] = h[2]; }
        float dx = h[0] - G[3 * j], dy = h[1] - G[3 * j + 1], dz = h[2] - G[3 * j + 2];
        pa_sum += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    out[4 * b + 0] = mpjpe_mm;
    out[4 * b + 1] = pa_sum / (float)J * 10.f;
    out[4 * b + 2] = pck;
    out[4 * b + 3] = auc;
}

}  // namespace

extern "C" int egr_pose_metrics_f32(const float* pred, const float* gt, int32_t b, int32_t joints, float pck_thr_mm,
                                    int32_t n_auc, float* out, float* aligned, void* stream) {
    if (!pred || !gt || !out) return EGR_ENULL;
    if (b <= 0 || joints <= 0 || joints > MAXJ || n_auc <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(pose_metrics_kernel, dim3((unsigned)((b + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pred, gt, b, joints,
                       pck_thr_mm, n_auc, out, aligned);
    return egr_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Ground-truth heat-map synthesis: replaces generate_target of generate_heatmap.py:10-48
// (offline numpy loop writing one .npy per frame and camera).  One workgroup per (sample, joint) map: zero fill and a
// (2*tmp+1)^2 Gaussian window centred on int(x / stride + 0.5), clipped to the map; the window table is computed on the
// host with numpy float32 exactly as the reference does, so values are bit-identical.
namespace {
__global__ __launch_bounds__(256) void gt_heatmap_kernel(const double* joints, int maps, double stride, int hs, int tmp,
                                                         const float* gauss, float* out) {
    const int map = blockIdx.x;
    if (map >= maps) return;
    float* o = out + (int64_t)map * hs * hs;
    const double jx = joints[2 * map], jy = joints[2 * map + 1];
    const int mu_x = (int)(jx / stride + 0.5), mu_y = (int)(jy / stride + 0.5);   // C cast == Python int(): toward zero
    const int ulx = mu_x - tmp, uly = mu_y - tmp, brx = mu_x + tmp + 1, bry = mu_y + tmp + 1;
    const bool visible = !(ulx >= hs || uly >= hs || brx < 0 || bry < 0);
    const int size = 2 * tmp + 1;
    for (int i = threadIdx.x; i < hs * hs; i += 256) {
        int y = i / hs, x = i - y * hs;
        float v = 0.f;
        if (visible && x >= ulx && x < brx && y >= uly && y < bry) v = gauss[(y - uly) * size + (x - ulx)];
        o[i] = v;
    }
}
}  // namespace

extern "C" int egr_gt_heatmap_f32(const double* joints, int32_t maps, double image_size, int32_t heatmap_size, int32_t tmp_size,
                                  const float* gauss, float* out, void* stream) {
    if (!joints || !gauss || !out) return EGR_ENULL;
    if (maps <= 0 || heatmap_size <= 0 || tmp_size < 0 || image_size <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(gt_heatmap_kernel, dim3((unsigned)maps), dim3(256), 0, (hipStream_t)stream, joints, maps,
                       image_size / (double)heatmap_size, heatmap_size, tmp_size, gauss, out);
    return egr_launch_status();
}
