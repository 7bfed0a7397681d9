// Fisheye anchor reprojection of ONE proposal joint into the four cameras (utils/camera_models.py:53-104 +
// egoposeformer_mvf_ex.py:340-348, 400-406), shared by fisheye_kernel (egr_attn.hip) and pose_query_kernel (egr_layer.hip).
#pragma once
#include <cstdint>

namespace egrf {

constexpr int CAM_REC = 17;  // [npoly, cx, cy, W, H, poly[12]]

__device__ __forceinline__ void fisheye_one(const float* cam, float x, float y, float z, float* u_out, float* v_out,
                                            uint8_t* ok_out) {
    const int npoly = (int)cam[0];
    const float cx = cam[1], cy = cam[2], W = cam[3], H = cam[4];
    float norm = sqrtf(x * x + y * y);
    float theta = atanf(-z / norm);
    // rho = sum_i a_i * theta^i, left to right from 0 (utils/camera_models.py:85)
    float rho = 0.f, pw = 1.f;
    for (int i = 0; i < npoly; ++i) {
        rho = rho + cam[5 + i] * pw;
        pw *= theta;
    }
    float u = x / norm * rho + cx;
    float v = y / norm * rho + cy;
    u = u / W;
    v = v / H;
    *ok_out = (u > 0.f && v > 0.f && u < 1.f && v < 1.f) ? 1 : 0;
    *u_out = fminf(fmaxf(u, 0.f), 1.f);
    *v_out = fminf(fmaxf(v, 0.f), 1.f);
}

// Joint j of frame b: (x, y, z) in, the (possibly mutated) point out.  ctm == nullptr: ego4view_syn - the reference mutates its
// argument in place, so the four cameras chain; otherwise rw: M . [p * 0.01, 1] * 100, the point untouched.
// anchors (b, 4, J, 2), valid (b, 4, J).
__device__ __forceinline__ void fisheye_joint(float& x, float& y, float& z, const float* ctm, const float* cams, int b, int j, int J,
                                              float* anchors, uint8_t* valid) {
    // ego4view_syn rigid offsets (cm): FL, FR, BL, BR; the back cameras flip x,y first (camera_models.py:29-40,59-63)
    const float offx[4] = {6.f, -6.f, -6.f, 6.f};
    const float offy[4] = {0.f, 0.f, 37.f, 37.f};
    for (int c = 0; c < 4; ++c) {
        float px, py, pz;
        if (ctm) {
            const float* m = ctm + ((int64_t)b * 4 + c) * 16;
            float sx = x * 0.01f, sy = y * 0.01f, sz = z * 0.01f;
            px = (m[0] * sx + m[1] * sy + m[2] * sz + m[3]) * 100.f;
            py = (m[4] * sx + m[5] * sy + m[6] * sz + m[7]) * 100.f;
            pz = (m[8] * sx + m[9] * sy + m[10] * sz + m[11]) * 100.f;
        } else {
            if (c >= 2) { x = x * -1.f; y = y * -1.f; }
            x += offx[c]; y += offy[c]; z += 0.f;
            px = x; py = y; pz = z;
        }
        float u, v;
        uint8_t ok;
        fisheye_one(cams + c * CAM_REC, px, py, pz, &u, &v, &ok);
        int64_t o = ((int64_t)b * 4 + c) * J + j;
        anchors[o * 2 + 0] = u;
        anchors[o * 2 + 1] = v;
        valid[o] = ok;
    }
}

}  // namespace egrf
