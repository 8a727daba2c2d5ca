// geom.hip — geometric verification stand-in for the inter-robot loop-closure path (SURVEY.md
// §8f, next-row N4): rigid 3D-3D registration of matched keypoints by RANSAC.
//
// The reference delegates this step to third-party code that is not in its tree (rtabmap
// `RegistrationVis::computeTransformation` behind LightGlue matches,
// src/front_end/rgbd_handler.cpp:317-412, 657-726), so there is nothing to pin parity against:
// this is a FUNCTIONAL stand-in with rtabmap's parameter vocabulary (Vis/Iterations,
// Vis/InlierDistance, Vis/MinInliers) that closes the loop success/failure → candidate/fixed.
//
// One wave per hypothesis: lane 0 builds the rigid transform of a 3-correspondence sample
// (orthonormal frames on both sides), all 64 lanes count the correspondences it explains, one
// packed atomicMax keeps (most inliers, lowest hypothesis index) — deterministic for a seed.
// A second launch re-scores the winner and reduces the centroids and the 3x3 cross-covariance of
// its inliers (fixed-order tree), from which the caller gets the least-squares (Kabsch) refinement.
#include <cstring>
#include <vector>

#include "common.hpp"

using namespace cslam;

namespace {

struct Rigid {
    double r[9];  // row-major rotation
    double t[3];
};

__device__ inline unsigned int mix(unsigned long long x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return (unsigned int)x;
}

__device__ inline bool frame_of(const double (&p0)[3], const double (&p1)[3], const double (&p2)[3],
                                double (&f)[9]) {
    double u[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    double v[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    const double nu = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (!(nu > 1e-9)) return false;
    for (int k = 0; k < 3; ++k) u[k] /= nu;
    const double d = v[0] * u[0] + v[1] * u[1] + v[2] * u[2];
    for (int k = 0; k < 3; ++k) v[k] -= d * u[k];
    const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (!(nv > 1e-3 * nu)) return false;  // (nearly) collinear sample
    for (int k = 0; k < 3; ++k) v[k] /= nv;
    const double w[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    // columns u, v, w
    for (int k = 0; k < 3; ++k) {
        f[k * 3 + 0] = u[k];
        f[k * 3 + 1] = v[k];
        f[k * 3 + 2] = w[k];
    }
    return true;
}

// transform of hypothesis h (sample of three distinct correspondences); false = degenerate
__device__ inline bool hypothesis(const float* __restrict__ src, const float* __restrict__ dst,
                                  int n, unsigned long long seed, int h, Rigid& T) {
    int idx[3];
    idx[0] = mix(seed + 3ull * h) % n;
    idx[1] = mix(seed + 3ull * h + 1) % n;
    idx[2] = mix(seed + 3ull * h + 2) % n;
    if (idx[0] == idx[1] || idx[0] == idx[2] || idx[1] == idx[2]) return false;
    double a[3][3], b[3][3];
    for (int p = 0; p < 3; ++p)
        for (int k = 0; k < 3; ++k) {
            a[p][k] = src[3 * idx[p] + k];
            b[p][k] = dst[3 * idx[p] + k];
        }
    double fa[9], fb[9];
    if (!frame_of(a[0], a[1], a[2], fa) || !frame_of(b[0], b[1], b[2], fb)) return false;
    // R = Fb Fa^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += fb[i * 3 + k] * fa[j * 3 + k];
            T.r[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        T.t[i] = b[0][i] - (T.r[i * 3] * a[0][0] + T.r[i * 3 + 1] * a[0][1] + T.r[i * 3 + 2] * a[0][2]);
    return true;
}

__device__ inline bool explains(const Rigid& T, const float* __restrict__ src,
                                const float* __restrict__ dst, int i, double dist2) {
    const double x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
    double e2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double d = T.r[k * 3] * x + T.r[k * 3 + 1] * y + T.r[k * 3 + 2] * z + T.t[k] - dst[3 * i + k];
        e2 += d * d;
    }
    return e2 <= dist2;
}

__global__ __launch_bounds__(256) void k_ransac_score(const float* __restrict__ src,
                                                      const float* __restrict__ dst, int n,
                                                      int iterations, double dist2,
                                                      unsigned long long seed,
                                                      unsigned long long* __restrict__ best) {
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= iterations) return;
    Rigid T;
    const bool ok = hypothesis(src, dst, n, seed, h, T);  // every lane: same data, same result
    if (!ok) return;
    int cnt = 0;
    for (int i = lane; i < n; i += 64) cnt += explains(T, src, dst, i, dist2) ? 1 : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0)
        atomicMax(best, ((unsigned long long)cnt << 32) | (0xFFFFFFFFu - (unsigned int)h));
}

// re-score the winning hypothesis; inlier mask + sums for the least-squares refinement:
// out[0..2] = sum src, out[3..5] = sum dst, out[6..14] = sum src_i dst_j (row-major), out[15] = count
__global__ __launch_bounds__(1024) void k_ransac_refine_sums(const float* __restrict__ src,
                                                            const float* __restrict__ dst, int n,
                                                            double dist2, unsigned long long seed,
                                                            const unsigned long long* __restrict__ best,
                                                            unsigned char* __restrict__ mask,
                                                            double* __restrict__ out,
                                                            double* __restrict__ transform) {
    __shared__ double red[16][16];
    const unsigned long long key = *best;
    const int h = (int)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFu));
    Rigid T;
    const bool ok = key != 0ull && hypothesis(src, dst, n, seed, h, T);
    double s[16];
    for (int q = 0; q < 16; ++q) s[q] = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const bool in = ok && explains(T, src, dst, i, dist2);
        mask[i] = in ? 1 : 0;
        if (in) {
            const double a[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
            const double b[3] = {dst[3 * i], dst[3 * i + 1], dst[3 * i + 2]};
            for (int k = 0; k < 3; ++k) {
                s[k] += a[k];
                s[3 + k] += b[k];
                for (int j = 0; j < 3; ++j) s[6 + 3 * k + j] += a[k] * b[j];
            }
            s[15] += 1.0;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = 0; q < 16; ++q) {
        double v = s[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[q][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        double v = 0.0;
        for (int w = 0; w < 16; ++w) v += red[threadIdx.x][w];
        out[threadIdx.x] = v;
    }
    if (threadIdx.x == 0) {
        for (int q = 0; q < 9; ++q) transform[q] = ok ? T.r[q] : (q % 4 == 0 ? 1.0 : 0.0);
        for (int q = 0; q < 3; ++q) transform[9 + q] = ok ? T.t[q] : 0.0;
    }
}

}  // namespace

extern "C" int cslam_ransac_rigid3d(const float* src, const float* dst, int64_t n, int iterations,
                                    double inlier_distance, uint64_t seed, int device_id,
                                    double* sample_transform, double* inlier_sums,
                                    uint8_t* inlier_mask, int32_t* best_hypothesis) {
    if (!src || !dst || n < 0 || iterations <= 0 || !(inlier_distance > 0.0) || !sample_transform ||
        !inlier_sums)
        return fail(CSLAM_E_INVALID, "ransac_rigid3d: bad args");
    if (n > (1 << 26)) return fail(CSLAM_E_INVALID, "ransac_rigid3d: too many correspondences");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(CSLAM_E_NODEVICE, "ransac_rigid3d: no HIP device visible");
    if (device_id < 0 || device_id >= ndev) return fail(CSLAM_E_INVALID, "ransac_rigid3d: bad device");
    CSLAM_HIP_TRY(hipSetDevice(device_id));
    for (int q = 0; q < 16; ++q) inlier_sums[q] = 0.0;
    for (int q = 0; q < 12; ++q) sample_transform[q] = (q < 9 && q % 4 == 0) ? 1.0 : 0.0;
    if (best_hypothesis) *best_hypothesis = -1;
    if (n < 3) {
        if (inlier_mask && n > 0) std::memset(inlier_mask, 0, (size_t)n);
        return CSLAM_OK;  // nothing to fit: zero inliers, identity
    }
    DevBuf dsrc, ddst, dbest, dmask, dout;
    int rc = CSLAM_OK;
    auto done = [&](int r) {
        dsrc.release();
        ddst.release();
        dbest.release();
        dmask.release();
        dout.release();
        return r;
    };
    const size_t pts = (size_t)n * 3 * sizeof(float);
    if ((rc = dsrc.reserve(pts)) || (rc = ddst.reserve(pts)) || (rc = dbest.reserve(8)) ||
        (rc = dmask.reserve((size_t)n)) || (rc = dout.reserve(28 * sizeof(double))))
        return done(rc);
    hipError_t e = hipMemcpy(dsrc.p, src, pts, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ddst.p, dst, pts, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dbest.p, 0, 8);
    if (e != hipSuccess) return done(fail(CSLAM_E_HIP, "ransac_rigid3d: upload: %s", hipGetErrorString(e)));
    const double d2 = inlier_distance * inlier_distance;
    hipLaunchKernelGGL(k_ransac_score, dim3((unsigned)((iterations + 3) / 4)), dim3(256), 0, 0,
                       (const float*)dsrc.p, (const float*)ddst.p, (int)n, iterations, d2,
                       (unsigned long long)seed, (unsigned long long*)dbest.p);
    double* dsum = (double*)dout.p;
    hipLaunchKernelGGL(k_ransac_refine_sums, dim3(1), dim3(1024), 0, 0, (const float*)dsrc.p,
                       (const float*)ddst.p, (int)n, d2, (unsigned long long)seed,
                       (const unsigned long long*)dbest.p, (unsigned char*)dmask.p, dsum, dsum + 16);
    e = hipGetLastError();
    double host[28];
    unsigned long long key = 0;
    if (e == hipSuccess) e = hipMemcpy(host, dout.p, sizeof(host), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&key, dbest.p, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && inlier_mask) e = hipMemcpy(inlier_mask, dmask.p, (size_t)n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return done(fail(CSLAM_E_HIP, "ransac_rigid3d: %s", hipGetErrorString(e)));
    for (int q = 0; q < 16; ++q) inlier_sums[q] = host[q];
    for (int q = 0; q < 12; ++q) sample_transform[q] = host[16 + q];
    if (best_hypothesis && key != 0ull) *best_hypothesis = (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu));
    return done(CSLAM_OK);
}
