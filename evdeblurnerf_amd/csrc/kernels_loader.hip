// Image batch assembly on the device ("next" row f-3, first half): LLFFDataset.__getitem__, reference data/loader.py:325-356.
// The reference keeps images [n_img, H, W, 3] and poses [n_img, 3, 4] on the device and builds a batch with six indexing / stacking
// launches + get_rays_pix; here one thread per ray id does the unravel (:118-123, C order), the three gathers and the pixel-centre ray.
// Integer / gather work, a few hundred KB per batch: latency-bound, one launch is the whole optimisation.
#include "evd_common.h"

namespace evd {

__global__ __launch_bounds__(256) void k_image_batch(const long long* __restrict__ ids, long n, const float* __restrict__ images,
                                                     const float* __restrict__ pts0, const float* __restrict__ poses, int n_img, int H, int W,
                                                     float k00, float hx, float k11, float hy,
                                                     float* __restrict__ rays, float* __restrict__ rays_x, float* __restrict__ rays_y,
                                                     long long* __restrict__ img_idx, float* __restrict__ rgb, float* __restrict__ poses_out,
                                                     float* __restrict__ rgb0, int* __restrict__ invalid) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    const long long id = ids[i];
    const long long hw = (long long)H * W;
    if (id < 0 || id >= hw * n_img) {               // the reference's indexing raises; here zeros, image id -1 and the flag word
        if (invalid) atomicExch(invalid, 1);
        for (int k = 0; k < 6; ++k) rays[i * 6 + k] = 0.f;
        rays_x[i] = rays_y[i] = 0.f;
        img_idx[i] = -1;
        rgb[i * 3] = rgb[i * 3 + 1] = rgb[i * 3 + 2] = 0.f;
        if (poses_out) for (int k = 0; k < 12; ++k) poses_out[i * 12 + k] = 0.f;
        if (rgb0) rgb0[i * 3] = rgb0[i * 3 + 1] = rgb0[i * 3 + 2] = 0.f;
        return;
    }
    const long long im = id / hw, rem = id - im * hw;       // unravel_index(ray_id, (n_imgs, h, w)), loader.py:118-123
    const int y = (int)(rem / W), x = (int)(rem - (long long)y * W);
    const float* c2w = poses + im * 12;
    float p[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) p[k] = c2w[k];
    // get_rays_pix on the integer pixel (utils/rays.py:25-36): the int64 coordinate meets a float scalar -> float32 arithmetic
    const float d0 = ((float)x + hx) / k00, d1 = -((float)y + hy) / k11, d2 = -1.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rays[i * 6 + r * 2] = p[r * 4 + 3];
        rays[i * 6 + r * 2 + 1] = __fadd_rn(__fadd_rn(__fmul_rn(d0, p[r * 4]), __fmul_rn(d1, p[r * 4 + 1])), __fmul_rn(d2, p[r * 4 + 2]));
    }
    rays_x[i] = (float)x + 0.5f;                            // ray_x + HALF_PIX (:343-344)
    rays_y[i] = (float)y + 0.5f;
    img_idx[i] = im;
    const long long px = id * 3;                            // images[img_id, ray_y, ray_x] of a contiguous [n_img, H, W, 3] tensor
    rgb[i * 3] = images[px];
    rgb[i * 3 + 1] = images[px + 1];
    rgb[i * 3 + 2] = images[px + 2];
    if (poses_out) {
#pragma unroll
        for (int k = 0; k < 12; ++k) poses_out[i * 12 + k] = p[k];
    }
    if (rgb0) {
        rgb0[i * 3] = pts0[px];
        rgb0[i * 3 + 1] = pts0[px + 1];
        rgb0[i * 3 + 2] = pts0[px + 2];
    }
}

}  // namespace evd

using namespace evd;

extern "C" {

int evd_image_batch(const long long* ray_ids, long n, const float* images, const float* pts0_images, const float* poses, int n_img, int H,
                    int W, const float* K, float* rays, float* rays_x, float* rays_y, long long* images_idx, float* rgbsf, float* poses_out,
                    float* rgbsf_pts0, int* invalid, void* stream) {
    EVD_REQUIRE(n >= 0 && n_img > 0 && H > 0 && W > 0 && K, "evd_image_batch: bad arguments");
    EVD_REQUIRE(!rgbsf_pts0 || pts0_images, "evd_image_batch: rgbsf_pts0 wanted without pts0_images");
    hipStream_t st = as_stream(stream);
    if (invalid) EVD_HIP(hipMemsetAsync(invalid, 0, sizeof(int), st));
    if (n == 0) return EVD_OK;
    EVD_REQUIRE(ray_ids && images && poses && rays && rays_x && rays_y && images_idx && rgbsf, "evd_image_batch: null argument");
    // (halfpix - K[0][2]) is a Python double in the reference, rounded to float32 when it meets the tensor (utils/rays.py:28-29)
    const float hx = (float)(0.5 - (double)K[2]), hy = (float)(0.5 - (double)K[5]);
    hipLaunchKernelGGL(k_image_batch, dim3((unsigned)cdiv(n, 256L)), dim3(256), 0, st, ray_ids, n, images, pts0_images, poses, n_img, H, W,
                       K[0], hx, K[4], hy, rays, rays_x, rays_y, images_idx, rgbsf, poses_out, rgbsf_pts0, invalid);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // extern "C"
