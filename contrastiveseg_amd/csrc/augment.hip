// GPU data pipeline: the reference's per-sample CPU augmentation + tensor conversion + batch
// collation as ONE kernel over the output batch.
// Reference chain for the hot-path configs (configs/cityscapes/H_48_D_4.json: train_trans.trans_seq = random_resize,
// random_crop, random_hflip, random_brightness; data_transformer = fix_size / only_pad / pad_mode random):
//   lib/datasets/tools/cv2_aug_transforms.py:327-443  RandomResize   cv2.resize INTER_CUBIC (image) / INTER_NEAREST (label)
//   lib/datasets/tools/cv2_aug_transforms.py:504-603  RandomCrop     array slice
//   lib/datasets/tools/cv2_aug_transforms.py:143-209  RandomHFlip    cv2.flip(.., 1)
//   lib/datasets/tools/cv2_aug_transforms.py:305-325  RandomBrightness  += shift, around, clip to [0,255]
//   lib/datasets/tools/transforms.py:15-36, 63-103    ToTensor, Normalize(div, mean, std), ToLabel, ReLabel(255, -1)
//   lib/datasets/tools/collate.py:37-175              pad to input_size at (left_pad, up_pad): image 0, label -1
// All random decisions are drawn on the host in the reference's order (contrastiveseg_amd/lib/datasets/tools/gpu_aug.py)
// and arrive here as one parameter record per image. Every output pixel walks the chain BACKWARDS (pad -> flip -> crop ->
// resize) to its source coordinate, samples the raw uint8 image with the bicubic kernel of cv2 (a = -0.75, replicated
// border, rounded and clipped to uint8 like cv2's result) or the raw label with cv2's nearest rule, applies brightness,
// normalisation and the label look-up, and writes fp32 NCHW / int64 labels: one read of the source, one write of the
// batch, no intermediate image. HBM-bound (50 MB + 33 MB out at bs8, 1024x512 crops from 2048x1024 sources).
// INTER_CUBIC on uint8 is OpenCV's 11-bit fixed-point rule, restated bit for bit (round 4; oracle/aug_oracle.py has the derivation
// and what "bit for bit" means without the binary): float coefficients of interpolateCubic (A = -0.75f, the source's operation
// order, no fused multiply-adds), each rounded half-even to a short at scale 2^11; horizontal and vertical passes in int32;
// (v + 2^21) >> 22, saturated to [0, 255]. Integer arithmetic after the coefficients: the result does not depend on the
// evaluation order, so walking the chain backwards per output pixel gives the pixel cv2's two separable passes give.
#include "cseg_common.h"

namespace {

struct AugDims {
    int B, Hs, Ws, Ht, Wt;
    float div, mean[3], std[3];
};

// Individually rounded float operations. hipcc's default is -ffp-contract=fast, which fuses a product and a sum into one FMA even
// across statements and inlined calls -- ROCm 7.2's own __fmul_rn / __fadd_rn are plain `x * y` / `x + y` and fuse as well
// (checked in the gfx950 assembly) -- so the operations are spelled here under `contract(off)`: they carry no `contract` flag and the
// backend has nothing it may fuse.
__device__ __forceinline__ float f_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float f_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float f_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// OpenCV interpolateCubic (modules/imgproc/src/resize.cpp) in float with the source's operation order, every product and sum rounded
// on its own, then saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE) with INTER_RESIZE_COEF_BITS = 11 (cvRound: half to even).
__device__ __forceinline__ void cubic_coeffs_q11(float t, int (&q)[4]) {
    const float A = -0.75f;
    const float x1 = f_add(t, 1.f);
    float c[4];
    c[0] = f_sub(f_mul(f_add(f_mul(f_sub(f_mul(A, x1), 5.f * A), x1), 8.f * A), x1), 4.f * A);
    c[1] = f_add(f_mul(f_mul(f_sub(f_mul(A + 2.f, t), A + 3.f), t), t), 1.f);
    const float u = f_sub(1.f, t);
    c[2] = f_add(f_mul(f_mul(f_sub(f_mul(A + 2.f, u), A + 3.f), u), u), 1.f);
    c[3] = f_sub(f_sub(f_sub(1.f, c[0]), c[1]), c[2]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float r = rintf(f_mul(c[k], 2048.f));
        q[k] = (int)fminf(fmaxf(r, -32768.f), 32767.f);
    }
}

__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ lab,
                                                      const int16_t* __restrict__ lut, const int32_t* __restrict__ params,
                                                      AugDims d, float* __restrict__ out_img,
                                                      int64_t* __restrict__ out_lab) {
    const int b = blockIdx.z;
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= d.Wt || Y >= d.Ht) return;
    const int32_t* p = params + (size_t)b * CSEG_AUG_PARAM_INTS;
    const int Wr = p[0], Hr = p[1], x_off = p[2], y_off = p[3], tw = p[4], th = p[5];
    const int flip = p[6], shift = p[7], left_pad = p[8], up_pad = p[9];
    const size_t plane = (size_t)d.Ht * d.Wt;
    float* o = out_img + (size_t)b * 3 * plane + (size_t)Y * d.Wt + X;
    int64_t* ol = out_lab ? out_lab + (size_t)b * plane + (size_t)Y * d.Wt + X : nullptr;
    const int xc = X - left_pad, yc = Y - up_pad;
    if (xc < 0 || xc >= tw || yc < 0 || yc >= th) {          // collate padding: normalised image 0, label -1
        o[0] = 0.f; o[plane] = 0.f; o[2 * plane] = 0.f;
        if (ol) *ol = -1;
        return;
    }
    const int xr = x_off + (flip ? tw - 1 - xc : xc), yr = y_off + yc;      // pixel of the resized image
    const uint8_t* src = img + (size_t)b * d.Hs * d.Ws * 3;
    float v[3];
    int ls;
    if (Wr == d.Ws && Hr == d.Hs) {                          // resize skipped (or identity): plain copy
        const uint8_t* s = src + ((size_t)yr * d.Ws + xr) * 3;
        v[0] = (float)s[0]; v[1] = (float)s[1]; v[2] = (float)s[2];
        ls = lab ? lab[(size_t)b * d.Hs * d.Ws + (size_t)yr * d.Ws + xr] : 255;
    } else {
        // cv::resize: scale = 1 / (dsize / ssize) in double; source coordinate (dst + 0.5) * scale - 0.5
        const double scx = 1.0 / ((double)Wr / (double)d.Ws), scy = 1.0 / ((double)Hr / (double)d.Hs);
        const float fx = (float)(((double)xr + 0.5) * scx - 0.5), fy = (float)(((double)yr + 0.5) * scy - 0.5);
        const int sx = (int)floorf(fx), sy = (int)floorf(fy);
        int ax[4], ay[4];
        cubic_coeffs_q11(f_sub(fx, (float)sx), ax);
        cubic_coeffs_q11(f_sub(fy, (float)sy), ay);
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = min(max(sy - 1 + j, 0), d.Hs - 1);
            const uint8_t* row = src + (size_t)yy * d.Ws * 3;
            int r0 = 0, r1 = 0, r2 = 0;                      // HResizeCubic<uchar, int, short>: one row of the horizontal pass
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xx = min(max(sx - 1 + i, 0), d.Ws - 1);
                const uint8_t* s = row + (size_t)xx * 3;
                r0 += ax[i] * (int)s[0]; r1 += ax[i] * (int)s[1]; r2 += ax[i] * (int)s[2];
            }
            acc[0] += ay[j] * r0; acc[1] += ay[j] * r1; acc[2] += ay[j] * r2;      // VResizeCubic
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)                          // FixedPtCast<int, uchar, 22>: (v + 2^21) >> 22, saturated
            v[c] = (float)min(max((acc[c] + (1 << 21)) >> 22, 0), 255);
        // INTER_NEAREST: min(floor(dst * scale), src - 1)
        const int nx = min((int)floor((double)xr * scx), d.Ws - 1), ny = min((int)floor((double)yr * scy), d.Hs - 1);
        ls = lab ? lab[(size_t)b * d.Hs * d.Ws + (size_t)ny * d.Ws + nx] : 255;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = fminf(fmaxf(v[c] + (float)shift, 0.f), 255.f);        // brightness: integer shift, clip
        o[c * plane] = (t / d.div - d.mean[c]) / d.std[c];
    }
    if (ol) {
        const int t = lut ? (int)lut[ls] : ls;
        *ol = (t == 255) ? -1 : (int64_t)t;                                    // ReLabel(255, -1)
    }
}

}  // namespace

extern "C" int cseg_augment_batch(const uint8_t* img, const uint8_t* lab, const int16_t* lut, const int32_t* params, int B,
                                  int Hs, int Ws, int Ht, int Wt, float div_value, const float* mean3, const float* std3,
                                  float* out_img, int64_t* out_lab, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(img && params && out_img && mean3 && std3, "augment_batch: null pointer");
    CSEG_REQUIRE((lab == nullptr) == (out_lab == nullptr), "augment_batch: label input and output must come together");
    CSEG_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Ht > 0 && Wt > 0 && B <= 65535, "augment_batch: bad shape");
    CSEG_REQUIRE(div_value > 0.f && std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, "augment_batch: div/std must be > 0");
    AugDims d;
    d.B = B; d.Hs = Hs; d.Ws = Ws; d.Ht = Ht; d.Wt = Wt; d.div = div_value;
    for (int c = 0; c < 3; ++c) { d.mean[c] = mean3[c]; d.std[c] = std3[c]; }
    dim3 grid((Wt + 63) / 64, (Ht + 3) / 4, B);
    hipLaunchKernelGGL(augment_kernel, grid, dim3(256), 0, stream, img, lab, lut, params, d, out_img, out_lab);
    CSEG_CHECK_LAUNCH("augment_kernel");
    return 1;
}
