// f4: image comparison metrics of img-compression/img_comparison_metrics.py
// (mse :6-16, psnr :19-33, _SSIMForMultiScale :84-157, ms_ssim :160-220) for batches of uint8
// images [B][H][W][C], in the reference's float64 arithmetic.  gfx950 / CDNA4 only.
//
//  * squared error: integer arithmetic, exact (the reference's f64 sums of integer squares are exact too).
//  * one MS-SSIM scale: 'valid' Gaussian window (separable: g_ij = e_i e_j / (sum e)^2) over
//    x, y, x^2, y^2, xy, then the ssim and cs maps and their per-image sums.  The reference gets
//    the five filtered planes from scipy.signal.fftconvolve, whose rounding is not reproducible;
//    this is the direct sum, compared at 1e-9 relative.  Partial sums are written per workgroup
//    and added in index order by a second kernel: the result does not depend on scheduling.
//  * 2x2 box + decimation between scales (scipy.ndimage.convolve(mode='reflect')[::2, ::2]):
//    out[y][x] = mean of in[2y..2y+1][2x..2x+1] with the index clamped at the far edge; exact in f64
//    (dyadic rationals of < 20 bits).
#include "vbq_common.h"

namespace vbq {
namespace {

constexpr int kTile = 16;        // output tile edge
constexpr int kMaxWin = 11;      // filter_size of the reference (img_comparison_metrics.py:84)

__global__ void __launch_bounds__(256)
k_sqerr_u8(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, long n_per_image, unsigned long long *__restrict__ out) {
    const long img = blockIdx.y;
    const uint8_t *pa = a + img * n_per_image, *pb = b + img * n_per_image;
    unsigned long long s = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_image; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)pa[i] - (int)pb[i];
        s += (unsigned long long)(d * d);
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&out[img], s);      // integer adds: order-independent
}

__global__ void __launch_bounds__(256)
k_u8_to_f64(const uint8_t *__restrict__ a, long n, double *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (double)a[i];
}

// One workgroup = one 16 x 16 tile of the 'valid' output of one (image, channel) plane.
__global__ void __launch_bounds__(256)
k_ssim_tile(const double *__restrict__ im1, const double *__restrict__ im2, int H, int W, int C, const double *__restrict__ win,
            int size, double c1, double c2, double *__restrict__ part_ssim, double *__restrict__ part_cs, int tiles_x,
            int tiles_y) {
    constexpr int P = kTile + kMaxWin - 1;                            // 26: patch edge
    __shared__ double x1[P][P + 1], x2[P][P + 1];
    __shared__ double h[5][P][kTile + 1];                            // horizontally filtered x, y, xx, yy, xy
    __shared__ double w[kMaxWin];
    __shared__ double red[2][4];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, c = blockIdx.y, b = blockIdx.z;
    const int Ho = H - size + 1, Wo = W - size + 1;
    const int ox = tx * kTile, oy = ty * kTile;
    const int ph = min(kTile, Ho - oy) + size - 1, pw = min(kTile, Wo - ox) + size - 1;
    if (tid < size) w[tid] = win[tid];
    const long base = (long)b * H * W * C + c;
    for (int i = tid; i < P * P; i += 256) {
        const int r = i / P, q = i - r * P;
        double v1 = 0.0, v2 = 0.0;
        if (r < ph && q < pw) {
            const long o = base + ((long)(oy + r) * W + (ox + q)) * C;
            v1 = im1[o];
            v2 = im2[o];
        }
        x1[r][q] = v1;
        x2[r][q] = v2;
    }
    __syncthreads();
    for (int i = tid; i < P * kTile; i += 256) {
        const int r = i / kTile, q = i - r * kTile;
        double s1 = 0, s2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int k = 0; k < size; ++k) {
            const double a = x1[r][q + k], bb = x2[r][q + k], g = w[k];
            s1 = fma(g, a, s1);
            s2 = fma(g, bb, s2);
            s11 = fma(g, a * a, s11);
            s22 = fma(g, bb * bb, s22);
            s12 = fma(g, a * bb, s12);
        }
        h[0][r][q] = s1; h[1][r][q] = s2; h[2][r][q] = s11; h[3][r][q] = s22; h[4][r][q] = s12;
    }
    __syncthreads();
    const int r = tid / kTile, q = tid - r * kTile;
    double ssim = 0.0, cs = 0.0;
    if (oy + r < Ho && ox + q < Wo) {
        double mu1 = 0, mu2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int k = 0; k < size; ++k) {
            const double g = w[k];
            mu1 = fma(g, h[0][r + k][q], mu1);
            mu2 = fma(g, h[1][r + k][q], mu2);
            s11 = fma(g, h[2][r + k][q], s11);
            s22 = fma(g, h[3][r + k][q], s22);
            s12 = fma(g, h[4][r + k][q], s12);
        }
        const double mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;    // img_comparison_metrics.py:141-146
        s11 -= mu11;
        s22 -= mu22;
        s12 -= mu12;
        const double v1 = 2.0 * s12 + c2, v2 = s11 + s22 + c2;                  // :151-152
        ssim = ((2.0 * mu12 + c1) * v1) / ((mu11 + mu22 + c1) * v2);            // :153
        cs = v1 / v2;                                                           // :154
    }
    for (int m = 32; m >= 1; m >>= 1) {
        ssim += __shfl_xor(ssim, m, 64);
        cs += __shfl_xor(cs, m, 64);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = ssim; red[1][tid >> 6] = cs; }
    __syncthreads();
    if (tid == 0) {
        const long slot = ((long)b * C + c) * (tiles_x * tiles_y) + blockIdx.x;
        part_ssim[slot] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part_cs[slot] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// fixed-order sum of an image's partials, divided by the number of map entries: np.mean(axis=(1,2,3))
__global__ void k_ssim_finish(const double *__restrict__ part_ssim, const double *__restrict__ part_cs, long per_image, double count,
                              double *__restrict__ out_ssim, double *__restrict__ out_cs) {
    __shared__ double r0[256], r1[256];
    const long b = blockIdx.x;
    double s = 0.0, c = 0.0;
    for (long i = threadIdx.x; i < per_image; i += 256) {
        s += part_ssim[b * per_image + i];
        c += part_cs[b * per_image + i];
    }
    r0[threadIdx.x] = s;
    r1[threadIdx.x] = c;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) { r0[threadIdx.x] += r0[threadIdx.x + m]; r1[threadIdx.x] += r1[threadIdx.x + m]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out_ssim[b] = r0[0] / count; out_cs[b] = r1[0] / count; }
}

__global__ void __launch_bounds__(256)
k_downsample2(const double *__restrict__ in, int B, int H, int W, int C, double *__restrict__ out) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long n = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long t = i / C;
        const int x = (int)(t % Wo);
        t /= Wo;
        const int y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const int y0 = 2 * y, y1 = min(2 * y + 1, H - 1), x0 = 2 * x, x1 = min(2 * x + 1, W - 1);
        const double *p = in + (long)b * H * W * C + c;
        const double s = (p[((long)y0 * W + x0) * C] + p[((long)y0 * W + x1) * C]) + (p[((long)y1 * W + x0) * C] + p[((long)y1 * W + x1) * C]);
        out[i] = 0.25 * s;
    }
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_image_sqerr_u8(const uint8_t *d_img1, const uint8_t *d_img2, int64_t n_images, int64_t n_per_image,
                                  int64_t *d_out_sum, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_images >= 0 && n_per_image >= 0 && n_per_image < (1ll << 47), VBQ_ERR_INVALID_ARGUMENT,
                "vbq_image_sqerr_u8: bad shape %lld x %lld", (long long)n_images, (long long)n_per_image);
    if (n_images == 0) return VBQ_OK;
    VBQ_REQUIRE(d_out_sum, VBQ_ERR_INVALID_ARGUMENT, "vbq_image_sqerr_u8: null output");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(d_out_sum, 0, sizeof(int64_t) * n_images, st) != hipSuccess) {
        set_error("vbq_image_sqerr_u8: hipMemsetAsync failed");
        return VBQ_ERR_LAUNCH;
    }
    if (n_per_image == 0) return VBQ_OK;
    VBQ_REQUIRE(d_img1 && d_img2, VBQ_ERR_INVALID_ARGUMENT, "vbq_image_sqerr_u8: null input");
    VBQ_REQUIRE(n_images <= 65535, VBQ_ERR_UNSUPPORTED, "vbq_image_sqerr_u8: at most 65535 images per call");
    int64_t gx = (n_per_image + 256 * 16 - 1) / (256 * 16);
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_sqerr_u8, dim3((unsigned)gx, (unsigned)n_images), dim3(256), 0, st, d_img1, d_img2, (long)n_per_image,
                       reinterpret_cast<unsigned long long *>(d_out_sum));
    VBQ_CHECK_LAUNCH("image_sqerr");
    return VBQ_OK;
}

extern "C" int vbq_u8_to_f64(const uint8_t *d_in, int64_t n, double *d_out, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_u8_to_f64: n < 0");
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_in && d_out, VBQ_ERR_INVALID_ARGUMENT, "vbq_u8_to_f64: null pointer");
    int64_t gx = (n + 255) / 256;
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(k_u8_to_f64, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, (long)n, d_out);
    VBQ_CHECK_LAUNCH("u8_to_f64");
    return VBQ_OK;
}

namespace vbq {
namespace {
// np.clip(np.round(x * 255), 0, 255).astype(np.uint8) (utils.py:555): one f32 multiply, round half to even (np.round = rint),
// clip, convert.  NaN -> 0 (NumPy's cast of NaN is undefined; the reconstructions are clipped to [0, 1] upstream).
__global__ void __launch_bounds__(256) k_unit_to_u8(const float *__restrict__ x, long n, uint8_t *__restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = rintf(__fmul_rn(x[i], 255.0f));
        v = v >= 0.0f ? (v <= 255.0f ? v : 255.0f) : 0.0f;
        out[i] = (uint8_t)v;
    }
}
}  // namespace
}  // namespace vbq

extern "C" int vbq_unit_to_u8_f32(const float *d_in, int64_t n, uint8_t *d_out, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_unit_to_u8_f32: n < 0");
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_in && d_out, VBQ_ERR_INVALID_ARGUMENT, "vbq_unit_to_u8_f32: null pointer");
    int64_t gx = (n + 255) / 256;
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(k_unit_to_u8, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, (long)n, d_out);
    VBQ_CHECK_LAUNCH("unit_to_u8");
    return VBQ_OK;
}

extern "C" size_t vbq_ssim_scale_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t size) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || size <= 0 || size > H || size > W) return 0;
    const int64_t tx = (W - size + 1 + vbq::kTile - 1) / vbq::kTile, ty = (H - size + 1 + vbq::kTile - 1) / vbq::kTile;
    return (size_t)2 * sizeof(double) * (size_t)B * C * tx * ty;
}

extern "C" int vbq_ssim_scale_f64(const double *d_im1, const double *d_im2, int32_t B, int32_t H, int32_t W, int32_t C,
                                  const double *d_window, int32_t size, double c1, double c2, double *d_out_ssim,
                                  double *d_out_cs, void *d_workspace, size_t workspace_bytes, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && C <= 65535 && B <= 65535, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_ssim_scale_f64: bad shape [%d][%d][%d][%d]", B, H, W, C);
    VBQ_REQUIRE(size >= 1 && size <= kMaxWin && size <= H && size <= W, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_ssim_scale_f64: window %d must be in [1, %d] and fit the image", size, kMaxWin);
    if (B == 0) return VBQ_OK;
    VBQ_REQUIRE(d_im1 && d_im2 && d_window && d_out_ssim && d_out_cs && d_workspace, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_ssim_scale_f64: null pointer");
    const size_t need = vbq_ssim_scale_workspace_bytes(B, H, W, C, size);
    VBQ_REQUIRE(workspace_bytes >= need, VBQ_ERR_WORKSPACE, "vbq_ssim_scale_f64: workspace %zu < %zu bytes", workspace_bytes, need);
    const int Ho = H - size + 1, Wo = W - size + 1;
    const int tx = (Wo + kTile - 1) / kTile, ty = (Ho + kTile - 1) / kTile;
    double *ps = reinterpret_cast<double *>(d_workspace);
    double *pc = ps + (size_t)B * C * tx * ty;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_ssim_tile, dim3((unsigned)(tx * ty), (unsigned)C, (unsigned)B), dim3(256), 0, st, d_im1, d_im2, (int)H, (int)W,
                       (int)C, d_window, (int)size, c1, c2, ps, pc, tx, ty);
    hipLaunchKernelGGL(k_ssim_finish, dim3((unsigned)B), dim3(256), 0, st, ps, pc, (long)C * tx * ty, (double)Ho * Wo * C,
                       d_out_ssim, d_out_cs);
    VBQ_CHECK_LAUNCH("ssim_scale");
    return VBQ_OK;
}

extern "C" int vbq_downsample2_f64(const double *d_in, int32_t B, int32_t H, int32_t W, int32_t C, double *d_out, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_downsample2_f64: bad shape");
    if (B == 0) return VBQ_OK;
    VBQ_REQUIRE(d_in && d_out, VBQ_ERR_INVALID_ARGUMENT, "vbq_downsample2_f64: null pointer");
    const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * C;
    int64_t gx = (n + 255) / 256;
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(k_downsample2, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, (int)B, (int)H,
                       (int)W, (int)C, d_out);
    VBQ_CHECK_LAUNCH("downsample2");
    return VBQ_OK;
}
