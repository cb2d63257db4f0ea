// kernel_ingest.h — N3: the frame preparation in front of the path, on the device.
//
//   ingest_geom_kernel     VideoCapture::start  VideoCapture.cpp:158-207   cvResize (INTER_LINEAR, 8-bit fixed point) -> cvFlip(mode 0)
//                                                                           -> ROI view, as ONE gather: every output pixel is computed
//                                                                           straight from the source frame (pointwise steps compose)
//   ingest_hist / lut / apply   PreProcessor::process  PreProcessor.cpp:63-64   cv::equalizeHist (1-channel)
//   ingest_blur7_kernel    PreProcessor::process  PreProcessor.cpp:68-69   cv::GaussianBlur(7x7, sigma 1.5): separable integer kernel,
//                                                                           BORDER_REFLECT_101, both passes in one launch through LDS
// Flip and ROI are exact by definition.  The OpenCV 2.4 arithmetic is restated from recall (OpenCV is neither in the reference
// tree nor in the build image: unpinned):
//   R1 cv::resize INTER_LINEAR 8U: scale = src/dst (double); fx = (float)((dx+0.5)*scale - 0.5); sx = floor(fx); fx -= sx; sx < 0 -> (0, 0);
//      sx >= w-1 -> (w-1, 0); coefficients saturate_cast<short>(c * 2048); horizontal pass in int; rows sy, sy+1 clipped to the image;
//      vertical pass uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2); an exact 2x reduction takes
//      INTER_AREA's fast path, (a + b + c + d + 2) >> 2.
//   R2 cv::equalizeHist: i = first non-empty bin; scale = 255.f / (total - hist[i]); lut[i] = 0, lut[j] = saturate_cast<uchar>(sum * scale).
//   R3 cv::GaussianBlur 8U, smooth symmetric kernel -> fixed point: integer kernel cvRound(cf * 256.f) both ways, row pass exact,
//      column pass (sum + 2^15) >> 16, BORDER_REFLECT_101.
// HBM-bound byte kernels: r (scale^2 x 3) + w 3 B/pixel for the gather, r 3 + w 3 for the blur.
#pragma once
#include "bgs_device.h"

namespace bgs {

struct IngestArgs {
  const uint8_t* src;  // [images] frames, rows src_step bytes apart, images src_rows*src_step apart
  uint8_t* dst;        // [images][rows][cols][C] contiguous
  int src_rows, src_cols, rw, rh;  // source and resized geometry
  int rows, cols;      // output geometry (the ROI, or rw x rh)
  int x0, y0, flip;
  int mode;            // 0: same size (copy), 1: bilinear, 2: exact 2x2 area average
  size_t src_step;
  double scale_x, scale_y;
};

__device__ __forceinline__ int ingest_sat_short(float v) {
  const int r = __float2int_rn(v);
  return min(max(r, -32768), 32767);
}

// one lane per output pixel
template <int C>
__global__ __launch_bounds__(kBlock) void ingest_geom_kernel(const IngestArgs a) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t n = (size_t)a.rows * a.cols;
  if (p >= n) return;
  const int y = (int)(p / a.cols), x = (int)(p - (size_t)y * a.cols);
  const uint8_t* src = a.src + (size_t)blockIdx.z * a.src_rows * a.src_step;
  uint8_t* dst = a.dst + ((size_t)blockIdx.z * n + p) * C;
  const int X = x + a.x0, Yv = y + a.y0, Y = a.flip ? a.rh - 1 - Yv : Yv;  // pixel (Y, X) of the resized frame
  if (a.mode == 0) {
    const uint8_t* s = src + (size_t)Y * a.src_step + (size_t)X * C;
#pragma unroll
    for (int c = 0; c < C; ++c) dst[c] = s[c];
    return;
  }
  if (a.mode == 2) {  // R1, INTER_AREA fast path of an exact 2x reduction
    const uint8_t* s = src + (size_t)(2 * Y) * a.src_step + (size_t)(2 * X) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) dst[c] = (uint8_t)((s[c] + s[C + c] + s[a.src_step + c] + s[a.src_step + C + c] + 2) >> 2);
    return;
  }
  // R1, INTER_LINEAR in 8-bit fixed point
  float fx = (float)(((double)X + 0.5) * a.scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) fx = 0, sx = 0;
  if (sx >= a.src_cols - 1) fx = 0, sx = a.src_cols - 1;
  const int a0 = ingest_sat_short((1.f - fx) * 2048.f), a1 = ingest_sat_short(fx * 2048.f);
  const int sx1 = sx + 1 < a.src_cols ? sx + 1 : sx;
  float fy = (float)(((double)Y + 0.5) * a.scale_y - 0.5);
  const int sy = (int)floorf(fy);
  fy -= (float)sy;
  const int b0 = ingest_sat_short((1.f - fy) * 2048.f), b1 = ingest_sat_short(fy * 2048.f);
  const uint8_t* S0 = src + (size_t)min(max(sy, 0), a.src_rows - 1) * a.src_step;
  const uint8_t* S1 = src + (size_t)min(max(sy + 1, 0), a.src_rows - 1) * a.src_step;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int r0 = S0[sx * C + c] * a0 + S0[sx1 * C + c] * a1, r1 = S1[sx * C + c] * a0 + S1[sx1 * C + c] * a1;
    dst[c] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
  }
}

// ---- R2 cv::equalizeHist on contiguous 1-channel images: hist [images][256] (zeroed by the caller), lut [images][256]
__global__ __launch_bounds__(kBlock) void ingest_hist_kernel(const uint8_t* img, size_t n, unsigned* hist) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint8_t* im = img + (size_t)blockIdx.z * n;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) atomicAdd(&h[im[i]], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[(size_t)blockIdx.z * 256 + threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(kBlock) void ingest_lut_kernel(const unsigned* hist, uint8_t* lut, unsigned total) {
  __shared__ unsigned h[256];
  const unsigned* hi = hist + (size_t)blockIdx.x * 256;
  h[threadIdx.x] = hi[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {  // 256 sequential steps, once per image
    uint8_t* l = lut + (size_t)blockIdx.x * 256;
    int i = 0;
    while (!h[i]) ++i;
    if (h[i] == total) {
      for (int k = 0; k < 256; ++k) l[k] = (uint8_t)i;  // dst.setTo(i)
      return;
    }
    const float scale = (256 - 1.f) / (float)(total - h[i]);
    int sum = 0;
    for (int k = 0; k <= i; ++k) l[k] = 0;
    for (++i; i < 256; ++i) {
      sum += (int)h[i];
      l[i] = (uint8_t)sat_u8((float)sum * scale);
    }
  }
}

__global__ __launch_bounds__(kBlock) void ingest_apply_lut_kernel(uint8_t* img, size_t n, const uint8_t* lut) {
  __shared__ uint8_t l[256];
  l[threadIdx.x] = lut[(size_t)blockIdx.z * 256 + threadIdx.x];
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) img[(size_t)blockIdx.z * n + i] = l[img[(size_t)blockIdx.z * n + i]];
}

// ---- R3 cv::GaussianBlur(7x7, 1.5): one workgroup = a 64 x 16 output tile; the tile + 3-pixel halo goes to LDS as bytes, the row
// pass writes 22 rows of exact integer sums to LDS, the column pass rounds (sum + 2^15) >> 16.
struct BlurArgs {
  const uint8_t* src;  // [images][rows][cols][C]
  uint8_t* dst;
  int rows, cols;
  int k[7];            // integer kernel (cvRound(cf * 256))
};
constexpr int kBlurTW = 64, kBlurTH = 16, kBlurR = 3;

__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
  return p;
}

template <int C>
__global__ __launch_bounds__(kBlock) void ingest_blur7_kernel(const BlurArgs a) {
  constexpr int HW = kBlurTW + 2 * kBlurR, HH = kBlurTH + 2 * kBlurR;
  __shared__ uint8_t tile[HH][HW * C];
  __shared__ int rowsum[HH][kBlurTW * C];
  const int x0 = blockIdx.x * kBlurTW, y0 = blockIdx.y * kBlurTH;
  const size_t img = (size_t)blockIdx.z * a.rows * a.cols * C;
  for (int i = threadIdx.x; i < HH * HW; i += kBlock) {
    const int ry = i / HW, rx = i - ry * HW;
    const int y = reflect101(y0 + ry - kBlurR, a.rows), x = reflect101(x0 + rx - kBlurR, a.cols);
    const uint8_t* s = a.src + img + ((size_t)y * a.cols + x) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) tile[ry][rx * C + c] = s[c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HH * kBlurTW * C; i += kBlock) {
    const int ry = i / (kBlurTW * C), b = i - ry * (kBlurTW * C);  // b = lx * C + c
    int s = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) s += a.k[j] * tile[ry][b + j * C];
    rowsum[ry][b] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kBlurTH * kBlurTW * C; i += kBlock) {
    const int ly = i / (kBlurTW * C), b = i - ly * (kBlurTW * C), lx = b / C;
    if (x0 + lx >= a.cols || y0 + ly >= a.rows) continue;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) s += a.k[j] * rowsum[ly + j][b];
    const int v = (s + (1 << 15)) >> 16;
    a.dst[img + ((size_t)(y0 + ly) * a.cols + x0) * C + b] = (uint8_t)min(max(v, 0), 255);
  }
}

}  // namespace bgs
