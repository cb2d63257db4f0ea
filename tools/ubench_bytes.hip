// ubench_bytes.hip — what is the ceiling of a "two interleaved-BGR byte streams in, one mask byte per pixel out" kernel on MI355X,
// and which access shape reaches it?  (Round-2 study behind the byte-stream kernels of kernel_pointwise.h: FrameDifference sits at
// 0.70 of the 8 TB/s peak with next to no arithmetic, a float4 copy at 0.79.)  Not product code; build: make -C tools ubench
//   E  reference: a perfectly coalesced copy-like kernel with the same bytes (16 B per lane per access, 6 B/px in, 1 B/px out)
//   A  product shape: lane owns 16 pixels = 48 contiguous bytes per input (3 x dwordx4 at a 48-byte lane stride)
//   A4 product shape, 4 pixels per lane (3 x dword at a 12-byte lane stride)
//   B  coalesced rows + LDS transpose: the wave loads 3 KB per input as 3 coalesced dwordx4 rows, stages them in LDS, then each
//      lane reads its own 48 bytes back
//   P  A as 1024-lane persistent workgroups
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ int gray(int b, int g, int r) { return (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14; }
__device__ __forceinline__ uint32_t absdiff4(uint32_t x, uint32_t y) {  // per-byte |x - y|
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int a = (x >> (8 * j)) & 255, b = (y >> (8 * j)) & 255;
    r |= (uint32_t)abs(a - b) << (8 * j);
  }
  return r;
}
// 16 pixels (12 dwords of per-byte differences) -> 16 mask bytes
__device__ __forceinline__ uint4 mask16(const uint32_t (&d)[12], int thr) {
  uint32_t m[4] = {0, 0, 0, 0};
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = 3 * p + k;
      c[k] = (d[i >> 2] >> (8 * (i & 3))) & 255;
    }
    m[p >> 2] |= (gray(c[0], c[1], c[2]) > thr ? 255u : 0u) << (8 * (p & 3));
  }
  return make_uint4(m[0], m[1], m[2], m[3]);
}

// E: same bytes, ideal shape: per 16 pixels one lane moves 48 + 48 bytes in and 16 out, all as wave-contiguous 16-byte pieces
__global__ __launch_bounds__(256) void kE(const uint4* __restrict__ x, const uint4* __restrict__ y, uint4* __restrict__ m, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // one lane per 16 pixels
  if (i >= n16) return;
  const size_t wave = i >> 6, lane = i & 63;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint4 a = x[wave * 192 + k * 64 + lane], b = y[wave * 192 + k * 64 + lane];
    acc.x ^= a.x ^ b.x, acc.y ^= a.y ^ b.y, acc.z ^= a.z ^ b.z, acc.w ^= a.w ^ b.w;
  }
  m[i] = acc;
}

__global__ __launch_bounds__(256) void kA(const uint8_t* __restrict__ x, const uint8_t* __restrict__ y, uint8_t* __restrict__ m, size_t n16, int thr) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const uint4* px = reinterpret_cast<const uint4*>(x + i * 48);
  const uint4* py = reinterpret_cast<const uint4*>(y + i * 48);
  uint32_t d[12];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint4 a = px[k], b = py[k];
    d[4 * k] = absdiff4(a.x, b.x), d[4 * k + 1] = absdiff4(a.y, b.y), d[4 * k + 2] = absdiff4(a.z, b.z), d[4 * k + 3] = absdiff4(a.w, b.w);
  }
  reinterpret_cast<uint4*>(m)[i] = mask16(d, thr);
}

__global__ __launch_bounds__(256) void kA4(const uint8_t* __restrict__ x, const uint8_t* __restrict__ y, uint8_t* __restrict__ m, size_t n4, int thr) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // one lane per 4 pixels
  if (i >= n4) return;
  const uint32_t* px = reinterpret_cast<const uint32_t*>(x + i * 12);
  const uint32_t* py = reinterpret_cast<const uint32_t*>(y + i * 12);
  uint32_t d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = absdiff4(px[k], py[k]);
  uint32_t mm = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = 3 * p + k;
      c[k] = (d[j >> 2] >> (8 * (j & 3))) & 255;
    }
    mm |= (gray(c[0], c[1], c[2]) > thr ? 255u : 0u) << (8 * p);
  }
  reinterpret_cast<uint32_t*>(m)[i] = mm;
}

// B: coalesced rows, LDS transpose.  Per wave: 3 KB of per-byte differences staged in LDS, read back as 48 bytes per lane.
// LDS row stride per lane 52 bytes (13 dwords) so the ds_read_b128 pieces of neighbouring lanes do not collide.
__global__ __launch_bounds__(256) void kB(const uint4* __restrict__ x, const uint4* __restrict__ y, uint8_t* __restrict__ m, size_t n16, int thr) {
  __shared__ uint32_t lds[4][64 * 13];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t wave = i >> 6;
  if (wave * 64 >= n16) return;  // whole waves only (n16 % 64 == 0 in this bench)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint4 a = x[wave * 192 + k * 64 + lane], b = y[wave * 192 + k * 64 + lane];
    // this piece is bytes [k*1024 + 16*lane, +16) of the wave's 3072: dword index q = k*256 + 4*lane .. +3 -> owner lane q / 12, slot q % 12
    const uint32_t dd[4] = {absdiff4(a.x, b.x), absdiff4(a.y, b.y), absdiff4(a.z, b.z), absdiff4(a.w, b.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = k * 256 + 4 * lane + j;
      lds[w][(q / 12) * 13 + (q % 12)] = dd[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  uint32_t d[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) d[j] = lds[w][lane * 13 + j];
  reinterpret_cast<uint4*>(m)[i] = mask16(d, thr);
}

__global__ __launch_bounds__(1024) void kP(const uint8_t* __restrict__ x, const uint8_t* __restrict__ y, uint8_t* __restrict__ m, size_t n16, int thr) {
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) {
    const uint4* px = reinterpret_cast<const uint4*>(x + i * 48);
    const uint4* py = reinterpret_cast<const uint4*>(y + i * 48);
    uint32_t d[12];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint4 a = px[k], b = py[k];
      d[4 * k] = absdiff4(a.x, b.x), d[4 * k + 1] = absdiff4(a.y, b.y), d[4 * k + 2] = absdiff4(a.z, b.z), d[4 * k + 3] = absdiff4(a.w, b.w);
    }
    reinterpret_cast<uint4*>(m)[i] = mask16(d, thr);
  }
}

int main() {
  const size_t S = 8, npix = S * 3840 * 2160, n16 = npix / 16;
  const int T = 6;  // rotate buffers so that nothing is served from the 256 MiB Infinity Cache
  std::vector<uint8_t*> X(T), Y(T), M(T);
  for (int t = 0; t < T; ++t) {
    CK(hipMalloc(&X[t], npix * 3));
    CK(hipMalloc(&Y[t], npix * 3));
    CK(hipMalloc(&M[t], npix));
    CK(hipMemset(X[t], 17 * t + 3, npix * 3));
    CK(hipMemset(Y[t], 5 * t + 1, npix * 3));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = 7.0 * npix;
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 12; ++i) launch(i % T);
    CK(hipEventRecord(e0, 0));
    const int N = 60;
    for (int i = 0; i < N; ++i) launch(i % T);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= N;
    printf("%-44s %.4f ms  %7.1f GB/s (7 B/px) = %.1f%% of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
  };
  const unsigned b16 = (unsigned)((n16 + 255) / 256), b4 = (unsigned)((npix / 4 + 255) / 256);
  for (int rep = 0; rep < 2; ++rep) {
    run("E  ideal shape (coalesced 16 B pieces)", [&](int t) { hipLaunchKernelGGL(kE, dim3(b16), dim3(256), 0, 0, (const uint4*)X[t], (const uint4*)Y[t], (uint4*)M[t], n16); });
    run("A  lane owns 48 B (3 x dwordx4, stride 48)", [&](int t) { hipLaunchKernelGGL(kA, dim3(b16), dim3(256), 0, 0, X[t], Y[t], M[t], n16, 15); });
    run("A4 lane owns 12 B (3 x dword, stride 12)", [&](int t) { hipLaunchKernelGGL(kA4, dim3(b4), dim3(256), 0, 0, X[t], Y[t], M[t], npix / 4, 15); });
    run("B  coalesced rows + LDS transpose", [&](int t) { hipLaunchKernelGGL(kB, dim3(b16), dim3(256), 0, 0, (const uint4*)X[t], (const uint4*)Y[t], M[t], n16, 15); });
    for (int g : {256, 512, 1024, 2048})
      run((std::string("P  A, persistent 1024-lane workgroups, grid ") + std::to_string(g)).c_str(),
          [&](int t) { hipLaunchKernelGGL(kP, dim3(g), dim3(1024), 0, 0, X[t], Y[t], M[t], n16, 15); });
  }
  return 0;
}
