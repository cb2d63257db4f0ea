// kernel_cc.h — N1: connected-component seeding of a foreground mask on the device, so that only blob rectangles
// (not full masks) cross PCIe to the blob tracker.
//
// In the reference the mask goes straight from IBGS::process into OpenCV-legacy's blob detector
// (ustc_src/trackingMain.cpp:56-57 cvCreateBlobDetectorCC/Simple, :166 DetectNewBlob), which finds the 8-connected
// foreground regions through cvFindContours and keeps their bounding rectangles; package_bgs/jmo/BlobExtraction.cpp is
// the in-tree run-length labeller of the same kind.  OpenCV legacy is not in the tree, so the contract here is the
// definition itself: components = maximal 8- (or 4-) connected sets of non-zero pixels; a component is named by the
// raster index of its first pixel ("root"); boxes come out sorted by root.
//
// Algorithm: label equivalence with union-find (the minimum index wins, so the root IS the first pixel in raster order):
//   cc_init      L[p] = head of p's horizontal run inside its wave's 64-pixel span (ballot + clz);  -1 for background
//   cc_merge     union(p, n) for the links not covered by a run: span boundaries, N, and the upper diagonals (atomicMin on the larger root)
//   cc_compress  every pixel now holds its component's root
//   cc_count / cc_scan / cc_scatter   roots -> dense ids in raster order (wave ballot + popcount prefix, two-level scan)
//   cc_boxes     atomicMin/Max/Add into the component's box, merged per lane run, per wave and per workgroup (LDS table) first
//   cc_finish    (minx, miny, maxx, maxy) -> (x, y, w, h)
#pragma once
#include "bgs_device.h"

namespace bgs {

struct CcBox {
  int x, y, w, h, area, root;  // during accumulation: x,y = min corner, w,h = max corner
};

// first and second moments of a component's pixel coordinates (image-relative): what OpenCV-legacy's blob detectors derive a
// CvBlob's centre and size from (N2, include/bgs_hip.h: bgs_moments)
struct CcMoments {
  long long sx, sy, sxx, syy;
};

constexpr int kCcPerBlock = kBlock * 4;  // pixels per workgroup in the counting kernels

__device__ __forceinline__ int cc_find(const int* L, int p) {
  int r = L[p];
  while (true) {
    const int q = L[r];
    if (q == r) return r;
    r = q;
  }
}

__device__ __forceinline__ void cc_union(int* L, int a, int b) {
  while (true) {
    a = cc_find(L, a), b = cc_find(L, b);
    if (a == b) return;
    if (a > b) {
      const int t = a;
      a = b, b = t;
    }
    const int old = atomicMin(&L[b], a);  // b was a root when we looked: make it point at the smaller root
    if (old == b) return;
    b = old;  // somebody re-rooted b in between: retry from there
  }
}

// `rows` = rows of the whole stack of images, `img_rows` = rows of one image: no link crosses an image boundary.
// Initial labels = horizontal runs: a wave covers 64 consecutive pixels; from the ballot of the foreground flags every lane
// finds the start of its run inside the wave (highest position <= lane where the "continues from the left" bit is clear), so
// every pixel starts one hop from its run head and the union-find trees stay shallow (linking N/NW first, as a plain
// neighbour rule does, builds chains as long as the image is tall).
__global__ __launch_bounds__(kBlock) void cc_init_kernel(const uint8_t* mask, int* L, int rows, int cols, int img_rows, int conn8) {
  const size_t N = (size_t)rows * cols;
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool in = p < N;
  const bool fg = in && mask[p] != 0;
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned long long m = __ballot(fg);
  const unsigned long long rowstart = __ballot(in && (p % (size_t)cols) == 0);
  const unsigned long long cont = m & (m << 1) & ~rowstart;  // bit i: pixels i-1 and i are both foreground and in the same row
  if (!in) return;
  if (!fg) {
    L[p] = -1;
    return;
  }
  const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  const int start = 63 - __clzll((long long)(~cont & upto));  // bit 0 of cont is always clear, so this is never empty
  L[p] = (int)(p - (size_t)(lane - start));
  (void)img_rows, (void)conn8;
}

__global__ __launch_bounds__(kBlock) void cc_compress_kernel(int* L, size_t N) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= N) return;
  const int l = L[p];
  if (l < 0 || l == (int)p) return;
  L[p] = cc_find(L, l);  // concurrent writers only ever store an ancestor: any value read on the way is valid
}

__global__ __launch_bounds__(kBlock) void cc_merge_kernel(int* L, int rows, int cols, int img_rows, int conn8) {
  const size_t N = (size_t)rows * cols;
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= N || L[p] < 0) return;
  const int y = (int)(p / cols), x = (int)(p - (size_t)y * cols);
  // horizontal: inside a wave's 64-pixel span (same p -> lane map as cc_init_kernel) a run already shares one label; only the
  // span's first pixel can continue a run of the previous span.  (No test on L[p] here: other lanes are re-rooting it.)
  if (x > 0 && (p & (kWave - 1)) == 0 && L[p - 1] >= 0) cc_union(L, (int)p, (int)p - 1);
  if (y % img_rows > 0) {
    // vertical: where this pixel and its W neighbour (same span, hence same label) both sit under foreground, W has already
    // made the link - only the first pixel of every overlap between a run and the run above unites them
    const bool wdone = x > 0 && (p & (kWave - 1)) != 0 && L[p - 1] >= 0 && L[p - cols - 1] >= 0;
    if (L[p - cols] >= 0 && !wdone) cc_union(L, (int)p, (int)p - cols);
    if (conn8) {
      // a diagonal neighbour is already joined through N or W/E unless that orthogonal pixel is background
      if (x > 0 && L[p - cols - 1] >= 0 && L[p - cols] < 0 && L[p - 1] < 0) cc_union(L, (int)p, (int)p - cols - 1);
      if (x + 1 < cols && L[p - cols + 1] >= 0 && L[p - cols] < 0) cc_union(L, (int)p, (int)p - cols + 1);
    }
  }
}

// number of roots in each chunk of kCcPerBlock pixels
__global__ __launch_bounds__(kBlock) void cc_count_kernel(const int* L, size_t N, int* blockCount) {
  __shared__ int wsum[kBlock / kWave];
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t p = (size_t)blockIdx.x * kCcPerBlock + k * kBlock + threadIdx.x;
    const bool root = p < N && L[p] == (int)p;
    c += __popcll(__ballot(root));  // same value in every lane of the wave
  }
  if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = c;
  __syncthreads();
  if (threadIdx.x == 0) blockCount[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of blockCount[nb] in place by one workgroup (each lane owns a contiguous slice, one scan of the 256 slice
// sums in between); total -> *count
__global__ __launch_bounds__(kBlock) void cc_scan_kernel(int* blockCount, int nb, int* count) {
  __shared__ int part[kBlock];
  const int per = (nb + kBlock - 1) / kBlock, lo = min((int)threadIdx.x * per, nb), hi = min(lo + per, nb);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += blockCount[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < kBlock; o <<= 1) {  // Hillis-Steele inclusive scan of the slice sums
    const int t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  int run = part[threadIdx.x] - sum;
  for (int i = lo; i < hi; ++i) {
    const int v = blockCount[i];
    blockCount[i] = run;
    run += v;
  }
  if (threadIdx.x == kBlock - 1) *count = part[kBlock - 1];
}

// dense id of every root, in raster order; the boxes start empty
__global__ __launch_bounds__(kBlock) void cc_scatter_kernel(const int* L, size_t N, const int* blockOffset, int* id, CcBox* boxes, int max_boxes, int* perImage, size_t imgN) {
  __shared__ int wbase[4][kBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  bool root[4];
  unsigned long long bal[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t p = (size_t)blockIdx.x * kCcPerBlock + k * kBlock + threadIdx.x;
    root[k] = p < N && L[p] == (int)p;
    bal[k] = __ballot(root[k]);
    if (lane == 0) wbase[k][wave] = __popcll(bal[k]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // 16 wave counts -> exclusive offsets in pixel order (k major, wave minor)
    int run = blockOffset[blockIdx.x];
    for (int k = 0; k < 4; ++k)
      for (int w = 0; w < kBlock / kWave; ++w) {
        const int c = wbase[k][w];
        wbase[k][w] = run;
        run += c;
      }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!root[k]) continue;
    const size_t p = (size_t)blockIdx.x * kCcPerBlock + k * kBlock + threadIdx.x;
    const int my = wbase[k][wave] + __popcll(bal[k] & ((1ull << lane) - 1ull));
    id[p] = my;
    if (my < max_boxes) boxes[my] = CcBox{0x7fffffff, 0x7fffffff, -1, -1, 0, (int)p};
    if (perImage) atomicAdd(&perImage[p / imgN + 1], 1);  // components of image k are counted in slot k+1 (prefix-summed later)
  }
}

// Boxes: a wave walks 1024 consecutive pixels (16 per lane, coalesced).  A lane keeps one open accumulator and flushes it
// when its component changes; at the end the wave merges the open accumulators of lanes that hold the same component (one
// butterfly per distinct component).  Flushes go to a 128-entry table in LDS (open addressing on the component id, LDS
// atomics), and the workgroup issues one set of global atomics per component it met - so a mask that is mostly one giant
// component (a model's first frames) no longer funnels every flush through the same five global addresses.
constexpr int kCcBoxPer = 16, kCcSlots = 128;

__device__ __forceinline__ void cc_box_atomics(CcBox* b, int mnx, int mny, int mxx, int mxy, int cnt) {
  atomicMin(&b->x, mnx), atomicMin(&b->y, mny), atomicMax(&b->w, mxx), atomicMax(&b->h, mxy), atomicAdd(&b->area, cnt);
}

struct CcTable {
  int id[kCcSlots];
  int box[kCcSlots][5];
};

__device__ __forceinline__ void cc_table_add(CcTable& t, CcBox* boxes, int id, int mnx, int mny, int mxx, int mxy, int cnt) {
  int slot = (int)(((unsigned)id * 2654435761u) >> 25);  // 7 bits
  for (int probe = 0; probe < 8; ++probe) {
    const int old = atomicCAS(&t.id[slot], -1, id);
    if (old == -1 || old == id) {
      atomicMin(&t.box[slot][0], mnx), atomicMin(&t.box[slot][1], mny), atomicMax(&t.box[slot][2], mxx), atomicMax(&t.box[slot][3], mxy);
      atomicAdd(&t.box[slot][4], cnt);
      return;
    }
    slot = (slot + 1) & (kCcSlots - 1);
  }
  cc_box_atomics(boxes + id, mnx, mny, mxx, mxy, cnt);  // table crowded around this hash: go straight to memory
}

// `moments` (optional, zeroed by the caller): coordinate sums per component.  They skip the LDS table: one set of four 64-bit
// global atomics per (wave, component) - a wave meets few components, and a mask that is one giant component still only
// issues N/1024 of them per address.
__global__ __launch_bounds__(kBlock) void cc_boxes_kernel(const int* L, const int* id, int rows, int cols, int img_rows, CcBox* boxes, int max_boxes, CcMoments* moments) {
  __shared__ CcTable tab;
  if (threadIdx.x < kCcSlots) {
    tab.id[threadIdx.x] = -1;
    tab.box[threadIdx.x][0] = tab.box[threadIdx.x][1] = 0x7fffffff;
    tab.box[threadIdx.x][2] = tab.box[threadIdx.x][3] = -1;
    tab.box[threadIdx.x][4] = 0;
  }
  __syncthreads();
  const size_t N = (size_t)rows * cols;
  const int lane = threadIdx.x & (kWave - 1);
  const size_t wave0 = ((size_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave) * (kWave * kCcBoxPer);
  int cid = -1, cl = -1, mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -1, mxy = -1, cnt = 0;
  unsigned long long sx = 0, sy = 0, sxx = 0, syy = 0;
  auto flush_moments = [&](int c, unsigned long long a, unsigned long long b, unsigned long long q, unsigned long long r) {
    CcMoments* m = moments + c;
    atomicAdd((unsigned long long*)&m->sx, a), atomicAdd((unsigned long long*)&m->sy, b);
    atomicAdd((unsigned long long*)&m->sxx, q), atomicAdd((unsigned long long*)&m->syy, r);
  };
  for (int k = 0; k < kCcBoxPer; ++k) {
    const size_t p = wave0 + (size_t)k * kWave + lane;
    const int l = p < N ? L[p] : -1;
    if (l < 0) continue;
    if (l != cl) {  // a different component than the one this lane has open
      if (cid >= 0 && cid < max_boxes) {
        cc_table_add(tab, boxes, cid, mnx, mny, mxx, mxy, cnt);
        if (moments) flush_moments(cid, sx, sy, sxx, syy);
      }
      cl = l, cid = id[l];
      mnx = mny = 0x7fffffff, mxx = mxy = -1, cnt = 0;
      sx = sy = sxx = syy = 0;
    }
    const int yy = (int)(p / cols), x = (int)(p - (size_t)yy * cols), y = yy % img_rows;
    mnx = min(mnx, x), mny = min(mny, y), mxx = max(mxx, x), mxy = max(mxy, y), cnt++;
    if (moments) sx += (unsigned)x, sy += (unsigned)y, sxx += (unsigned long long)x * (unsigned)x, syy += (unsigned long long)y * (unsigned)y;
  }
  if (cid >= max_boxes) cid = -1;
  unsigned long long pending = __ballot(cid >= 0);
  while (pending) {  // wave-uniform
    const int leader = __ffsll((long long)pending) - 1;
    const int lid = __shfl(cid, leader);
    const bool mine = cid == lid;
    int a = mine ? mnx : 0x7fffffff, b = mine ? mny : 0x7fffffff, c = mine ? mxx : -1, d = mine ? mxy : -1, e = mine ? cnt : 0;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
      a = min(a, __shfl_xor(a, o)), b = min(b, __shfl_xor(b, o));
      c = max(c, __shfl_xor(c, o)), d = max(d, __shfl_xor(d, o));
      e += __shfl_xor(e, o);
    }
    if (lane == leader) cc_table_add(tab, boxes, lid, a, b, c, d, e);
    if (moments) {  // wave-uniform
      unsigned long long m0 = mine ? sx : 0, m1 = mine ? sy : 0, m2 = mine ? sxx : 0, m3 = mine ? syy : 0;
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) {
        m0 += (unsigned long long)__shfl_xor((long long)m0, o), m1 += (unsigned long long)__shfl_xor((long long)m1, o);
        m2 += (unsigned long long)__shfl_xor((long long)m2, o), m3 += (unsigned long long)__shfl_xor((long long)m3, o);
      }
      if (lane == leader) flush_moments(lid, m0, m1, m2, m3);
    }
    pending &= ~__ballot(mine);
  }
  __syncthreads();
  if (threadIdx.x < kCcSlots && tab.id[threadIdx.x] >= 0)
    cc_box_atomics(boxes + tab.id[threadIdx.x], tab.box[threadIdx.x][0], tab.box[threadIdx.x][1], tab.box[threadIdx.x][2], tab.box[threadIdx.x][3], tab.box[threadIdx.x][4]);
}

__global__ __launch_bounds__(kBlock) void cc_finish_kernel(CcBox* boxes, const int* count, int max_boxes, int imgN) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= min(*count, max_boxes)) return;
  CcBox b = boxes[i];
  b.w = b.w - b.x + 1, b.h = b.h - b.y + 1;
  b.root %= imgN;  // raster index inside its own image
  boxes[i] = b;
}

// batch: per-image counts (slots 1..images) -> offsets[0..images]; labels -> roots relative to their image
__global__ void cc_offsets_kernel(int* offsets, int images) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int run = 0;
    offsets[0] = 0;
    for (int k = 1; k <= images; ++k) run += offsets[k], offsets[k] = run;
  }
}

__global__ __launch_bounds__(kBlock) void cc_localize_kernel(int* L, size_t N, size_t imgN) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= N) return;
  const int l = L[p];
  if (l >= 0) L[p] = (int)((size_t)l - (p / imgN) * imgN);
}

}  // namespace bgs
