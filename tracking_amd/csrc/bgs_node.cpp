// bgs_node.cpp — libbgs_node (include/bgs_node.h): one caller, all the GPUs of a node.
//
// Camera streams shard across devices in contiguous blocks with NO collective on the update path; the one exchange
// step is the gather of the bit-packed foreground masks to the root device, the one that feeds the blob detector
// (ustc_src/trackingMain.cpp:166).  xGMI is point-to-point, so the gather is what it looks like on the wire: every peer sends its block
// to the root over its own link, the root posts one receive per peer, all inside one ncclGroupStart / ncclGroupEnd per device - never
// a ring.  Per device: one engine (libbgs_hip, through its public C ABI only), one host thread, an UPDATE stream and a GATHER stream;
// two sets of mask buffers alternate so that the gather of step t runs beside the kernels of step t+1.  Ordering is by events on the
// streams, the host never waits for the device inside bgs_node_step_device:
//     update stream:  wait(gather[b] of step t-2 done) -> kernel(step t) writes buffer b -> record kernel[b]
//     gather stream:  wait(kernel[b]) -> send / receive buffer b -> record gather[b]
// The root's own kernel writes its block straight into the gather buffer (zero copy) unless BGS_NODE_LOOPBACK asks otherwise.
#include "../../include/bgs_node.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_nerr;

int nfail(int code, const char* fmt, ...) {
  char buf[640];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_nerr = buf;
  return code;
}

#define NHIP(expr)                                                                                   \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess) return nfail(BGS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)
#define NNCCL(expr)                                                                                   \
  do {                                                                                                \
    ncclResult_t r__ = (expr);                                                                        \
    if (r__ != ncclSuccess) return nfail(BGS_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(r__)); \
  } while (0)

struct Block {
  int first = 0, count = 0;
};

Block block_of(int total, int n, int index) {  // = tracking_amd/sharding.py stream_block
  const int base = total / n, extra = total % n;
  Block b;
  b.count = base + (index < extra ? 1 : 0);
  b.first = index * base + (index < extra ? index : extra);
  return b;
}

struct Dev {
  int hip_device = 0;
  int rank = 0;  // its rank in the communicator = its index among all devices of the node
  Block blk;
  bgs_engine* eng = nullptr;
  hipStream_t cs = nullptr, ms = nullptr;  // update, gather
  uint64_t* bits[2] = {nullptr, nullptr};  // [count][W]: what this device sends (unused by a root that writes in place)
  hipEvent_t ev_kernel[2] = {nullptr, nullptr}, ev_gather[2] = {nullptr, nullptr};
  bool posted[2] = {false, false};
  ncclComm_t comm = nullptr;
  // worker thread (single-process nodes with more than one device)
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  int job = 0;  // 0 idle, 1 step, 2 quit
  const void* job_frames = nullptr;
  int job_buf = 0;
  bool done = true;
  int rc = 0;
  uint32_t flags = 0;
  std::string err;
};

}  // namespace

struct bgs_node {
  bgs_algo algo;
  int total = 0, world = 1, root = 0;
  int transport = BGS_NODE_RCCL;
  unsigned flags = 0;
  bool rank_mode = false;
  int rows = 0, cols = 0, ch = 0;
  size_t W = 0;
  std::vector<Block> blocks;  // of every rank of the node
  std::vector<Dev*> devs;     // the local ones
  int root_local = -1;        // index into devs of the root device, -1 if another process owns it
  uint64_t* G[2] = {nullptr, nullptr};  // root device: [total][W], the gathered masks of the two steps in flight
  int64_t step = 0;
  int last_buf = -1;
  double enqueue_ms = 0;
  int64_t steps = 0;
};

namespace {

bool loopback(const bgs_node* n) { return (n->flags & BGS_NODE_LOOPBACK) != 0; }

// one device's share of one step: kernel on the update stream, its part of the gather on the gather stream
int dev_step(bgs_node* n, Dev* d, const void* frames, int b, uint32_t* out_flags) {
  NHIP(hipSetDevice(d->hip_device));
  const bool is_root = d->rank == n->root;
  const size_t W = n->W;
  uint64_t* out = (is_root && !loopback(n)) ? n->G[b] + (size_t)d->blk.first * W : d->bits[b];
  if (d->posted[b]) NHIP(hipStreamWaitEvent(d->cs, d->ev_gather[b], 0));  // the gather of step t-2 has finished with this buffer
  if (d->blk.count > 0) {
    uint32_t fl = 0;
    const int rc = bgs_process_batch_device(d->eng, frames, nullptr, nullptr, out, (void*)d->cs, &fl);
    if (rc) return nfail(rc, "device %d: %s", d->hip_device, bgs_last_error());
    if (out_flags) *out_flags = fl;
  } else if (out_flags) {
    *out_flags = ~0u;  // a device without streams constrains nothing
  }
  NHIP(hipEventRecord(d->ev_kernel[b], d->cs));
  NHIP(hipStreamWaitEvent(d->ms, d->ev_kernel[b], 0));
  const bool sends = (!is_root || loopback(n)) && d->blk.count > 0;
  if (n->transport == BGS_NODE_PEER_COPY) {
    if (sends) {
      const int root_dev = n->devs[(size_t)n->root_local]->hip_device;
      NHIP(hipMemcpyPeerAsync(n->G[b] + (size_t)d->blk.first * W, root_dev, d->bits[b], d->hip_device, (size_t)d->blk.count * W * 8, d->ms));
    }
  } else if (n->world > 1 || loopback(n)) {
    // point-to-point over xGMI: the root receives one block per peer (7 links side by side on an 8-GPU node), a peer sends one
    NNCCL(ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    if (is_root)
      for (int p = 0; p < n->world && r == ncclSuccess; ++p)
        if ((p != n->root || loopback(n)) && n->blocks[(size_t)p].count > 0)
          r = ncclRecv(n->G[b] + (size_t)n->blocks[(size_t)p].first * W, (size_t)n->blocks[(size_t)p].count * W, ncclUint64, p, d->comm, d->ms);
    if (sends && r == ncclSuccess) r = ncclSend(d->bits[b], (size_t)d->blk.count * W, ncclUint64, n->root, d->comm, d->ms);
    const ncclResult_t e = ncclGroupEnd();
    if (r != ncclSuccess) return nfail(BGS_ERR_HIP, "ncclSend / ncclRecv failed: %s", ncclGetErrorString(r));
    if (e != ncclSuccess) return nfail(BGS_ERR_HIP, "ncclGroupEnd failed: %s", ncclGetErrorString(e));
  }
  NHIP(hipEventRecord(d->ev_gather[b], d->ms));
  d->posted[b] = true;
  return BGS_OK;
}

void worker(bgs_node* n, Dev* d) {
  for (;;) {
    std::unique_lock<std::mutex> lk(d->mu);
    d->cv.wait(lk, [&] { return d->job != 0; });
    if (d->job == 2) return;
    const void* frames = d->job_frames;
    const int b = d->job_buf;
    lk.unlock();
    uint32_t fl = 0;
    const int rc = dev_step(n, d, frames, b, &fl);
    lk.lock();
    d->rc = rc, d->flags = fl, d->err = rc ? g_nerr : std::string();  // (the error text is thread-local: carry it over)
    d->job = 0, d->done = true;
    lk.unlock();
    d->cv.notify_all();
  }
}

void stop_workers(bgs_node* n) {
  for (Dev* d : n->devs)
    if (d->th.joinable()) {
      {
        std::lock_guard<std::mutex> lk(d->mu);
        d->job = 2;
      }
      d->cv.notify_all();
      d->th.join();
    }
}

void free_buffers(bgs_node* n) {
  for (Dev* d : n->devs) {
    (void)hipSetDevice(d->hip_device);
    for (int b = 0; b < 2; ++b)
      if (d->bits[b]) (void)hipFree(d->bits[b]), d->bits[b] = nullptr;
  }
  if (n->root_local >= 0) {
    (void)hipSetDevice(n->devs[(size_t)n->root_local]->hip_device);
    for (int b = 0; b < 2; ++b)
      if (n->G[b]) (void)hipFree(n->G[b]), n->G[b] = nullptr;
  }
}

void destroy(bgs_node* n) {
  if (!n) return;
  for (Dev* d : n->devs) {
    if (hipSetDevice(d->hip_device) != hipSuccess) continue;
    if (d->cs) (void)hipStreamSynchronize(d->cs);
    if (d->ms) (void)hipStreamSynchronize(d->ms);
  }
  stop_workers(n);
  for (Dev* d : n->devs)
    if (d->comm) (void)ncclCommDestroy(d->comm), d->comm = nullptr;
  free_buffers(n);
  for (Dev* d : n->devs) {
    (void)hipSetDevice(d->hip_device);
    if (d->eng) bgs_destroy(d->eng);
    for (int b = 0; b < 2; ++b) {
      if (d->ev_kernel[b]) (void)hipEventDestroy(d->ev_kernel[b]);
      if (d->ev_gather[b]) (void)hipEventDestroy(d->ev_gather[b]);
    }
    if (d->cs) (void)hipStreamDestroy(d->cs);
    if (d->ms) (void)hipStreamDestroy(d->ms);
    delete d;
  }
  delete n;
}

// engine, streams and events of one device
int dev_setup(bgs_node* n, Dev* d, const bgs_params* params) {
  NHIP(hipSetDevice(d->hip_device));
  if (d->blk.count > 0) {
    const int rc = bgs_create(n->algo, params, d->hip_device, d->blk.count, &d->eng);
    if (rc) return nfail(rc, "device %d: %s", d->hip_device, bgs_last_error());
  }
  NHIP(hipStreamCreateWithFlags(&d->cs, hipStreamNonBlocking));
  NHIP(hipStreamCreateWithFlags(&d->ms, hipStreamNonBlocking));
  for (int b = 0; b < 2; ++b) {
    NHIP(hipEventCreateWithFlags(&d->ev_kernel[b], hipEventDisableTiming));
    NHIP(hipEventCreateWithFlags(&d->ev_gather[b], hipEventDisableTiming));
  }
  return BGS_OK;
}

int check_common(bgs_algo algo, int total, bgs_node** out) {
  if (!out) return nfail(BGS_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if ((int)algo < 0 || algo >= BGS_ALGO_COUNT) return nfail(BGS_ERR_INVALID, "unknown algorithm %d", (int)algo);
  if (total < 1) return nfail(BGS_ERR_INVALID, "total_streams must be >= 1");
  return BGS_OK;
}

int locate(const bgs_node* n, int stream, Dev** d, int* local) {
  if (!n) return nfail(BGS_ERR_INVALID, "node is NULL");
  if (stream < 0 || stream >= n->total) return nfail(BGS_ERR_INVALID, "stream %d outside 0..%d", stream, n->total - 1);
  for (Dev* q : n->devs)
    if (stream >= q->blk.first && stream < q->blk.first + q->blk.count) {
      *d = q, *local = stream - q->blk.first;
      return BGS_OK;
    }
  return nfail(BGS_ERR_INVALID, "stream %d belongs to another rank of the node (this process serves its own block)", stream);
}

}  // namespace

extern "C" {

const char* bgs_node_last_error(void) { return g_nerr.c_str(); }

int bgs_node_stream_block(int total_streams, int n_devices, int index, int* first, int* count) {
  if (total_streams < 0 || n_devices < 1 || index < 0 || index >= n_devices) return nfail(BGS_ERR_INVALID, "bgs_node_stream_block: bad argument");
  const Block b = block_of(total_streams, n_devices, index);
  if (first) *first = b.first;
  if (count) *count = b.count;
  return BGS_OK;
}

int bgs_node_stream_owner(int total_streams, int n_devices, int stream, int* index, int* local_stream) {
  if (total_streams < 1 || n_devices < 1 || stream < 0 || stream >= total_streams) return nfail(BGS_ERR_INVALID, "bgs_node_stream_owner: bad argument");
  for (int i = 0; i < n_devices; ++i) {
    const Block b = block_of(total_streams, n_devices, i);
    if (stream >= b.first && stream < b.first + b.count) {
      if (index) *index = i;
      if (local_stream) *local_stream = stream - b.first;
      return BGS_OK;
    }
  }
  return nfail(BGS_ERR_INVALID, "bgs_node_stream_owner: stream %d has no owner", stream);
}

int bgs_node_unique_id(void* id) {
  static_assert(sizeof(ncclUniqueId) == BGS_NODE_ID_BYTES, "BGS_NODE_ID_BYTES");
  if (!id) return nfail(BGS_ERR_INVALID, "id is NULL");
  ncclUniqueId u;
  NNCCL(ncclGetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return BGS_OK;
}

int bgs_node_create(bgs_algo algo, const bgs_params* params, const int* hip_devices, int n_devices, int total_streams, int root_index, int transport,
                    unsigned flags, bgs_node** out) {
  int rc = check_common(algo, total_streams, out);
  if (rc) return rc;
  if (n_devices < 1) return nfail(BGS_ERR_INVALID, "n_devices must be >= 1");
  if (root_index < 0 || root_index >= n_devices) return nfail(BGS_ERR_INVALID, "root_index %d outside 0..%d", root_index, n_devices - 1);
  if (transport != BGS_NODE_RCCL && transport != BGS_NODE_PEER_COPY) return nfail(BGS_ERR_INVALID, "unknown transport %d", transport);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return nfail(BGS_ERR_HIP, "no HIP device visible: libbgs_node has no CPU path");
  std::vector<int> devlist((size_t)n_devices);
  for (int i = 0; i < n_devices; ++i) {
    devlist[(size_t)i] = hip_devices ? hip_devices[i] : i;
    if (devlist[(size_t)i] < 0 || devlist[(size_t)i] >= ndev) return nfail(BGS_ERR_INVALID, "hip device %d outside 0..%d", devlist[(size_t)i], ndev - 1);
    for (int j = 0; j < i; ++j)
      if (devlist[(size_t)j] == devlist[(size_t)i] && !((flags & BGS_NODE_ALLOW_DUPLICATE_DEVICES) && transport == BGS_NODE_PEER_COPY))
        return nfail(BGS_ERR_INVALID, "hip device %d listed twice (RCCL refuses two ranks on one device)", devlist[(size_t)i]);
  }
  bgs_node* n = new (std::nothrow) bgs_node();
  if (!n) return nfail(BGS_ERR_NOMEM, "out of host memory");
  n->algo = algo, n->total = total_streams, n->world = n_devices, n->root = root_index, n->transport = transport, n->flags = flags;
  n->root_local = root_index;
  for (int i = 0; i < n_devices; ++i) {
    n->blocks.push_back(block_of(total_streams, n_devices, i));
    Dev* d = new Dev();
    d->hip_device = devlist[(size_t)i], d->rank = i, d->blk = n->blocks.back();
    n->devs.push_back(d);
  }
  for (Dev* d : n->devs) {
    rc = dev_setup(n, d, params);
    if (rc) {
      destroy(n);
      return rc;
    }
  }
  if (transport == BGS_NODE_RCCL && (n_devices > 1 || loopback(n))) {
    std::vector<ncclComm_t> comms((size_t)n_devices);
    const ncclResult_t r = ncclCommInitAll(comms.data(), n_devices, devlist.data());
    if (r != ncclSuccess) {
      destroy(n);
      return nfail(BGS_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", n_devices, ncclGetErrorString(r));
    }
    for (int i = 0; i < n_devices; ++i) n->devs[(size_t)i]->comm = comms[(size_t)i];
  }
  if (transport == BGS_NODE_PEER_COPY)
    for (Dev* d : n->devs)
      if (d->hip_device != devlist[(size_t)root_index] && hipSetDevice(d->hip_device) == hipSuccess) {
        (void)hipDeviceEnablePeerAccess(devlist[(size_t)root_index], 0);  // direct xGMI writes when the topology allows; the copy works either way
        (void)hipGetLastError();
      }
  if (n_devices > 1)
    for (Dev* d : n->devs) d->th = std::thread(worker, n, d);
  *out = n;
  return BGS_OK;
}

int bgs_node_create_rank(bgs_algo algo, const bgs_params* params, int hip_device, int rank, int world, int total_streams, int root_rank, const void* id,
                         unsigned flags, bgs_node** out) {
  int rc = check_common(algo, total_streams, out);
  if (rc) return rc;
  if (world < 1 || rank < 0 || rank >= world) return nfail(BGS_ERR_INVALID, "rank %d outside 0..%d", rank, world - 1);
  if (root_rank < 0 || root_rank >= world) return nfail(BGS_ERR_INVALID, "root_rank %d outside 0..%d", root_rank, world - 1);
  if (!id && (world > 1 || (flags & BGS_NODE_LOOPBACK))) return nfail(BGS_ERR_INVALID, "id is NULL");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return nfail(BGS_ERR_HIP, "no HIP device visible: libbgs_node has no CPU path");
  if (hip_device < 0 || hip_device >= ndev) return nfail(BGS_ERR_INVALID, "hip device %d outside 0..%d", hip_device, ndev - 1);
  bgs_node* n = new (std::nothrow) bgs_node();
  if (!n) return nfail(BGS_ERR_NOMEM, "out of host memory");
  n->algo = algo, n->total = total_streams, n->world = world, n->root = root_rank, n->transport = BGS_NODE_RCCL, n->flags = flags & ~BGS_NODE_ALLOW_DUPLICATE_DEVICES;
  n->rank_mode = true;
  for (int i = 0; i < world; ++i) n->blocks.push_back(block_of(total_streams, world, i));
  Dev* d = new Dev();
  d->hip_device = hip_device, d->rank = rank, d->blk = n->blocks[(size_t)rank];
  n->devs.push_back(d);
  n->root_local = rank == root_rank ? 0 : -1;
  rc = dev_setup(n, d, params);
  if (rc) {
    destroy(n);
    return rc;
  }
  if (world > 1 || loopback(n)) {
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    const ncclResult_t r = ncclCommInitRank(&d->comm, world, u, rank);
    if (r != ncclSuccess) {
      d->comm = nullptr;
      destroy(n);
      return nfail(BGS_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r));
    }
  }
  *out = n;
  return BGS_OK;
}

int bgs_node_set_geometry(bgs_node* n, int rows, int cols, int channels) {
  if (!n) return nfail(BGS_ERR_INVALID, "node is NULL");
  if (n->W) {
    if (rows == n->rows && cols == n->cols && channels == n->ch) return BGS_OK;
    return nfail(BGS_ERR_GEOMETRY, "node is %dx%dx%d, asked for %dx%dx%d", n->rows, n->cols, n->ch, rows, cols, channels);
  }
  if (rows <= 0 || cols <= 0) return nfail(BGS_ERR_INVALID, "bad geometry %dx%d", rows, cols);
  for (Dev* d : n->devs)
    if (d->eng) {
      const int rc = bgs_set_geometry(d->eng, rows, cols, channels);
      if (rc) return nfail(rc, "device %d: %s", d->hip_device, bgs_last_error());
    }
  const size_t W = ((size_t)rows * cols + 63) / 64;
  for (Dev* d : n->devs) {
    NHIP(hipSetDevice(d->hip_device));
    const bool in_place = d->rank == n->root && !loopback(n);
    if (!in_place && d->blk.count > 0)
      for (int b = 0; b < 2; ++b) {
        NHIP(hipMalloc((void**)&d->bits[b], (size_t)d->blk.count * W * 8));
        NHIP(hipMemset(d->bits[b], 0, (size_t)d->blk.count * W * 8));
      }
  }
  if (n->root_local >= 0) {
    NHIP(hipSetDevice(n->devs[(size_t)n->root_local]->hip_device));
    for (int b = 0; b < 2; ++b) {
      NHIP(hipMalloc((void**)&n->G[b], (size_t)n->total * W * 8));
      NHIP(hipMemset(n->G[b], 0, (size_t)n->total * W * 8));
    }
    NHIP(hipDeviceSynchronize());
  }
  n->rows = rows, n->cols = cols, n->ch = channels, n->W = W;
  return BGS_OK;
}

int bgs_node_set_option(bgs_node* n, int option, int64_t value) {
  if (!n) return nfail(BGS_ERR_INVALID, "node is NULL");
  for (Dev* d : n->devs)
    if (d->eng) {
      const int rc = bgs_set_option(d->eng, option, value);
      if (rc) return nfail(rc, "device %d: %s", d->hip_device, bgs_last_error());
    }
  return BGS_OK;
}

int bgs_node_local_devices(const bgs_node* n) { return n ? (int)n->devs.size() : BGS_ERR_INVALID; }

int bgs_node_local_block(const bgs_node* n, int local_index, int* hip_device, int* first_stream, int* count) {
  if (!n || local_index < 0 || local_index >= (int)n->devs.size()) return nfail(BGS_ERR_INVALID, "bgs_node_local_block: bad argument");
  const Dev* d = n->devs[(size_t)local_index];
  if (hip_device) *hip_device = d->hip_device;
  if (first_stream) *first_stream = d->blk.first;
  if (count) *count = d->blk.count;
  return BGS_OK;
}

bgs_engine* bgs_node_engine(bgs_node* n, int local_index) {
  if (!n || local_index < 0 || local_index >= (int)n->devs.size()) return nullptr;
  return n->devs[(size_t)local_index]->eng;
}

size_t bgs_node_words_per_stream(const bgs_node* n) { return n ? n->W : 0; }

int bgs_node_is_root(const bgs_node* n) { return n && n->root_local >= 0; }

int bgs_node_step_device(bgs_node* n, const void* const* d_frames, uint32_t* out_flags) {
  if (out_flags) *out_flags = 0;
  if (!n || !d_frames) return nfail(BGS_ERR_INVALID, "NULL argument");
  if (!n->W) return nfail(BGS_ERR_INVALID, "geometry not set: call bgs_node_set_geometry first");
  for (size_t i = 0; i < n->devs.size(); ++i)
    if (!d_frames[i] && n->devs[i]->blk.count > 0) return nfail(BGS_ERR_INVALID, "d_frames[%zu] is NULL", i);
  const auto t0 = std::chrono::steady_clock::now();
  const int b = (int)(n->step & 1);
  uint32_t all = ~0u;
  int rc = BGS_OK;
  if (n->devs.size() == 1) {
    uint32_t fl = 0;
    rc = dev_step(n, n->devs[0], d_frames[0], b, &fl);
    all &= fl;
  } else {
    // every device's thread enqueues its own launches and posts its own share of the gather, side by side
    for (size_t i = 0; i < n->devs.size(); ++i) {
      Dev* d = n->devs[i];
      {
        std::lock_guard<std::mutex> lk(d->mu);
        d->job_frames = d_frames[i], d->job_buf = b, d->done = false, d->job = 1;
      }
      d->cv.notify_all();
    }
    for (Dev* d : n->devs) {
      std::unique_lock<std::mutex> lk(d->mu);
      d->cv.wait(lk, [&] { return d->done; });
      if (d->rc && !rc) rc = nfail(d->rc, "%s", d->err.c_str());
      all &= d->flags;
    }
  }
  if (rc) return rc;
  n->last_buf = b;
  n->step++;
  n->steps++;
  n->enqueue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (out_flags) *out_flags = all;
  return BGS_OK;
}

int bgs_node_collect(bgs_node* n, const uint64_t** d_masks, void* hip_stream) {
  if (d_masks) *d_masks = nullptr;
  if (!n) return nfail(BGS_ERR_INVALID, "node is NULL");
  if (n->last_buf < 0) return nfail(BGS_ERR_STATE, "bgs_node_collect: no step has been posted");
  const int b = n->last_buf;
  // RCCL: the root's own gather event covers all its receives; peer copies complete on the SENDERS' streams, so the root waits for each
  for (Dev* d : n->devs) {
    if (!d->posted[b]) continue;
    const bool mine = d->rank == n->root || n->root_local < 0 || n->transport == BGS_NODE_PEER_COPY;
    if (!mine) continue;
    if (hip_stream && n->root_local >= 0) {
      NHIP(hipSetDevice(n->devs[(size_t)n->root_local]->hip_device));
      NHIP(hipStreamWaitEvent((hipStream_t)hip_stream, d->ev_gather[b], 0));
    } else {
      NHIP(hipSetDevice(d->hip_device));
      NHIP(hipEventSynchronize(d->ev_gather[b]));
    }
  }
  if (d_masks && n->root_local >= 0) *d_masks = n->G[b];
  return BGS_OK;
}

int bgs_node_copy_masks(bgs_node* n, void* d_dst, void* hip_stream) {
  if (!n || !d_dst) return nfail(BGS_ERR_INVALID, "NULL argument");
  if (n->root_local < 0) return nfail(BGS_ERR_STATE, "bgs_node_copy_masks: this process does not own the root device");
  const uint64_t* src = nullptr;
  const int rc = bgs_node_collect(n, &src, hip_stream);  // orders hip_stream behind the gather (NULL: the host waits for it)
  if (rc) return rc;
  NHIP(hipSetDevice(n->devs[(size_t)n->root_local]->hip_device));
  NHIP(hipMemcpyAsync(d_dst, src, (size_t)n->total * n->W * 8, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
  return BGS_OK;
}

int bgs_node_sync(bgs_node* n) {
  if (!n) return nfail(BGS_ERR_INVALID, "node is NULL");
  for (Dev* d : n->devs) {
    NHIP(hipSetDevice(d->hip_device));
    NHIP(hipStreamSynchronize(d->cs));
    NHIP(hipStreamSynchronize(d->ms));
  }
  return BGS_OK;
}

int bgs_node_step_stats(bgs_node* n, double* enqueue_ms, int64_t* steps, int reset) {
  if (!n) return nfail(BGS_ERR_INVALID, "node is NULL");
  if (enqueue_ms) *enqueue_ms = n->enqueue_ms;
  if (steps) *steps = n->steps;
  if (reset) n->enqueue_ms = 0, n->steps = 0;
  return BGS_OK;
}

int bgs_node_process(bgs_node* n, int stream, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* fg, size_t fg_step, uint8_t* bg,
                     size_t bg_step, uint32_t* out_flags) {
  if (out_flags) *out_flags = 0;
  Dev* d = nullptr;
  int local = 0;
  int rc = locate(n, stream, &d, &local);
  if (rc) return rc;
  rc = bgs_process(d->eng, local, in, rows, cols, channels, in_step, fg, fg_step, bg, bg_step, out_flags);
  return rc ? nfail(rc, "device %d: %s", d->hip_device, bgs_last_error()) : BGS_OK;
}

int bgs_node_submit(bgs_node* n, int stream, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* fg, size_t fg_step, uint8_t* bg,
                    size_t bg_step) {
  Dev* d = nullptr;
  int local = 0;
  int rc = locate(n, stream, &d, &local);
  if (rc) return rc;
  rc = bgs_submit(d->eng, local, in, rows, cols, channels, in_step, fg, fg_step, bg, bg_step);
  return rc ? nfail(rc, "device %d: %s", d->hip_device, bgs_last_error()) : BGS_OK;
}

int bgs_node_wait(bgs_node* n, int stream, uint32_t* out_flags) {
  if (out_flags) *out_flags = 0;
  Dev* d = nullptr;
  int local = 0;
  int rc = locate(n, stream, &d, &local);
  if (rc) return rc;
  rc = bgs_wait(d->eng, local, out_flags);
  return rc ? nfail(rc, "device %d: %s", d->hip_device, bgs_last_error()) : BGS_OK;
}

void bgs_node_destroy(bgs_node* n) { destroy(n); }

}  // extern "C"
