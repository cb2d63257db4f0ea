/*
 * bgs_node.h — C ABI of libbgs_node: ONE caller, ALL the GPUs of a node (BASELINE configs[4]).
 *
 * The reference's callers are C++ objects that own their cameras - FrameProcessor::process hands one frame to every enabled IBGS
 * (FrameProcessor.cpp:157-167), USTC_BGS::Process feeds one IBGS per tracker (ustc_src/ustc_bgs.cpp:87-113, called from
 * ustc_src/trackingMain.cpp:166) - and know nothing about devices.  A bgs_node is what such a caller holds instead of one engine when
 * it owns more cameras than one GPU serves: camera streams are independent units (no halo, no shared state), so they shard across
 * the devices in CONTIGUOUS BLOCKS - stream s lives on device s / streams_per_device, each device's masks form one buffer - with no
 * collective on the update path.  The one exchange step is the hand-off of the bit-packed foreground masks to the device that feeds
 * the blob detector (CvBlobDetector consumes the mask right after FG detection, trackingMain.cpp:166): root posts one ncclRecv per
 * peer, every peer one ncclSend, inside ncclGroupStart / ncclGroupEnd (RCCL over xGMI: 7 transfers over 7 distinct links, per-link
 * bound, not a ring), on a second HIP stream per device, double-buffered, so the gather of step t overlaps the kernels of step t+1.
 *
 * libbgs_node sits strictly ABOVE include/bgs_hip.h: it creates one bgs_engine per device and drives it through the public calls.
 * No OpenCV, HIP, RCCL or torch types in any signature; 0 or a negative bgs_status; text of a failure in bgs_last_error()
 * (failures inside libbgs_hip) or bgs_node_last_error() (failures of this layer: RCCL, threads, arguments).
 */
#ifndef BGS_NODE_H
#define BGS_NODE_H

#include "bgs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bgs_node bgs_node;

/* The stream -> device map, as a pure function: device `index` of `n_devices` owns the global streams [*first, *first + *count) of
 * `total_streams` (contiguous blocks; the first total % n devices own one stream more).  No GPU needed. */
int bgs_node_stream_block(int total_streams, int n_devices, int index, int* first, int* count);
/* ... and its inverse: which device owns global stream `stream`, and as which of its local streams. */
int bgs_node_stream_owner(int total_streams, int n_devices, int stream, int* index, int* local_stream);

/* How the packed masks travel to the root device. */
#define BGS_NODE_RCCL 0        /* ncclSend / ncclRecv grouped (default; what SURVEY.md §8e and BASELINE.json's north_star name) */
#define BGS_NODE_PEER_COPY 1   /* hipMemcpyPeerAsync from each device into the root's buffer: single-process nodes only; no RCCL call at all */
/* flags of bgs_node_create* */
#define BGS_NODE_LOOPBACK 1u   /* the root's own block also goes through the transport (RCCL: a send to itself matched by a receive from
                                  itself) instead of being written in place by its kernel: exercises the collective calls where only
                                  one GPU is available; not for production */
#define BGS_NODE_ALLOW_DUPLICATE_DEVICES 2u /* test rigs: the same HIP device may appear several times in hip_devices (BGS_NODE_PEER_COPY only:
                                  RCCL refuses two ranks on one device) */

/*
 * SINGLE-PROCESS node: this process drives all `n_devices` devices (hip_devices NULL = 0 .. n_devices-1).  One engine, one host
 * thread and two HIP streams (update, gather) per device; ncclCommInitAll over the device list.  `root_index` = the entry of
 * hip_devices that receives the masks.  The geometry is fixed by bgs_node_set_geometry (or the first bgs_node_process).
 */
int bgs_node_create(bgs_algo algo, const bgs_params* params, const int* hip_devices, int n_devices, int total_streams, int root_index,
                    int transport, unsigned flags, bgs_node** out);

/*
 * ONE PROCESS PER GPU (how torch.distributed.run / mpirun launch a job): every rank calls bgs_node_create_rank with the same
 * 128-byte id, which rank 0 obtained from bgs_node_unique_id and handed to the others by whatever means the job has (a file, a
 * TCP store, MPI_Bcast ...).  ncclCommInitRank; the rank drives the one device `hip_device`.
 */
#define BGS_NODE_ID_BYTES 128
int bgs_node_unique_id(void* id);
int bgs_node_create_rank(bgs_algo algo, const bgs_params* params, int hip_device, int rank, int world, int total_streams, int root_rank,
                         const void* id, unsigned flags, bgs_node** out);

int bgs_node_set_geometry(bgs_node* n, int rows, int cols, int channels);
int bgs_node_set_option(bgs_node* n, int option, int64_t value); /* bgs_set_option on every local engine */

/* what this process drives */
int bgs_node_local_devices(const bgs_node* n);                                         /* 1 for a rank node */
int bgs_node_local_block(const bgs_node* n, int local_index, int* hip_device, int* first_stream, int* count);
bgs_engine* bgs_node_engine(bgs_node* n, int local_index);                             /* for bgs_get_state, bgs_enable_kernel_timing ...; owned by the node */
size_t bgs_node_words_per_stream(const bgs_node* n);                                   /* ceil(rows*cols / 64); 0 before the geometry is known */
int bgs_node_is_root(const bgs_node* n);                                               /* 1 when this process owns the root device */

/*
 * One frame of every LOCAL stream, device buffers: d_frames[i] is a pointer ON local device i to [count_i][rows][cols][channels]
 * uint8.  Each device's thread launches bgs_process_batch_device on its engine (bit-packed masks straight into this step's gather
 * buffer) and posts this step's share of the gather on the device's second stream; the call returns when everything is enqueued.
 * out_flags (may be NULL): AND of the engines' out_flags.
 */
int bgs_node_step_device(bgs_node* n, const void* const* d_frames, uint32_t* out_flags);
/*
 * Wait until the gather of the most recent step is complete.  On the process that owns the root device *d_masks is a pointer ON THE
 * ROOT DEVICE to [total_streams][words_per_stream] uint64 in global stream order - the masks of that step - valid until the step
 * after the next one is posted (two buffers alternate); elsewhere NULL.  `hip_stream` (void* = hipStream_t, may be NULL): instead of
 * blocking the host, make that stream of the root device wait for the gather (the blob kernels are enqueued behind it).
 */
int bgs_node_collect(bgs_node* n, const uint64_t** d_masks, void* hip_stream);
/* The same gathered masks copied into the caller's buffer d_dst ([total_streams][words_per_stream] uint64 on the root device),
 * asynchronously on `hip_stream` (NULL = the default stream) and ordered behind the gather: for a consumer that keeps a step's masks
 * longer than the two alternating buffers allow.  BGS_ERR_STATE on a process that does not own the root device. */
int bgs_node_copy_masks(bgs_node* n, void* d_dst, void* hip_stream);
/* drain every stream of every local device */
int bgs_node_sync(bgs_node* n);
/* host time (ms) this process has spent inside bgs_node_step_device (launching kernels and posting the gather: nothing in it waits
 * for the device - buffer reuse is ordered by events on the streams), and the steps posted, since creation or the last call with
 * reset != 0: the diagnostics of a scaling run */
int bgs_node_step_stats(bgs_node* n, double* enqueue_ms, int64_t* steps, int reset);

/*
 * The host path over a node: IBGS::process for camera `stream` (GLOBAL id) - routed to the engine of the device that owns it, as its
 * local stream; arguments and results of bgs_process / bgs_submit / bgs_wait.  Single-process nodes only (a rank node serves its own
 * block: streams outside it are BGS_ERR_INVALID).  A C++ FrameProcessor that owns 256 cameras holds ONE bgs_node.
 */
int bgs_node_process(bgs_node* n, int stream, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* fg, size_t fg_step,
                     uint8_t* bg, size_t bg_step, uint32_t* out_flags);
int bgs_node_submit(bgs_node* n, int stream, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* fg, size_t fg_step,
                    uint8_t* bg, size_t bg_step);
int bgs_node_wait(bgs_node* n, int stream, uint32_t* out_flags);

void bgs_node_destroy(bgs_node* n);
const char* bgs_node_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* BGS_NODE_H */
