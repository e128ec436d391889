/* n2m_peer.h -- peer-store exchange of the sharded optimizer (multi-GPU).
 *
 * No reference counterpart: the reference trains data-parallel through torch DDP's gradient all-reduce (nerf/utils.py:517-519,
 * nerf/provider.py:302-303).  nerf2mesh_amd shards the optimizer over the table rows instead (reduce-scatter of the
 * gradients, Adam on 1/W of the rows, all-gather of the 8-byte packed rows) and THIS interface is the exchange without a collective
 * library in the data path: every rank maps every other rank's staging buffers (hipIpcGetMemHandle / hipIpcOpenMemHandle), the table
 * backward's flush STORES each gradient row straight into the slot the row's owner keeps for the storing rank, the owner sums the W slots
 * in rank order (deterministic) and, after its Adam pass, stores its refreshed packed rows into every rank's packed table.  Hand-over is
 * by epoch counters in fine-grained memory: a signal kernel behind the producing kernel (system-scope release), a bounded-spin wait
 * kernel in front of the consuming one (system-scope acquire).  Data and flags never share a kernel, so what crosses the link is ordered
 * by kernel boundaries exactly as an in-place RCCL all-gather into a hipMalloc'ed buffer is.
 *
 * STATUS: built and tested between two PROCESSES ON ONE GPU (tests/test_parallel_gpu.py: replicas bit-identical, same bits as the
 * reduce-scatter path's rank-order sum, a rank without samples keeps the exchange in step, a missing peer times out instead of hanging).
 * NOT run over xGMI: opt-in (N2M_PEER_STORE=1), the default multi-GPU path stays RCCL.
 */
#ifndef N2M_PEER_H
#define N2M_PEER_H
#include <stddef.h>
#include <stdint.h>
#include "n2m_hip.h"      /* N2mAdamDesc (n2m_adam_step_peer) */
#ifdef __cplusplus
extern "C" {
#endif

#define N2M_PEER_MAX 8            /* ranks of one node */
#define N2M_PEER_HANDLE_BYTES 64  /* sizeof(hipIpcMemHandle_t) */

/* Device memory that can be exported: zero-filled; fine_grained != 0 -> hipExtMallocWithFlags(hipDeviceMallocFinegrained) (flags: visible to
 * other agents while a kernel runs), else hipMalloc (data: visible at kernel boundaries). */
int n2m_peer_alloc(size_t bytes, int fine_grained, void** out);
int n2m_peer_free(void* ptr);
/* handle [N2M_PEER_HANDLE_BYTES] of the allocation `ptr` is the base of; import maps it into this process (another process than the exporter's;
 * the same or another device) and returns the mapped base; unmap closes it. */
int n2m_peer_export(void* ptr, void* handle);
int n2m_peer_import(const void* handle, void** out);
int n2m_peer_unmap(void* ptr);

typedef struct {
    void* ptr[N2M_PEER_MAX];
    uint32_t count;
} N2mPeerPtrs;

/* *flags.ptr[i] = value for every i (uint32, system-scope release): "everything this stream did before is done". */
int n2m_peer_signal(const N2mPeerPtrs* flags, uint32_t value, void* stream);
/* Blocks the STREAM (one wave spins, sleeping between polls) until flags[i * stride_words] >= value for i < count (serial-number compare:
 * (int32)(flag - value) >= 0), or until timeout_ms have passed: then *error (device uint32, optional) = 1 + index of the first missing flag and
 * the stream goes on -- a dead peer never hangs the GPU.  timeout_ms = 0: 10 000. */
int n2m_peer_wait(const uint32_t* flags, uint32_t count, uint32_t stride_words, uint32_t value, uint32_t timeout_ms, uint32_t* error, void* stream);
/* The same `bytes` (multiple of 4; src NULL: zeros) stored to every destination: a rank's refreshed packed rows to all peers' tables, or
 * zeros into its gradient slots when its batch was empty. */
int n2m_peer_copy(const void* src, const N2mPeerPtrs* dst, size_t bytes, void* stream);
/* Owner side: g1[r] = sum_s stage1[s * rows + r] (fp32), g2[r, 0..1] = (half)(sum_s (float)stage2[(s * rows + r) * 2 + 0..1]), slots added in
 * rank order s = 0 .. world-1 (fp16: summed in fp32, rounded once); found_inf (optional) raised when a sum is not finite / not representable in
 * its type.  Reads bypass the caches. */
int n2m_peer_reduce_slices(const float* stage1, const void* stage2, uint32_t world, uint32_t rows, float* g1, void* g2, float* found_inf,
                           void* stream);

/* Routing of the table backward's flush (n2m_grid_encode_backward_binned_pair*, overwrite mode, partition-major path): rows [0, split_row) are
 * owned in `world` chunks of rows_c (the last chunk may be shorter: (world - 1) * rows_c < split_row <= world * rows_c -- chunks padded to a
 * multiple of four rows keep every chunk 16-byte aligned in the packed table), rows [split_row, ...) in chunks of rows_f; the gradient of absolute row a goes to
 * g1[half][owner] + local (fp32) / g2[half][owner] + 2 * local (fp16 pairs), local = the row's index inside the owner's chunk -- pointers into
 * the owners' staging slots of the calling rank (its own included).  A setting of the CALLING THREAD, consumed by its table-backward
 * calls until cleared with NULL; the grad_table arguments of those calls are not written. */
typedef struct {
    uint32_t world, split_row, rows_c, rows_f;
    float* g1[2][N2M_PEER_MAX];
    void* g2[2][N2M_PEER_MAX];
} N2mPeerRoute;
int n2m_grid_backward_peer_route(const N2mPeerRoute* route);

/* n2m_adam_step (include/n2m_hip.h) with the two exchange passes around it folded in: for a descriptor entry k with slots[k][0] != NULL the
 * gradient is the rank-order sum of the `world` staging slots slots[k][0..world) -- this rank's own memory, the other ranks have stored into it:
 * what n2m_peer_reduce_slices would have left in desc->grad[k] (fp16 gradients: summed in fp32, rounded to half once; reads bypass the caches) --
 * and every packed row the pass refreshes (shadow modes 2 / 3) is stored to packed_remote[r] + (its offset from packed_local) as well: what
 * n2m_peer_copy would have sent afterwards.  Element for element the arithmetic of the three separate passes.  Covers the sharded tables at an even
 * split (element counts multiples of 4, the paired 16-byte packed-row form); N2M_EUNSUPPORTED otherwise -- the caller keeps the separate passes.
 * Hand-over as before: n2m_peer_wait in front (the slots are complete), n2m_peer_signal behind (the rows have been stored). */
typedef struct {
    uint32_t world;
    const void* slots[N2M_ADAM_MAX][N2M_PEER_MAX];
    void* packed_local;
    void* packed_remote[N2M_PEER_MAX];
    uint32_t n_remote;
} N2mAdamPeer;
int n2m_adam_step_peer(const N2mAdamDesc* desc, double beta1, double beta2, float eps, const float* scale, const float* found_inf,
                       const float* bias, const N2mAdamPeer* peer, void* stream);

#ifdef __cplusplus
}
#endif
#endif
