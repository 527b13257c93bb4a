/*
 * llenv_xfer.h -- C ABI of the CU-free trajectory hand-off: the learner rank PULLS the other ranks' finished unroll blocks out of their
 * HBM with peer-to-peer DMA copies, instead of every rank pushing through a collective that keeps workgroups resident.
 *
 * Replaces the actor's push of one unroll to the learner, learning/actors/distill_actor.py:159-167 (ZeroMQ `_data_server.send`), for
 * the ranks of one node -- as an alternative transport to the RCCL gather of gather.py (which stays the default: BASELINE north_star).
 * Why it exists: at 4096 envs the step kernel is one 512-register wavefront per SIMD; nothing shares a SIMD with it, so whatever a
 * collective keeps resident on the compute units is paid by the step launches in full (profiles/r03_simd_sharing.txt).  A copy with
 * hipMemcpyDeviceToDeviceNoCU runs on the SDMA engines and occupies no compute unit.
 *
 * Mechanism: HIP IPC.  A rank exports (ll_xfer_export_mem) the allocation that holds its unroll blocks and an interprocess event
 * (ll_xfer_event_create); the 64-byte handles travel over any host channel (gather.py: torch.distributed object collectives); the learner
 * rank opens them (ll_xfer_open_mem, ll_xfer_event_open).  Per unroll the producer records its event on its engine's stream behind the
 * kernels that wrote the block (ll_xfer_event_record); the learner's copy stream waits for that event (ll_xfer_stream_wait) and pulls
 * (ll_xfer_pull).  Ordering of record-before-wait between processes is the caller's business (one host barrier per unroll, gather.py).
 *
 * Conventions as in llenv.h: int return codes (LL_OK / LL_E*), ll_last_error(); handles are plain bytes; streams and events are the HIP
 * objects as void*; every call acts on `device` (hipSetDevice) where it takes one.
 */
#ifndef LLENV_XFER_H
#define LLENV_XFER_H

#include <stddef.h>
#include <stdint.h>

#include "llenv.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LL_XFER_HANDLE_BYTES 64          /* sizeof(hipIpcMemHandle_t) == sizeof(hipIpcEventHandle_t) */

typedef struct { unsigned char bytes[LL_XFER_HANDLE_BYTES]; } ll_xfer_handle;

/* Handle of the device allocation that contains d_ptr, and d_ptr's offset inside it (IPC exports whole allocations). */
int ll_xfer_export_mem(int device, const void* d_ptr, ll_xfer_handle* out, uint64_t* offset);
/* Map another process's allocation; *d_ptr_out = its base + offset.  The mapping lives until ll_xfer_close_mem(*d_ptr_out - offset). */
int ll_xfer_open_mem(int device, const ll_xfer_handle* h, uint64_t offset, void** d_ptr_out);
int ll_xfer_close_mem(int device, void* d_base);

/* An interprocess event (hipEventInterprocess | hipEventDisableTiming) and the handle other processes open it with. */
int ll_xfer_event_create(int device, void** event_out, ll_xfer_handle* out);
int ll_xfer_event_open(int device, const ll_xfer_handle* h, void** event_out);
int ll_xfer_event_destroy(void* event);
int ll_xfer_event_record(void* event, void* hip_stream);
int ll_xfer_event_synchronize(void* event);
/* hip_stream will not run anything queued after this call before the work captured by the LAST record of `event` has finished */
int ll_xfer_stream_wait(void* hip_stream, void* event);

/* A stream of the learner rank's own for the pulls, so that they are ordered against nothing of the engine's. */
int ll_xfer_stream_create(int device, void** stream_out);
int ll_xfer_stream_destroy(void* hip_stream);
int ll_xfer_stream_synchronize(void* hip_stream);
/* d_dst <- d_src, asynchronous on hip_stream.  no_cu != 0: hipMemcpyDeviceToDeviceNoCU (SDMA engines, no compute unit);
 * no_cu == 0: the runtime's default device-to-device path (a copy kernel when source and destination share a device): the A/B leg. */
int ll_xfer_pull(void* d_dst, const void* d_src, size_t bytes, int no_cu, void* hip_stream);


/* ---- stream-ordered hand-shake on a 32-bit word (round 5): takes the HOST out of the producer's side of the hand-off.  ROCm implements a
 * stream wait on an INTERPROCESS event as a host-side wait (the call returns when the event has completed), so a producer that must not overwrite
 * block k - 1 before the learner has copied it used to block its launching thread.  Instead: a word of signal memory of the producer's own
 * (hipExtMallocWithFlags(hipMallocSignalMemory)); the engine's stream waits on the DEVICE until the word has reached the unroll number
 * (ll_xfer_stream_wait_value: hipStreamWaitValue32, >=), and a helper thread -- the only one that ever blocks on the interprocess event -- raises it
 * from a stream of its own (ll_xfer_stream_write_value).  The launching thread queues and goes on. */
int ll_xfer_can_wait_value(int device, int* yes);                                  /* hipDeviceAttributeCanUseStreamWaitValue */
int ll_xfer_signal_create(int device, void** d_word_out);                          /* one zeroed 64-bit word of signal memory */
int ll_xfer_signal_destroy(int device, void* d_word);
int ll_xfer_stream_wait_value(void* hip_stream, void* d_word, uint32_t value);    /* nothing queued on hip_stream afterwards runs before *d_word >= value */
int ll_xfer_stream_write_value(void* hip_stream, void* d_word, uint32_t value);   /* *d_word = value, in stream order */
int ll_xfer_set_device(int device);                                                /* hipSetDevice for the CALLING thread (helper threads start on device 0) */

#ifdef __cplusplus
}
#endif
#endif
