/* viennaray_amd_rccl.h — RCCL (xGMI) collective for vr_apply_sharded, in a library of its own
 * (viennaray_amd/libviennaray_amd_rccl.so) so that the tracer itself has no RCCL dependency.
 *
 *   one process per GPU:
 *     vr_rccl_comm *comm;
 *     vr_rccl_unique_id(id);                       // rank 0, then broadcast the 128 bytes (MPI, file, socket)
 *     vr_rccl_init_rank(&comm, id, rank, world);   // every rank, after hipSetDevice
 *     vr_apply_sharded(ctx, rank, world, vr_rccl_allreduce, comm);
 *     vr_rccl_destroy(comm);
 */
#ifndef VIENNARAY_AMD_RCCL_H
#define VIENNARAY_AMD_RCCL_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct vr_rccl_comm vr_rccl_comm;
#define VR_RCCL_UNIQUE_ID_BYTES 128
int vr_rccl_unique_id(char id[VR_RCCL_UNIQUE_ID_BYTES]);
int vr_rccl_init_rank(vr_rccl_comm **out, const char id[VR_RCCL_UNIQUE_ID_BYTES], int rank, int world);
void vr_rccl_destroy(vr_rccl_comm *comm);
/* a vr_allreduce_fn (include/viennaray_amd.h): ncclAllReduce(int64, sum) in place on `hipStream`; user = vr_rccl_comm* */
int vr_rccl_allreduce(void *user, void *devInt64, size_t count, void *hipStream);
const char *vr_rccl_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
