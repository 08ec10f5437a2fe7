// vr_rccl.cpp — the RCCL all-reduce callback of vr_apply_sharded (include/viennaray_amd_rccl.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "../../include/viennaray_amd_rccl.h"

struct vr_rccl_comm {
  ncclComm_t comm = nullptr;
};

static thread_local std::string g_err;

extern "C" {

const char *vr_rccl_last_error(void) { return g_err.c_str(); }

int vr_rccl_unique_id(char id[VR_RCCL_UNIQUE_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= VR_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId grew");
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) {
    g_err = ncclGetErrorString(r);
    return -1;
  }
  std::memset(id, 0, VR_RCCL_UNIQUE_ID_BYTES);
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int vr_rccl_init_rank(vr_rccl_comm **out, const char id[VR_RCCL_UNIQUE_ID_BYTES], int rank, int world) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) {
    g_err = "vr_rccl_init_rank: bad argument (out, id, 0 <= rank < world)";
    return -1;
  }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  vr_rccl_comm *c = new vr_rccl_comm();
  ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    g_err = ncclGetErrorString(r);
    delete c;
    return -1;
  }
  *out = c;
  return 0;
}

void vr_rccl_destroy(vr_rccl_comm *c) {
  if (!c)
    return;
  if (c->comm)
    ncclCommDestroy(c->comm);
  delete c;
}

int vr_rccl_allreduce(void *user, void *devInt64, size_t count, void *hipStream) {
  vr_rccl_comm *c = static_cast<vr_rccl_comm *>(user);
  if (!c || !c->comm) {
    g_err = "vr_rccl_allreduce: no communicator (vr_rccl_init_rank)";
    return -1;
  }
  ncclResult_t r = ncclAllReduce(devInt64, devInt64, count, ncclInt64, ncclSum, c->comm, (hipStream_t)hipStream);
  if (r != ncclSuccess) {
    g_err = ncclGetErrorString(r);
    return -1;
  }
  return 0;
}

} // extern "C"
