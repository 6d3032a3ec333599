// rccl_leg.h - the library's own collective leg: RCCL (librccl.so.1, bound with dlopen at first use) over xGMI.
//
// The path has ONE exchange step per LM iteration: a sum of the reduce payload
// [block-sparse S | rhs | diag B | g_c | cost] and a sum of a handful of step scalars.  Both are in-place
// ncclAllReduce calls on the handle's own stream, so they are ordered against the library's kernels by construction -
// no host callback, no second stream.  The callback of soslam_ba_set_allreduce stays as the alternative for hosts that
// bring their own collective (MPI, torch.distributed/gloo rehearsals).
//
// librccl is not a link-time dependency: a process that already carries it (torch ships a copy with the same SONAME)
// keeps that single copy, a plain C++ host gets ROCm's.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace soslam {

constexpr int kRcclIdBytes = 128;   // NCCL_UNIQUE_ID_BYTES

struct RcclComm;   // one communicator (one per handle and device)

// 0 on success; on failure set_last_error() holds the reason (library not found, RCCL error string)
int rccl_get_unique_id(void* id128);
int rccl_comm_create(const void* id128, int rank, int world, int device, RcclComm** out);
void rccl_comm_destroy(RcclComm* c);
// in-place all-reduce of count f64 on `stream`; op: SOSLAM_REDUCE_SUM / SOSLAM_REDUCE_MAX
int rccl_allreduce_f64(RcclComm* c, double* buf, size_t count, int op, hipStream_t stream);

}  // namespace soslam
