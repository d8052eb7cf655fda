// Internal interface between engine.hip and the native RCCL communicator (rccl_comm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

struct admp_rccl;

namespace admp {
int rccl_all_reduce(admp_rccl* c, hipStream_t st, void* buf, int64_t count, int dtype, int op, int tag);
int rccl_all_to_all_v(admp_rccl* c, hipStream_t st, const void* send, const int64_t* sc, void* recv, const int64_t* rc, int dtype,
                      int tag);
int rccl_shift(admp_rccl* c, hipStream_t st, const void* send, void* recv, int64_t count, int dtype, int to_next, int tag);
int rccl_rank(const admp_rccl* c);
int rccl_nranks(const admp_rccl* c);
int rccl_device(const admp_rccl* c);
std::string rccl_error();
}  // namespace admp
