// rp_internal.h — shared host-side helpers of librp_mi355x.so (not part of the ABI).
#ifndef RP_INTERNAL_H
#define RP_INTERNAL_H

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/rp_mi355x.h"

#define RP_MAX_ACTIONS 16u   // widest infoset row the device kernels are compiled for (NLHE needs 9..14)
#define RP_MAX_PLAYERS 8u
#define RP_STR2(x) #x
#define RP_STR(x) RP_STR2(x)

namespace rp {

// records the message for rp_last_error() and returns the code
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace rp

// comm.cpp: the collectives behind rp_comm, enqueued on the caller's HIP stream
struct rp_comm;
struct ihipStream_t;
namespace rp {
int comm_world(const rp_comm* c);
int comm_rank(const rp_comm* c);
int comm_all_gather(rp_comm* c, const void* send, void* recv, size_t bytes, ihipStream_t* stream);
int comm_all_reduce_sum(rp_comm* c, void* buf, size_t count, int kind /* 0 = i32, 1 = i64 */, ihipStream_t* stream);
}  // namespace rp

#endif
