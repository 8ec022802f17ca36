// rp_internal.h — shared host-side helpers of librp_mi355x.so (not part of the ABI).
#ifndef RP_INTERNAL_H
#define RP_INTERNAL_H

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/rp_mi355x.h"
#include "../../include/rp_mi355x_diag.h"

#define RP_MAX_ACTIONS 16u   // widest infoset row the device kernels are compiled for (NLHE needs 9..14)
#define RP_MAX_PLAYERS 8u
#define RP_STR2(x) #x
#define RP_STR(x) RP_STR2(x)

struct ihipStream_t;

namespace rp {

// records the message for rp_last_error() and returns the code
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// Checkpoint's display line / Progress::format (metrics/checkpoint.rs:39-50, progress.rs:8-18)
void format_progress(char* buf, size_t cap, uint64_t epoch, uint64_t nodes, uint64_t infos, double rate);

}  // namespace rp

// sparse.hip / deuce.hip: what nlmc.hip needs of the profile and of the encoder tables
struct rp_profile;
struct rp_lookup;
namespace rp {
float* profile_table(rp_profile* h);
ihipStream_t* profile_stream(rp_profile* h);
uint64_t profile_epoch(const rp_profile* h);
unsigned char* profile_entries(rp_profile* h);
int lookup_view(const rp_lookup* t, const uint64_t** keys, const uint8_t** abs_, uint64_t* n, int* street);
}  // namespace rp

// comm.cpp: the collectives behind rp_comm, enqueued on the caller's HIP stream
struct rp_comm;
namespace rp {
int comm_world(const rp_comm* c);
int comm_rank(const rp_comm* c);
int comm_all_gather(rp_comm* c, const void* send, void* recv, size_t bytes, ihipStream_t* stream);
int comm_all_reduce_sum(rp_comm* c, void* buf, size_t count, int kind /* 0 = i32, 1 = i64 */, ihipStream_t* stream);
}  // namespace rp

#endif
