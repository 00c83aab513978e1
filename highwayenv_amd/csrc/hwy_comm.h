// hwy_comm.h -- RCCL gather behind hwy_comm_* / hwy_gather (hwy_comm.hip); librccl is dlopen'ed on first use.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace hwy {
struct Comm;
int comm_unique_id(uint8_t *id, std::string &err);
int comm_init(Comm **out, const uint8_t *id, int rank, int world, std::string &err);
int comm_gather(Comm *c, const void *d_send, void *d_recv, size_t bytes, int root, hipStream_t stream, std::string &err);
void comm_destroy(Comm *c);
int comm_rank(const Comm *c);
int comm_world(const Comm *c);
}  // namespace hwy
