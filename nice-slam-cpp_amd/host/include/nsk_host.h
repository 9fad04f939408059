// nsk_host.h -- glue shared by the host classes: one process-wide nsk context per device, and the marshalling of
// libtorch tensors (host-side containers only; no libtorch GPU op is ever called) into the C-ABI of include/nsk.h.
#pragma once
#include <torch/torch.h>
#include <stdexcept>
#include <string>
#include "nsk.h"

namespace nskh {

typedef c10::Dict<std::string, torch::Tensor> GridDict;

nsk_ctx* ctx();                                 // created on first use on device NSK_DEVICE (default 0)
void check(int rc);                             // throws std::runtime_error(nsk_last_error()) like libtorch's c10::Error would
int stage_id(const std::string& stage);         // "coarse" | "middle" | "fine" | "color"

// Make the context's copy of the grids / decoders current.  Uploads happen only when a tensor changed (data pointer
// or version counter), so calling this on every render costs nothing in the steady state.
void sync_grids(const GridDict& c);
void fetch_grids(GridDict& c);                  // write the context's (optimised) grids back into the Dict's tensors

// N > 1: one process per GPU (NSK_DEVICE = the local rank), rays shard over the ranks, ONE all-reduce (sum, fp32) per mapping iteration
// (SURVEY.md section 8e).  `comm` is an ncclComm_t (RCCL; nskh::rccl_comm_from_file bootstraps one without MPI); a test may install a
// function instead that sums a device buffer over the ranks in place (tests/test_gpu_host_cpp.py: two processes on one GPU through
// shared memory -- RCCL refuses two ranks on one device).
struct Dist {
    int rank = 0, world = 1;
    void* comm = nullptr;
    void (*hook)(float* d_buf, size_t n_floats, void* user) = nullptr; void* user = nullptr;
    bool on() const { return world > 1; }
    void allreduce_grads() const;                         // nsk_grad_pack -> sum over the ranks -> nsk_grad_unpack, on the context's stream
    void allreduce(float* d_buf, size_t n) const;         // a plain device vector (n floats)
    void broadcast0(float* d_buf, size_t n) const;        // rank 0's values to every rank (the Tracker's pose: 8 floats), as a sum with zeros
    void shard(int n, int* lo, int* hi) const;            // this rank's contiguous range of n rays; ranges differ by at most one ray
};
// ncclGetUniqueId on rank 0, its 128 bytes through a file every rank can read, ncclCommInitRank on the current device (librccl is dlopen'ed)
void* rccl_comm_from_file(int rank, int world, const std::string& id_file);

// small device-buffer helper: float data of a (CPU or CUDA) tensor made available at a device pointer
struct DevBuf {
    float* p = nullptr; size_t n = 0;
    ~DevBuf();
    void ensure(size_t count);
    void upload(const torch::Tensor& t);        // any float tensor -> contiguous fp32 on the device
    torch::Tensor download(at::IntArrayRef shape) const;   // -> CPU tensor
};

}  // namespace nskh
