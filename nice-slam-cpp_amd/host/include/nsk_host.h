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

// small device-buffer helper: float data of a (CPU or CUDA) tensor made available at a device pointer
struct DevBuf {
    float* p = nullptr; size_t n = 0;
    ~DevBuf();
    void ensure(size_t count);
    void upload(const torch::Tensor& t);        // any float tensor -> contiguous fp32 on the device
    torch::Tensor download(at::IntArrayRef shape) const;   // -> CPU tensor
};

}  // namespace nskh
