// Umbrella header of the GPU API (reference: /root/reference/wf/windflow_gpu.hpp:34-44).
// Must be compiled with hipcc --offload-arch=gfx950.
#pragma once
#include "windflow.hpp"
#include "gpu/hip_utils.hpp"
#include "gpu/kernels_basic.hpp"
#include "gpu/batch_ops.hpp"
#include "gpu/map_gpu.hpp"
#include "gpu/filter_gpu.hpp"
#include "gpu/primitives.hpp"
#include "gpu/flatfat_gpu.hpp"
#include "gpu/key_ffat_gpu.hpp"
#include "gpu/win_seq_gpu.hpp"
#include "gpu/keyby_gpu.hpp"
#include "gpu/keyed_gpu.hpp"
#include "builders_gpu.hpp"
#include "dist/transport.hpp"
#include "dist/exchange_gpu.hpp"
