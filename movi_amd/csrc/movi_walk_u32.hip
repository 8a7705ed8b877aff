// movi_walk_u32.hip -- the walk kernel's instantiations for 32-bit row indexes, whole reads
// (one translation unit per (index width, segment class): they compile in parallel; movi_walk.hpp has the kernel).
#include "movi_walk.hpp"

namespace movi {

hipError_t launch_walk_u32(const WalkLaunch &L, LaunchInfo *info) {
    return walk_dispatch<uint32_t, 0>(L, info);
}

// Loads this translation unit's code object now (hipFuncGetAttributes on one of its kernels does), so that the first walk on a
// prepared handle does not (movi_index_prepare).
hipError_t preload_walk_u32() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&pml_kernel_flatp<6, uint32_t, 0, 0, 0, 1, 1, 0, 0>));
}

}  // namespace movi
