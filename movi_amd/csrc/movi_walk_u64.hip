// movi_walk_u64.hip -- the walk kernel's instantiations for 64-bit row indexes, whole reads
// (one translation unit per (index width, segment class): they compile in parallel; movi_walk.hpp has the kernel).
#include "movi_walk.hpp"

namespace movi {

hipError_t launch_walk_u64(const WalkLaunch &L, LaunchInfo *info) {
    return walk_dispatch<uint64_t, 0>(L, info);
}

}  // namespace movi
