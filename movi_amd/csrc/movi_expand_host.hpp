// movi_expand_host.hpp -- reset masks -> u16 PML vectors on the host (movi_pml_expand_host; the harvest stage of
// movi_pml_host when only masks cross PCIe) and the worker pool that runs it.  Pure C++ (no HIP): compiled by g++.
#pragma once
#include <stdint.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace movi {

// Reads [i0, i1) of a batch whose masks are laid out as include/movi_hip.h says: the words of read i start at
// words[((offs[i] - o0 + phase) >> 5) + (i - ibase)], its PMLs go to out[offs[i] + k].
struct ExpandJob {
    const uint32_t *words = nullptr;
    const uint64_t *offs = nullptr;
    uint64_t o0 = 0;
    uint32_t phase = 0;
    uint64_t ibase = 0;
    uint64_t i0 = 0, i1 = 0;
    uint16_t *out = nullptr;
};
void expand_reads(const ExpandJob &job);          // on the calling thread (AVX2 where the CPU has it)
void expand_reads_scalar(const ExpandJob &job);   // the plain loop (what the tests hold the vector code to)

int host_threads_default();                       // three quarters of the CPUs the process may use (affinity, cgroup quota), at most 24

// A small persistent pool: tasks are submitted in groups, a group can be waited for.  One pool per process, grown on demand.
class HostPool {
public:
    struct Group {
        std::mutex m;
        std::condition_variable cv;
        uint64_t pending = 0;
    };
    static HostPool &get();
    void ensure_threads(int n);
    void submit(Group *g, std::function<void()> fn);
    void wait(Group *g);
    ~HostPool();

private:
    HostPool() = default;
    void worker();
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::pair<Group *, std::function<void()>>> q_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
};

// Expands reads [0, n_reads) with `threads` workers of the pool, cut into tasks of about 2^17 bases; with `group` the tasks are
// only submitted (the caller waits), without it the call returns when they are done.
void expand_parallel(const ExpandJob &whole, int threads, HostPool::Group *group);

}  // namespace movi
