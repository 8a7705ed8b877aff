// movi_expand_host.cpp -- reset masks -> u16 PML vectors on the host.
// PML[k] = reset(k) ? 0 : PML[k - 1] + 1, u16-clamped (MoveQuery::add_ml, include/move_query.hpp:26-38, of the match_len
// process_char keeps, src/read_processor.cpp:193-215): the run length since the last reset.  16 bases per step with AVX2:
// lane b holds (position + 1 of the latest reset at or before b, by a prefix maximum) and its PML is the distance to it -- or
// match_len before the group + b + 1 (saturating add) where the group has had no reset yet.
#include "movi_expand_host.hpp"

#include <immintrin.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace movi {

namespace {

inline void expand_read_scalar(const uint32_t *M, uint32_t len, uint16_t *O) {
    uint32_t run = 0;
    for (uint32_t k = 0; k < len; ++k) {
        run = ((M[k >> 5] >> (k & 31u)) & 1u) ? 0u : run + 1u;
        O[k] = (uint16_t)(run > 65535u ? 65535u : run);
    }
}

// 16 bases (bits of h, base 0 in bit 0) from match_len `run`: the 16 PMLs, and `run` afterwards
__attribute__((target("avx2"))) inline __m256i expand16_avx2(uint32_t h, uint32_t &run) {
    const __m256i bit = _mm256_setr_epi16(1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, (short)0x8000);
    const __m256i iota1 = _mm256_setr_epi16(1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16);
    const __m256i zero = _mm256_setzero_si256();
    const __m256i fresh = _mm256_adds_epu16(iota1, _mm256_set1_epi16((short)(run > 65535u ? 65535u : run)));   // match_len + 1 .. + 16, clamped
    if (h == 0) {                                          // no reset among the 16
        run += 16;
        return fresh;
    }
    const __m256i hb = _mm256_set1_epi16((short)h);
    const __m256i isr = _mm256_cmpeq_epi16(_mm256_and_si256(hb, bit), bit);
    __m256i x = _mm256_and_si256(isr, iota1);              // b + 1 where base b is a reset
    x = _mm256_max_epu16(x, _mm256_slli_si256(x, 2));
    x = _mm256_max_epu16(x, _mm256_slli_si256(x, 4));
    x = _mm256_max_epu16(x, _mm256_slli_si256(x, 8));                   // prefix maxima inside each half
    const __m128i lo_top = _mm_set1_epi16((short)_mm256_extract_epi16(x, 7));
    x = _mm256_max_epu16(x, _mm256_inserti128_si256(zero, lo_top, 1));  // ... and across the halves
    const __m256i since = _mm256_sub_epi16(iota1, x);                   // bases since the latest reset (0 at the reset)
    run = (uint32_t)__builtin_clz(h << 16);                             // bases after the group's last reset
    return _mm256_blendv_epi8(since, fresh, _mm256_cmpeq_epi16(x, zero));
}

__attribute__((target("avx2"))) inline void expand_read_avx2(const uint32_t *M, uint32_t len, uint16_t *O) {
    uint32_t run = 0;
    uint32_t k = 0;
    for (; k + 16 <= len; k += 16)
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(O + k), expand16_avx2((M[k >> 5] >> (k & 16u)) & 0xFFFFu, run));
    for (; k < len; ++k) {
        run = ((M[k >> 5] >> (k & 31u)) & 1u) ? 0u : run + 1u;
        O[k] = (uint16_t)(run > 65535u ? 65535u : run);
    }
}

// The same for a RANGE of reads whose vectors are one contiguous stretch of the output (out[offs[i0]] .. out[offs[i1]]): the PMLs are
// produced into a cache-resident block and leave through aligned NON-TEMPORAL 32-byte stores -- a plain store first reads the line it is
// about to overwrite (read-for-ownership), which doubles the memory traffic of a pass that writes 2 bytes per base and is bound by it
// (16 threads on the GPU box: 21.8 Gbases/s with plain stores).
struct NtStream {
    uint16_t *dst;                                         // where buf[0] goes
    size_t n = 0;
    alignas(64) uint16_t buf[8192 + 64];
    __attribute__((target("avx2"))) void flush(bool last) {
        size_t i = 0;
        while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31u)) { dst[i] = buf[i]; ++i; }
        for (; i + 16 <= n; i += 16)
            _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(buf + i)));
        if (last) {
            for (; i < n; ++i) dst[i] = buf[i];
            dst += n;
            n = 0;
        } else {                                           // what does not fill a 32-byte store waits for the next flush
            const size_t rem = n - i;
            memmove(buf, buf + i, rem * 2);
            dst += i;
            n = rem;
        }
    }
};

bool have_avx2() {
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
}

}  // namespace

void expand_reads_scalar(const ExpandJob &j) {
    for (uint64_t i = j.i0; i < j.i1; ++i) {
        const uint64_t beg = j.offs[i];
        expand_read_scalar(j.words + ((beg - j.o0 + j.phase) >> 5) + (i - j.ibase), (uint32_t)(j.offs[i + 1] - beg), j.out + beg);
    }
}

__attribute__((target("avx2"))) static void expand_reads_avx2(const ExpandJob &j) {
    static const bool nt = [] { const char *e = getenv("MOVI_EXPAND_NT"); return !e || e[0] != '0'; }();   // MOVI_EXPAND_NT=0: plain stores (A/B)
    if (!nt || j.offs[j.i1] - j.offs[j.i0] < 4096) {       // a small range: straight to the output
        for (uint64_t i = j.i0; i < j.i1; ++i) {
            const uint64_t beg = j.offs[i];
            expand_read_avx2(j.words + ((beg - j.o0 + j.phase) >> 5) + (i - j.ibase), (uint32_t)(j.offs[i + 1] - beg), j.out + beg);
        }
        return;
    }
    NtStream st;
    st.dst = j.out + j.offs[j.i0];
    for (uint64_t i = j.i0; i < j.i1; ++i) {
        const uint64_t beg = j.offs[i];
        const uint32_t len = (uint32_t)(j.offs[i + 1] - beg);
        const uint32_t *M = j.words + ((beg - j.o0 + j.phase) >> 5) + (i - j.ibase);
        uint32_t run = 0, k = 0;
        while (k < len) {                                  // the read, in pieces that fit the block
            if (st.n + 16 > 8192) st.flush(false);
            const uint32_t room = (uint32_t)((8192 - st.n) & ~(size_t)15);
            const uint32_t piece_end = len - k <= room ? len : k + room;
            uint16_t *B = st.buf + st.n;
            const uint32_t k0 = k;
            for (; k + 16 <= piece_end; k += 16)
                _mm256_storeu_si256(reinterpret_cast<__m256i *>(B + (k - k0)), expand16_avx2((M[k >> 5] >> (k & 16u)) & 0xFFFFu, run));
            if (piece_end == len)
                for (; k < len; ++k) {
                    run = ((M[k >> 5] >> (k & 31u)) & 1u) ? 0u : run + 1u;
                    B[k - k0] = (uint16_t)(run > 65535u ? 65535u : run);
                }
            st.n += k - k0;
        }
    }
    st.flush(true);
    _mm_sfence();
}

void expand_reads(const ExpandJob &j) {
    if (have_avx2()) expand_reads_avx2(j);
    else expand_reads_scalar(j);
}

// Worker threads of the expansion by default: three quarters of the CPUs the process may really use -- its affinity mask capped by the
// cgroup's CPU quota (the GPU boxes show 256 logical CPUs and grant 16) --, at most 32.  Not all of them: the calling thread drives the
// GPU pipeline beside the workers, and a cgroup that runs over its quota is throttled as a whole (16 workers on a quota of 16:
// movi_pml_host 17.8 - 25 Gbases/s from run to run, 12 workers: 27; profiles/r06_mask_path.txt).
int host_threads_default() {
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {      // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32] = {0};
        long long per = 0;
        if (fscanf(f, "%31s %lld", q, &per) == 2 && q[0] != 'm' && per > 0) {
            const long long cpus = atoll(q) / per;
            if (cpus >= 1 && cpus < n) n = (int)cpus;
        }
        fclose(f);
    }
    n = n > 32 ? 32 : n;
    return n >= 4 ? n * 3 / 4 : n;
}

HostPool &HostPool::get() {
    static HostPool *pool = new HostPool();               // (never destroyed: worker threads may outlive static destructors)
    return *pool;
}

HostPool::~HostPool() {
    {
        std::lock_guard<std::mutex> g(m_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
}

void HostPool::ensure_threads(int n) {
    std::lock_guard<std::mutex> g(m_);
    while ((int)threads_.size() < n) threads_.emplace_back([this] { worker(); });
}

void HostPool::worker() {
    for (;;) {
        std::pair<Group *, std::function<void()>> job;
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;
            job = std::move(q_.front());
            q_.pop_front();
        }
        job.second();
        if (job.first) {
            std::lock_guard<std::mutex> g(job.first->m);
            if (--job.first->pending == 0) job.first->cv.notify_all();
        }
    }
}

void HostPool::submit(Group *g, std::function<void()> fn) {
    if (g) {
        std::lock_guard<std::mutex> lk(g->m);
        g->pending += 1;
    }
    {
        std::lock_guard<std::mutex> lk(m_);
        q_.emplace_back(g, std::move(fn));
    }
    cv_.notify_one();
}

void HostPool::wait(Group *g) {
    std::unique_lock<std::mutex> lk(g->m);
    g->cv.wait(lk, [g] { return g->pending == 0; });
}

void expand_parallel(const ExpandJob &whole, int threads, HostPool::Group *group) {
    if (whole.i1 <= whole.i0) return;
    if (threads <= 0) threads = host_threads_default();
    const uint64_t bases = whole.offs[whole.i1] - whole.offs[whole.i0];
    constexpr uint64_t kTaskBases = 1ull << 17;
    if (threads == 1 || bases <= kTaskBases) {
        if (!group) { expand_reads(whole); return; }
    }
    HostPool &pool = HostPool::get();
    pool.ensure_threads(threads);
    HostPool::Group local;
    HostPool::Group *g = group ? group : &local;
    for (uint64_t i = whole.i0; i < whole.i1;) {
        // the longest run of reads from i with at most kTaskBases bases (at least one read)
        const uint64_t *lo = whole.offs + i + 1, *end = whole.offs + whole.i1 + 1;
        uint64_t last = (uint64_t)(std::upper_bound(lo, end, whole.offs[i] + kTaskBases) - whole.offs) - 1;
        if (last <= i) last = i + 1;
        ExpandJob part = whole;
        part.i0 = i;
        part.i1 = last;
        pool.submit(g, [part] { expand_reads(part); });
        i = last;
    }
    if (!group) pool.wait(&local);
}

}  // namespace movi
