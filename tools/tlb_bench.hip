// tlb_bench.hip -- does the ALLOCATION flavour change the random-gather ceiling on a table far larger than the TLB reach?
//
// The PML walk on a 1 B-row (8 GB) table misses the per-CU TLB on 70 % of its lookups (profiles/r01_c4_*), and the pure
// gather rate drops from ~65 G/s (64 MB table) to ~49 G/s (8 GB).  This tool repeats the dependent 8-byte gather of
// tools/gather_bench.hip (and its 16 B / 32 B / neighbour forms) over tables obtained from
//   hipMalloc | hipExtMallocWithFlags(hipDeviceMallocContiguous) | hipMemCreate + hipMemMap (one physical handle)
// at several sizes, to see whether physically contiguous memory (larger page-table fragments) buys TLB reach.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tlb_bench tlb_bench.hip ; run: ./tlb_bench [log2 rows ...]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void fill_table(uint64_t *t, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) t[i] = mix(i + 0x9E3779B97F4A7C15ull);
}
__device__ __forceinline__ uint32_t next_idx(uint64_t v, uint32_t n_rows) {
    return __umulhi((uint32_t)(v ^ (v >> 32)), n_rows);
}

// VAR 0: 8 B row; 1: aligned 16 B pair; 2: aligned 32 B quad (2 x 16 B); 3: 8 B row + dependent neighbour row;
// 4: the pair as two 8-byte loads; 5: the quad as four 8-byte loads (round 3: how many address translations a wide
// divergent load costs against the same bytes fetched by 8-byte loads)
template <int VAR>
__global__ __launch_bounds__(256) void chase(const uint64_t *__restrict__ table, uint32_t n_rows, int steps,
                                             uint64_t n_lanes, uint64_t *__restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    uint32_t idx = next_idx(mix(VAR == 6 ? t >> 1 : (VAR == 7 ? t >> 2 : t)), n_rows);   // (6 / 7: the lanes of a pair / quad walk one chain)
    uint64_t acc = 0;
    for (int k = 0; k < steps; ++k) {
        uint64_t v;
        if (VAR == 0 || VAR == 3) {
            v = table[idx];
            if (VAR == 3) v ^= table[idx + 1 + (uint32_t)(v & 1)] >> 7;
        } else if (VAR == 1) {
            const uint4 q = *reinterpret_cast<const uint4 *>(table + (idx & ~1u));
            v = (idx & 1) ? ((uint64_t)q.w << 32 | q.z) : ((uint64_t)q.y << 32 | q.x);
            acc += q.x ^ q.z;
        } else if (VAR == 4) {
            // the aligned pair as TWO independent 8-byte loads of one line (explicit: the compiler would merge them)
            const uint64_t *p = table + (idx & ~1u);
            uint64_t a0, a1;
            asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %2, off offset:8\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a0), "=&v"(a1) : "v"(p) : "memory");
            v = (idx & 1) ? a1 : a0;
            acc += (uint32_t)a0 ^ (uint32_t)a1;
        } else if (VAR == 5) {
            // the aligned quad as FOUR independent 8-byte loads
            const uint64_t *p = table + (idx & ~3u);
            uint64_t a0, a1, a2, a3;
            asm volatile("global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %4, off offset:8\n\t"
                         "global_load_dwordx2 %2, %4, off offset:16\n\tglobal_load_dwordx2 %3, %4, off offset:24\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(p) : "memory");
            const uint64_t lo = (idx & 1) ? a1 : a0, hi = (idx & 1) ? a3 : a2;
            v = (idx & 2) ? hi : lo;
            acc += (uint32_t)a0 ^ (uint32_t)a2;
        } else if (VAR == 6) {
            // round 4: the aligned quad fetched by a PAIR of lanes in ONE instruction -- lanes 2i and 2i+1 walk the same chain
            // (same idx), each loads one 16-byte half, the halves are exchanged across the pair -- to see whether the two
            // lanes' translation requests for the same page are one request
            const uint4 *p = reinterpret_cast<const uint4 *>(table + (idx & ~3u)) + (threadIdx.x & 1u);
            const uint4 mine = p[0];
            uint4 other;
            other.x = __shfl_xor(mine.x, 1, 64); other.y = __shfl_xor(mine.y, 1, 64);
            other.z = __shfl_xor(mine.z, 1, 64); other.w = __shfl_xor(mine.w, 1, 64);
            const uint4 q0 = (threadIdx.x & 1u) ? other : mine, q1 = (threadIdx.x & 1u) ? mine : other;
            const uint4 q = (idx & 2) ? q1 : q0;
            v = (idx & 1) ? ((uint64_t)q.w << 32 | q.z) : ((uint64_t)q.y << 32 | q.x);
            acc += q0.x ^ q1.z;
        } else if (VAR == 7) {
            // the same by FOUR lanes with 8-byte loads (one instruction covers the quad)
            const uint64_t mine = table[(idx & ~3u) + (threadIdx.x & 3u)];
            const uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
            const int src = (int)((threadIdx.x & ~3u) | (idx & 3u));
            v = (uint64_t)__shfl(lo, src, 64) | ((uint64_t)__shfl(hi, src, 64) << 32);
            acc += __shfl_xor(lo, 2, 64);
        } else if (VAR == 8) {
            // every lane walks its OWN chain, the two lanes of a pair SHARE their loads: instruction 1 fetches the even lane's
            // quad (each lane one 16-byte half), instruction 2 the odd lane's; one exchange hands every lane its missing half
            const uint32_t odd = threadIdx.x & 1u;
            const uint32_t pidx = __shfl_xor(idx, 1, 64);
            const uint32_t ie = odd ? pidx : idx, io = odd ? idx : pidx;
            const uint4 r1 = *(reinterpret_cast<const uint4 *>(table + (ie & ~3u)) + odd);
            const uint4 r2 = *(reinterpret_cast<const uint4 *>(table + (io & ~3u)) + odd);
            const uint4 send = odd ? r1 : r2;
            uint4 recv;
            recv.x = __shfl_xor(send.x, 1, 64); recv.y = __shfl_xor(send.y, 1, 64);
            recv.z = __shfl_xor(send.z, 1, 64); recv.w = __shfl_xor(send.w, 1, 64);
            const uint4 q0 = odd ? recv : r1, q1 = odd ? r2 : recv;
            const uint4 q = (idx & 2) ? q1 : q0;
            v = (idx & 1) ? ((uint64_t)q.w << 32 | q.z) : ((uint64_t)q.y << 32 | q.x);
            acc += q0.x ^ q1.z;
        } else if (VAR >= 9 && VAR <= 12) {
            // round 4: the look-ahead rows' fetch -- 64 bytes per step out of one 128-byte line: a 32-byte window (at +0 or +32) and
            // its 32 bytes of entries (64 bytes further on).  9: one lane, four 16-byte loads; 10: a PAIR of lanes shares the loads
            // (four instructions, each covering one lane's window or entries: the kernel's PSH = 1); 11: FOUR lanes share them
            // (four instructions, each covering ONE lane's window AND entries: lanes 0 / 1 the window's halves, 2 / 3 the entries');
            // 12: the same with the 64 bytes contiguous (line = rows|entries|rows|entries).  Every lane walks its own chain; the next
            // index depends on what the lane itself loaded (no exchange: this measures the memory side only).
            const uint32_t lane = threadIdx.x;
            auto piece = [&](uint32_t i, uint32_t q) -> const uint4 * {   // 16-byte piece q (0..3) of the 64 bytes of index i
                const uint64_t *line = table + (i & ~15u);
                if (VAR == 12) return reinterpret_cast<const uint4 *>(line + ((i & 4u) ? 8u : 0u)) + q;
                return reinterpret_cast<const uint4 *>(line + ((i & 4u) ? 4u : 0u) + (q >> 1) * 8u) + (q & 1u);
            };
            uint4 r0, r1, r2, r3;
            if (VAR == 9) {
                r0 = *piece(idx, 0); r1 = *piece(idx, 1); r2 = *piece(idx, 2); r3 = *piece(idx, 3);
            } else if (VAR == 10) {
                const uint32_t odd = lane & 1u, pidx = __shfl_xor(idx, 1, 64);
                const uint32_t ie = odd ? pidx : idx, io = odd ? idx : pidx;
                r0 = *piece(ie, odd); r1 = *piece(ie, 2u + odd); r2 = *piece(io, odd); r3 = *piece(io, 2u + odd);
            } else {
                const uint32_t q = lane & 3u, b = lane & ~3u;
                const uint32_t i0 = __shfl(idx, (int)b, 64), i1 = __shfl(idx, (int)(b | 1u), 64), i2 = __shfl(idx, (int)(b | 2u), 64),
                               i3 = __shfl(idx, (int)(b | 3u), 64);
                r0 = *piece(i0, q); r1 = *piece(i1, q); r2 = *piece(i2, q); r3 = *piece(i3, q);
            }
            v = ((uint64_t)(r0.x ^ r1.y ^ r2.z ^ r3.w) << 32) | (uint64_t)(r0.y ^ r1.z ^ r2.w ^ r3.x);
            acc += r0.z ^ r3.y;
        } else {
            const uint4 *p = reinterpret_cast<const uint4 *>(table + (idx & ~3u));
            const uint4 q0 = p[0], q1 = p[1];
            const uint4 q = (idx & 2) ? q1 : q0;
            v = (idx & 1) ? ((uint64_t)q.w << 32 | q.z) : ((uint64_t)q.y << 32 | q.x);
            acc += q0.x ^ q1.z;
        }
        acc ^= v;
        idx = next_idx(v + k, n_rows);
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

template <int VAR>
double run(const uint64_t *table, uint32_t n_rows, int steps, uint64_t lanes, uint64_t *sink, int reps) {
    dim3 block(256), grid((unsigned)((lanes + 255) / 256));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(chase<VAR>, grid, block, 0, 0, table, n_rows, steps, lanes, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(chase<VAR>, grid, block, 0, 0, table, n_rows, steps, lanes, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return (double)ms / reps;
}

int main(int argc, char **argv) {
    int sizes[8], ns = 0;
    for (int i = 1; i < argc && ns < 8; i++) sizes[ns++] = atoi(argv[i]);
    if (ns == 0) { sizes[0] = 24; sizes[1] = 27; sizes[2] = 30; ns = 3; }
    const int steps = 150;
    uint64_t *sink;
    CHECK(hipMalloc(&sink, 64));
    const char *alloc_names[3] = {"hipMalloc", "hipExtMallocWithFlags(hipDeviceMallocContiguous)", "hipMemCreate+hipMemMap (one handle)"};
    const char *names[13] = {"8B row", "16B pair", "32B quad", "8B row + dependent neighbour", "pair as 2 x 8B loads", "quad as 4 x 8B loads",
                             "32B quad by a lane PAIR (2 x 16B, one instr; steps = lanes / 2)", "32B quad by FOUR lanes (4 x 8B, one instr; steps = lanes / 4)",
                             "32B quad, own chain per lane, loads SHARED by the pair (2 instr)",
                             "64B of a line (32B window + 32B entries at +64): one lane, 4 x 16B", "... loads shared by a lane PAIR (4 instr)",
                             "... loads shared by FOUR lanes (4 instr, each one lane's 64B)", "... by FOUR lanes, the 64B contiguous"};
    const char *only_var = getenv("TLB_VAR");               // e.g. TLB_VAR=4: that variant only (PMC passes)
    for (int si = 0; si < ns; si++) {
        const uint64_t n = 1ull << sizes[si];
        const char *only = getenv("TLB_ALLOC");             // e.g. TLB_ALLOC=0: hipMalloc only (profiling runs)
        for (int which = 0; which < 3; which++) {
            if (only && atoi(only) != which) continue;
            uint64_t *table = nullptr;
            hipMemGenericAllocationHandle_t handle{};
            size_t vm_bytes = 0;
            hipError_t ea = hipSuccess;
            if (which == 0) ea = hipMalloc(&table, n * 8 + 64);
            else if (which == 1) ea = hipExtMallocWithFlags((void **)&table, n * 8 + 64, hipDeviceMallocContiguous);
            else {
                hipMemAllocationProp prop{};
                prop.type = hipMemAllocationTypePinned;
                prop.location.type = hipMemLocationTypeDevice;
                prop.location.id = 0;
                size_t gran = 0;
                ea = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
                if (ea == hipSuccess) {
                    printf("  (VMM recommended granularity: %zu bytes)\n", gran);
                    vm_bytes = ((n * 8 + 64 + gran - 1) / gran) * gran;
                    ea = hipMemCreate(&handle, vm_bytes, &prop, 0);
                }
                if (ea == hipSuccess) ea = hipMemAddressReserve((void **)&table, vm_bytes, 1ull << 30, nullptr, 0);
                if (ea == hipSuccess) ea = hipMemMap(table, vm_bytes, 0, handle, 0);
                if (ea == hipSuccess) {
                    hipMemAccessDesc acc{};
                    acc.location.type = hipMemLocationTypeDevice;
                    acc.location.id = 0;
                    acc.flags = hipMemAccessFlagsProtReadWrite;
                    ea = hipMemSetAccess(table, vm_bytes, &acc, 1);
                }
            }
            printf("table 2^%d rows = %.1f MB, alloc: %s, ptr %p\n", sizes[si], n * 8 / 1e6, alloc_names[which], (void *)table);
            if (ea != hipSuccess) { printf("  allocation failed: %s\n", hipGetErrorString(ea)); (void)hipGetLastError(); continue; }
            hipLaunchKernelGGL(fill_table, dim3(4096), dim3(256), 0, 0, table, n);
            CHECK(hipDeviceSynchronize());
            const uint32_t n_rows = (uint32_t)(n - 16);
            for (uint64_t lanes : {(uint64_t)1 << 19, (uint64_t)1 << 20}) {
                double ms[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                auto want = [&](int v) { return !only_var || atoi(only_var) == v || (only_var[0] == '>' && v >= atoi(only_var + 1)); };   // TLB_VAR=4, or ">9": from 9 on
                if (want(0)) ms[0] = run<0>(table, n_rows, steps, lanes, sink, 4);
                if (want(1)) ms[1] = run<1>(table, n_rows, steps, lanes, sink, 4);
                if (want(2)) ms[2] = run<2>(table, n_rows, steps, lanes, sink, 4);
                if (want(3)) ms[3] = run<3>(table, n_rows, steps, lanes, sink, 4);
                if (want(4)) ms[4] = run<4>(table, n_rows, steps, lanes, sink, 4);
                if (want(5)) ms[5] = run<5>(table, n_rows, steps, lanes, sink, 4);
                if (want(6)) ms[6] = run<6>(table, n_rows, steps, 2 * lanes, sink, 4);      // the same number of CHAINS: twice / four times the lanes
                if (want(7)) ms[7] = run<7>(table, n_rows, steps, 4 * lanes, sink, 4);
                if (want(8)) ms[8] = run<8>(table, n_rows, steps, lanes, sink, 4);
                if (want(9)) ms[9] = run<9>(table, n_rows, steps, lanes, sink, 4);
                if (want(10)) ms[10] = run<10>(table, n_rows, steps, lanes, sink, 4);
                if (want(11)) ms[11] = run<11>(table, n_rows, steps, lanes, sink, 4);
                if (want(12)) ms[12] = run<12>(table, n_rows, steps, lanes, sink, 4);
                for (int v = 0; v < 13; v++)
                    if (want(v))
                        printf("  chains=%8llu  %-64s %8.3f ms  %7.2f Gsteps/s\n", (unsigned long long)lanes, names[v], ms[v],
                               lanes * (double)steps / ms[v] / 1e6);
            }
            if (which == 2) {
                (void)hipMemUnmap(table, vm_bytes);
                (void)hipMemRelease(handle);
                (void)hipMemAddressFree(table, vm_bytes);
            } else CHECK(hipFree(table));
        }
    }
    return 0;
}
