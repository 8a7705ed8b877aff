// gather_bench.hip -- microbenchmark: dependent random row gathers on MI355X.
//
// Measures the ceiling the PML walk lives under: each lane chases pointers through a
// table of 8-byte rows (next index = f(loaded row)), like LF_move does.  Variants
// differ in how much of the 64-byte sector around the row is fetched per step and
// how, and in how the per-step output is stored.  Used to choose the kernel design
// and to state the achievable random-gather roofline in DESIGN.md.
//
// build: hipcc -O3 --offload-arch=gfx950 -o gather_bench gather_bench.hip
// run:   ./gather_bench [rows_log2_small=23] [rows_log2_big=30]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void fill_table(uint64_t *t, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) t[i] = mix(i + 0x9E3779B97F4A7C15ull);
}

__device__ __forceinline__ uint32_t next_idx(uint64_t v, uint32_t n_rows) {
    return __umulhi((uint32_t)(v ^ (v >> 32)), n_rows);
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// VAR: 0 = 8 B row; 1 = 16 B aligned pair; 2 = 32 B aligned; 3 = 64 B sector (4 x dwordx4 to VGPRs);
//      4 = 64 B sector via 4 x global_load_lds_dwordx4 + ds_read_b64; 5 = 8 B row + dependent neighbour row (j+1);
//      6 = 8 B row + 2-byte store per step; 7 = 8 B row + 16-byte store every 8 steps;
//      8 = 8 B row + 1-byte read-stream load per step (like the base fetch)
template <int VAR>
__global__ __launch_bounds__(256) void chase(const uint64_t *__restrict__ table, uint32_t n_rows, int steps,
                                             uint64_t n_lanes, uint64_t *__restrict__ sink,
                                             uint16_t *__restrict__ out, const uint8_t *__restrict__ stream) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[4][4 * 1024];   // [wave][plane*1024 + lane*16]
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t idx = next_idx(mix(t), n_rows);
    uint64_t acc = 0;
    uint4 pk = make_uint4(0, 0, 0, 0);
    for (int k = 0; k < steps; ++k) {
        uint64_t v;
        if (VAR == 9) {
            v = __builtin_nontemporal_load(table + idx);
        } else if (VAR == 10) {
            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(table + idx) : "memory");
        } else if (VAR == 11) {
            asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(table + idx) : "memory");
        } else if (VAR == 12) {
            asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(table + idx) : "memory");
        } else if (VAR == 13) {
            asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(table + idx) : "memory");
        } else if (VAR == 0 || VAR >= 5) {
            v = table[idx];
            if (VAR == 5) {
                const uint64_t v2 = table[idx + 1 + (uint32_t)(v & 1)];   // issued only after v arrives
                v ^= v2 >> 7;
            }
        } else if (VAR == 1) {
            const uint4 q = *reinterpret_cast<const uint4 *>(table + (idx & ~1u));
            v = (idx & 1) ? ((uint64_t)q.w << 32 | q.z) : ((uint64_t)q.y << 32 | q.x);
            acc += q.x ^ q.z;
        } else if (VAR == 2) {
            const uint4 *p = reinterpret_cast<const uint4 *>(table + (idx & ~3u));
            const uint4 q0 = p[0], q1 = p[1];
            const uint4 q = (idx & 2) ? q1 : q0;
            v = (idx & 1) ? ((uint64_t)q.w << 32 | q.z) : ((uint64_t)q.y << 32 | q.x);
            acc += q0.x ^ q1.z;
        } else if (VAR == 3) {
            const uint4 *p = reinterpret_cast<const uint4 *>(table + (idx & ~7u));
            const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            const uint4 qa = (idx & 2) ? q1 : q0, qb = (idx & 2) ? q3 : q2;
            const uint4 q = (idx & 4) ? qb : qa;
            v = (idx & 1) ? ((uint64_t)q.w << 32 | q.z) : ((uint64_t)q.y << 32 | q.x);
            acc += q0.x ^ q1.z ^ q2.y ^ q3.w;
        } else {   // VAR == 4
            const uint8_t *src = reinterpret_cast<const uint8_t *>(table + (idx & ~7u));
            uint8_t *dst = &stage[wave][0];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((glb_void *)(src + q * 16), (lds_void *)(dst + q * 1024), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) lgkmcnt(0): the DMA has landed
            const uint32_t m = idx & 7;
            v = *reinterpret_cast<const uint64_t *>(dst + (m >> 1) * 1024 + lane * 16 + (m & 1) * 8);
        }
        if (VAR == 6) out[t * steps + k] = (uint16_t)v;
        if (VAR == 7) {
            pk.x = pk.y; pk.y = pk.z; pk.z = pk.w; pk.w = (uint32_t)v;
            if ((k & 7) == 7) *reinterpret_cast<uint4 *>(out + t * steps + (k - 7)) = pk;
        }
        if (VAR == 8) v ^= stream[t * steps + k];
        acc ^= v;
        idx = next_idx(v + k, n_rows);
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

template <int VAR>
double run(const uint64_t *table, uint32_t n_rows, int steps, uint64_t lanes, uint64_t *sink, uint16_t *out,
           const uint8_t *stream, int reps) {
    dim3 block(256), grid((unsigned)((lanes + 255) / 256));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(chase<VAR>, grid, block, 0, 0, table, n_rows, steps, lanes, sink, out, stream);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL(chase<VAR>, grid, block, 0, 0, table, n_rows, steps, lanes, sink, out, stream);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return (double)ms / reps;
}

int main(int argc, char **argv) {
    int small_log2 = argc > 1 ? atoi(argv[1]) : 23;   // 8 M rows = 64 MB (Infinity-Cache resident)
    int big_log2 = argc > 2 ? atoi(argv[2]) : 30;     // 1 G rows = 8 GB (HBM)
    const int steps = 150;
    const uint64_t max_lanes = 1u << 20;
    uint64_t *sink;
    uint16_t *out;
    uint8_t *stream;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&out, max_lanes * steps * 2 + 64));
    CHECK(hipMalloc(&stream, max_lanes * steps + 64));
    CHECK(hipMemset(stream, 1, max_lanes * steps));
    const char *names[14] = {"8B row", "16B pair", "32B quad", "64B sector->VGPR", "64B sector->LDS (glds)",
                            "8B row + dependent neighbour", "8B row + 2B store/step", "8B row + 16B store/8 steps",
                            "8B row + 1B stream load/step", "8B row nt", "8B row sc1", "8B row sc0 sc1",
                            "8B row sc0 sc1 nt", "8B row sc0"};
    const char *alloc_names[4] = {"hipMalloc", "hipMalloc", "hipExtMallocWithFlags(Uncached)", "hipExtMallocWithFlags(FineGrained)"};
    for (int which = 0; which < 4; which++) {
        const int lg = which == 0 ? small_log2 : big_log2;
        const uint64_t n = 1ull << lg;
        uint64_t *table;
        hipError_t ea;
        if (which == 2) ea = hipExtMallocWithFlags((void **)&table, n * 8 + 64, hipDeviceMallocUncached);
        else if (which == 3) ea = hipExtMallocWithFlags((void **)&table, n * 8 + 64, hipDeviceMallocFinegrained);
        else ea = hipMalloc(&table, n * 8 + 64);
        printf("alloc: %s\n", alloc_names[which]);
        if (ea != hipSuccess) { printf("alloc of %llu rows failed: %s\n", (unsigned long long)n, hipGetErrorString(ea)); continue; }
        hipLaunchKernelGGL(fill_table, dim3(4096), dim3(256), 0, 0, table, n);
        CHECK(hipDeviceSynchronize());
        const uint32_t n_rows = (uint32_t)(n - 8);
        printf("table: 2^%d rows = %.1f MB\n", lg, n * 8 / 1e6);
        for (uint64_t lanes : {(uint64_t)1 << 18, (uint64_t)1 << 20}) {
            double ms[14];
            ms[0] = run<0>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[1] = run<1>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[2] = run<2>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[3] = run<3>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[4] = run<4>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[5] = run<5>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[6] = run<6>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[7] = run<7>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[8] = run<8>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[9] = run<9>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[10] = run<10>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[11] = run<11>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[12] = run<12>(table, n_rows, steps, lanes, sink, out, stream, 5);
            ms[13] = run<13>(table, n_rows, steps, lanes, sink, out, stream, 5);
            for (int v = 0; v < 14; v++)
                printf("  lanes=%8llu  %-32s %8.3f ms  %7.2f Gsteps/s\n", (unsigned long long)lanes, names[v], ms[v],
                       lanes * (double)steps / ms[v] / 1e6);
        }
        CHECK(hipFree(table));
    }
    return 0;
}
