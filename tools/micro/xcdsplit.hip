// Micro-benchmark: the neighbour exchange of a whole-loop kernel when the tiles of ONE shot are spread over several
// XCDs, so that halos cross from one XCD's L2 to another's.  As xcdsync.hip: every block reads its XCC id, takes a
// tile, and per step waits for its four neighbours' counters, reads one float4 per thread from each neighbour's tile
// of the previous level (checking every value read for staleness), stores its own tile, waits for the stores and
// raises its counter.  Here XCD x serves shot x / split, and within that shot the band of tiles number x % split.
// Usage: xcdsplit [nsteps=2000] [nshots=8] [split=1] [store=0|16] [flag=0|16] [load=16] [work=0] [skip=0] [period=1]
//   store: cache policy of the tile stores (0 plain, 16 sc1 = written through to memory, 17 sc0 sc1)
//   flag:  0 = counter stored with group scope (plain store), 16 = agent scope (sc1 store)
//   load:  cache policy of counter and halo loads (16 sc1, 17 sc0 sc1, 2 nt)
//   work:  s_sleep units spent per step between the halo loads and the tile store (stands for the stencil)
//   skip:  links of the chain left out, to cost them by difference (results are then meaningless and the staleness
//          check is off): 1 the halo loads, 2 the wait for the tile stores before the counter moves, 4 the tile
//          stores, 8 the wait for the neighbours' counters
//   period: exchange (wait, halo loads, acknowledged stores, counter) only every period-th step; the steps between
//          only store their tile and meet at the block's barrier (the cost model of a ring 'period' radii wide)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int vi4 __attribute__((ext_vector_type(4)));
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ vf4 raw_buffer_load_f4(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ int raw_buffer_load_i1(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ void raw_buffer_store_f4(vf4 v, vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ void raw_buffer_store_i1(int v, vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i32");
__device__ inline vi4 buf_of(const void *p)
{
    union { struct { const void *p; unsigned n, f; } s; vi4 v; } r;
    r.s.p = p; r.s.n = 0xffffffffu; r.s.f = 0x00020000u;
    return r.v;
}

constexpr int NTX = 14, NTZ = 7, NTILE = NTX * NTZ, TPTS = 1024;
constexpr int SPIN_LIMIT = 1 << 18;

struct Args {
    float *buf;        // [3][nshots][NTILE][TPTS]
    int *done;         // [nshots][NTILE]
    int *counter;      // [16] blocks taken per XCC
    int *abort_flag;
    int *mismatch;
    int *xcc_seen;
    int nsteps, nshots, split, work, skip, period;
};

template <int STORE, int FLAG, int LOAD>
__global__ __launch_bounds__(256) void k_persist(Args a)
{
    __shared__ int s_slot, s_xcc, s_fail;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;
        s_xcc = xcc;
        s_slot = atomicAdd(&a.counter[xcc], 1);
        atomicAdd(&a.xcc_seen[xcc], 1);
        s_fail = 0;
    }
    __syncthreads();
    const int shot = s_xcc / a.split, part = s_xcc % a.split;
    const int lo = part * NTILE / a.split, hi = (part + 1) * NTILE / a.split;
    const int tile = lo + s_slot;
    if (shot >= a.nshots || tile >= hi) return;   // block-uniform
    const int tx = tile / NTZ, tz = tile % NTZ;
    int nb = -1;
    if (tid == 0 && tx > 0) nb = tile - NTZ;
    if (tid == 1 && tx < NTX - 1) nb = tile + NTZ;
    if (tid == 2 && tz > 0) nb = tile - 1;
    if (tid == 3 && tz < NTZ - 1) nb = tile + 1;
    const int n0 = tx > 0 ? tile - NTZ : -1, n1 = tx < NTX - 1 ? tile + NTZ : -1, n2 = tz > 0 ? tile - 1 : -1,
              n3 = tz < NTZ - 1 ? tile + 1 : -1;
    int *done = a.done + shot * NTILE;
    const vi4 doneB = buf_of(done);
    const long lvl = (long)a.nshots * NTILE * TPTS;
    int bad = 0;
    for (int t = 1; t <= a.nsteps; ++t) {
        int fail = 0;
        const bool exch = (t % a.period) == 0 || a.period == 1;
        if (nb >= 0 && exch && !(a.skip & 8)) {
            int n = 0;
            while (raw_buffer_load_i1(doneB, nb * 4, 0, LOAD) < t - a.period) {
                if (++n > SPIN_LIMIT || ((n & 255) == 0 && raw_buffer_load_i1(buf_of(a.abort_flag), 0, 0, LOAD))) {
                    fail = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (fail) s_fail = 1;
        __syncthreads();
        if (s_fail) {
            if (tid == 0) raw_buffer_store_i1(1, buf_of(a.abort_flag), 0, 0, 16);
            return;
        }
        // three rotating levels as in the real kernel: the level written now was last read two steps ago
        const float *prev = a.buf + (long)((t - 1) % 3) * lvl + (long)shot * NTILE * TPTS;
        float s = 0.f;
        if (t > 1 && exch && !(a.skip & 1)) {
            vf4 v0{}, v1{}, v2{}, v3{};
            if (n0 >= 0) v0 = raw_buffer_load_f4(buf_of(prev), (n0 * TPTS + tid * 4) * 4, 0, LOAD);
            if (n1 >= 0) v1 = raw_buffer_load_f4(buf_of(prev), (n1 * TPTS + tid * 4) * 4, 0, LOAD);
            if (n2 >= 0) v2 = raw_buffer_load_f4(buf_of(prev), (n2 * TPTS + tid * 4) * 4, 0, LOAD);
            if (n3 >= 0) v3 = raw_buffer_load_f4(buf_of(prev), (n3 * TPTS + tid * 4) * 4, 0, LOAD);
            if (n0 >= 0 && v0.x != n0 * 4096.f + (t - 1) && a.skip == 0 && a.period == 1) ++bad;
            if (n1 >= 0 && v1.x != n1 * 4096.f + (t - 1) && a.skip == 0 && a.period == 1) ++bad;
            if (n2 >= 0 && v2.x != n2 * 4096.f + (t - 1) && a.skip == 0 && a.period == 1) ++bad;
            if (n3 >= 0 && v3.x != n3 * 4096.f + (t - 1) && a.skip == 0 && a.period == 1) ++bad;
            s = v0.y + v1.y + v2.y + v3.y;
        }
        for (int w = 0; w < a.work; ++w) __builtin_amdgcn_s_sleep(8);
        float *cur = a.buf + (long)(t % 3) * lvl + (long)shot * NTILE * TPTS + (long)tile * TPTS;
        if (!(a.skip & 4)) raw_buffer_store_f4(vf4{tile * 4096.f + t, s * 1e-30f, 0.f, 0.f}, buf_of(cur), tid * 16, 0, STORE);
        if (!(a.skip & 2) && exch) __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0 && exch) raw_buffer_store_i1(t, doneB, tile * 4, 0, FLAG);
    }
    if (bad) atomicAdd(a.mismatch, bad);
}

template <int STORE, int FLAG>
static void launch_l(int load, int nblocks, Args a)
{
    switch (load) {
        case 2: hipLaunchKernelGGL((k_persist<STORE, FLAG, 2>), dim3(nblocks), dim3(256), 0, 0, a); break;
        case 17: hipLaunchKernelGGL((k_persist<STORE, FLAG, 17>), dim3(nblocks), dim3(256), 0, 0, a); break;
        default: hipLaunchKernelGGL((k_persist<STORE, FLAG, 16>), dim3(nblocks), dim3(256), 0, 0, a);
    }
}
static void launch(int store, int flag, int load, int nblocks, Args a)
{
    if (store == 0 && flag == 0) launch_l<0, 0>(load, nblocks, a);
    else if (store == 0) launch_l<0, 16>(load, nblocks, a);
    else if (store == 17) launch_l<17, 17>(load, nblocks, a);
    else if (flag == 0) launch_l<16, 0>(load, nblocks, a);
    else launch_l<16, 16>(load, nblocks, a);
}

int main(int argc, char **argv)
{
    const int nsteps = argc > 1 ? atoi(argv[1]) : 2000, nshots = argc > 2 ? atoi(argv[2]) : 8, split = argc > 3 ? atoi(argv[3]) : 1;
    const int store = argc > 4 ? atoi(argv[4]) : 0, flag = argc > 5 ? atoi(argv[5]) : 0, load = argc > 6 ? atoi(argv[6]) : 16;
    const int work = argc > 7 ? atoi(argv[7]) : 0, skip = argc > 8 ? atoi(argv[8]) : 0, period = argc > 9 ? atoi(argv[9]) : 1;
    if (nshots * split > 8 || split < 1) { printf("nshots * split must not exceed 8\n"); return 1; }
    Args a;
    a.nsteps = nsteps, a.nshots = nshots, a.split = split, a.work = work, a.skip = skip, a.period = period < 1 ? 1 : period;
    const size_t nbuf = (size_t)3 * nshots * NTILE * TPTS;
    CK(hipMalloc(&a.buf, sizeof(float) * nbuf));
    int *flags;
    const int nflag = nshots * NTILE + 16 + 1 + 1 + 16;
    CK(hipMalloc(&flags, sizeof(int) * nflag));
    a.done = flags, a.counter = flags + nshots * NTILE, a.abort_flag = a.counter + 16, a.mismatch = a.abort_flag + 1,
    a.xcc_seen = a.mismatch + 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int per_xcd = 112, nblocks = 8 * per_xcd;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(flags, 0, sizeof(int) * nflag));
        CK(hipMemset(a.buf, 0, sizeof(float) * nbuf));
        CK(hipEventRecord(e0));
        launch(store, flag, load, nblocks, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<int> h(nflag);
        CK(hipMemcpy(h.data(), flags, sizeof(int) * nflag, hipMemcpyDeviceToHost));
        int mindone = 1 << 30;
        for (int i = 0; i < nshots * NTILE; ++i) mindone = h[i] < mindone ? h[i] : mindone;
        printf("rep %d: shots %d split %d store %d flag %d load %d work %d skip %d period %d: %.3f us/step  abort=%d mismatch=%d min_done=%d\n", rep,
               nshots, split, store, flag, load, work, skip, a.period, ms * 1e3 / nsteps, h[nshots * NTILE + 16], h[nshots * NTILE + 17], mindone);
    }
    return 0;
}
