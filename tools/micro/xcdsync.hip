// Micro-benchmark: what does one time step cost when the blocks of a shot stay resident for the whole time
// loop and synchronise with their four neighbour tiles through flags, all inside one XCD?
// Each block reads its XCC id, takes a tile of the shot assigned to that XCD, and per step: waits for the
// neighbours' flags, reads one float4 per thread from each neighbour's tile of the previous level, checks
// it, stores its own tile of the new level, waits for the stores, raises its flag.
// Every wait is bounded: a block that waits too long sets the abort flag and every block leaves.
// Usage: xcdsync [nsteps=2000] [nshots=8] [mode=0]   mode 0: agent-scope flags + L1 invalidate after the wait, 1: no invalidate (stale reads),
//   2: flags and neighbour data read with group-scope loads (miss the L1, hit the XCD's L2), plain flag store
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
__device__ inline vi4 buf_of(const void *p)
{
    union { struct { const void *p; unsigned n, f; } s; vi4 v; } r;
    r.s.p = p; r.s.n = 0xffffffffu; r.s.f = 0x00020000u;
    return r.v;
}
constexpr int kSc0 = 1;   // group scope: misses the per-CU L1, may hit the XCD's L2

// mode = cache-policy bits of the flag and neighbour loads: 1 sc0, 2 nt, 16 sc1 (and sums); +256: agent-scope flag store
__device__ inline int flag_load(vi4 b, int off, int mode)
{
    switch (mode & 255) {
        case 1: return raw_buffer_load_i1(b, off, 0, 1);
        case 2: return raw_buffer_load_i1(b, off, 0, 2);
        case 3: return raw_buffer_load_i1(b, off, 0, 3);
        case 16: return raw_buffer_load_i1(b, off, 0, 16);
        case 17: return raw_buffer_load_i1(b, off, 0, 17);
        case 18: return raw_buffer_load_i1(b, off, 0, 18);
        default: return raw_buffer_load_i1(b, off, 0, 19);
    }
}
__device__ inline vf4 data_load(vi4 b, int off, int mode)
{
    switch (mode & 255) {
        case 1: return raw_buffer_load_f4(b, off, 0, 1);
        case 2: return raw_buffer_load_f4(b, off, 0, 2);
        case 3: return raw_buffer_load_f4(b, off, 0, 3);
        case 16: return raw_buffer_load_f4(b, off, 0, 16);
        case 17: return raw_buffer_load_f4(b, off, 0, 17);
        case 18: return raw_buffer_load_f4(b, off, 0, 18);
        default: return raw_buffer_load_f4(b, off, 0, 19);
    }
}

constexpr int NTX = 14, NTZ = 7, NTILE = NTX * NTZ, TPTS = 1024;
constexpr int SPIN_LIMIT = 1 << 16;

struct Args {
    float *buf;        // [2][nshots][NTILE][TPTS]
    int *done;         // [nshots][NTILE]
    int *counter;      // [16] blocks taken per XCC
    int *abort_flag;   // 1 = a wait ran out
    int *mismatch;     // values read that were not the expected ones
    int *xcc_seen;     // [16] blocks seen per XCC (diagnostic)
    int nsteps, nshots, mode;
};

__global__ __launch_bounds__(256) void k_persist(Args a)
{
    __shared__ int s_slot, s_xcc;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        s_xcc = xcc;
        s_slot = atomicAdd(&a.counter[xcc], 1);
        atomicAdd(&a.xcc_seen[xcc], 1);
    }
    __syncthreads();
    const int shot = s_xcc, tile = s_slot;
    if (shot >= a.nshots || tile >= NTILE) return;   // block-uniform
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
        if (nb >= 0) {
            int n = 0;
            if (a.mode & 512) {
                // pipelined polling: four loads in flight, a fresh answer every few dozen cycles instead of every
                // L2 round trip
                while (true) {
                    const int v0 = raw_buffer_load_i1(doneB, nb * 4, 0, 16);
                    __builtin_amdgcn_s_sleep(1);
                    const int v1 = raw_buffer_load_i1(doneB, nb * 4, 0, 16);
                    __builtin_amdgcn_s_sleep(1);
                    const int v2 = raw_buffer_load_i1(doneB, nb * 4, 0, 16);
                    __builtin_amdgcn_s_sleep(1);
                    const int v3 = raw_buffer_load_i1(doneB, nb * 4, 0, 16);
                    if (v0 >= t - 1) break;
                    if (v1 >= t - 1) break;
                    if (v2 >= t - 1) break;
                    if (v3 >= t - 1) break;
                    if (++n > SPIN_LIMIT) {
                        fail = 1;
                        break;
                    }
                }
            } else
            while (flag_load(doneB, nb * 4, a.mode) < t - 1) {
                if (++n > SPIN_LIMIT || __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    fail = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        fail = __syncthreads_or(fail);
        if (fail) {
            if (tid == 0) __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const float *prev = a.buf + (long)((t - 1) & 1) * lvl + (long)shot * NTILE * TPTS;
        float s = 0.f;
        if (t > 1) {
            if (n0 >= 0) { vf4 v = data_load(buf_of(prev), (n0 * TPTS + tid * 4) * 4, a.mode); if (v.x != n0 * 4096.f + (t - 1)) ++bad; s += v.y; }
            if (n1 >= 0) { vf4 v = data_load(buf_of(prev), (n1 * TPTS + tid * 4) * 4, a.mode); if (v.x != n1 * 4096.f + (t - 1)) ++bad; s += v.y; }
            if (n2 >= 0) { vf4 v = data_load(buf_of(prev), (n2 * TPTS + tid * 4) * 4, a.mode); if (v.x != n2 * 4096.f + (t - 1)) ++bad; s += v.y; }
            if (n3 >= 0) { vf4 v = data_load(buf_of(prev), (n3 * TPTS + tid * 4) * 4, a.mode); if (v.x != n3 * 4096.f + (t - 1)) ++bad; s += v.y; }
        }
        float *cur = a.buf + (long)(t & 1) * lvl + (long)shot * NTILE * TPTS + (long)tile * TPTS;
        reinterpret_cast<float4 *>(cur)[tid] = make_float4(tile * 4096.f + t, s * 1e-30f, 0.f, 0.f);
        __builtin_amdgcn_s_waitcnt(0);   // every counter to zero: the stores of this wave have reached the L2
        __syncthreads();
        if (tid == 0) {
            if (a.mode & 256)
                __hip_atomic_store(&done[tile], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                __hip_atomic_store(&done[tile], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (bad) atomicAdd(a.mismatch, bad);
}


// Mailbox variant: no counters.  A block writes, for each of its four neighbours, a strip of 16-byte chunks
// (three payload floats + the step number) into that neighbour's mailbox of the step's parity, and a reader polls
// the chunks themselves until they carry the step it wants: one round trip instead of counter + data, and the
// writer never waits for its stores.  CHUNKS chunks per edge.
constexpr int CHUNKS = 86;
__device__ void raw_buffer_store_f4(vf4 v, vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__global__ __launch_bounds__(256) void k_mailbox(Args a)
{
    __shared__ int s_slot, s_xcc;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        s_xcc = xcc;
        s_slot = atomicAdd(&a.counter[xcc], 1);
        atomicAdd(&a.xcc_seen[xcc], 1);
    }
    __syncthreads();
    const int shot = s_xcc, tile = s_slot;
    if (shot >= a.nshots || tile >= NTILE) return;
    const int tx = tile / NTZ, tz = tile % NTZ;
    const int nb[4] = {tx > 0 ? tile - NTZ : -1, tx < NTX - 1 ? tile + NTZ : -1, tz > 0 ? tile - 1 : -1, tz < NTZ - 1 ? tile + 1 : -1};
    // mailbox of (shot, tile, edge e, parity p): CHUNKS float4; edge e of a tile is written by neighbour nb[e]
    // through that neighbour's opposite edge (e ^ 1)
    float *box = a.buf + (long)shot * NTILE * 4 * 2 * CHUNKS * 4;
    const vi4 boxB = buf_of(box);
    auto box_off = [&](int t, int e, int par, int c) { return ((((t * 4 + e) * 2 + par) * CHUNKS) + c) * 16; };
    // each thread handles chunks tid and tid + 256 of the 4 * CHUNKS incoming ones
    int bad = 0;
    for (int t = 1; t <= a.nsteps; ++t) {
        // write own strips for step t (payload encodes writer and step)
        for (int c = tid; c < 4 * CHUNKS; c += 256) {
            const int e = c / CHUNKS, cc = c - e * CHUNKS;
            if (nb[e] < 0) continue;
            raw_buffer_store_f4(vf4{tile * 4096.f + t, (float)cc, 0.f, __int_as_float(t)}, boxB, box_off(nb[e], e ^ 1, t & 1, cc), 0, 0);
        }
        // read the strips the neighbours wrote for step t
        int fail = 0;
        for (int c = tid; c < 4 * CHUNKS; c += 256) {
            const int e = c / CHUNKS, cc = c - e * CHUNKS;
            if (nb[e] < 0) continue;
            int n = 0;
            vf4 v;
            while (true) {
                v = raw_buffer_load_f4(boxB, box_off(tile, e, t & 1, cc), 0, 16);
                if (__float_as_int(v.w) == t) break;
                if (++n > SPIN_LIMIT || ((n & 63) == 0 && __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    fail = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!fail && (v.x != nb[e] * 4096.f + t || v.y != (float)cc)) ++bad;
        }
        fail = __syncthreads_or(fail);
        if (fail) {
            if (tid == 0) __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __syncthreads();   // stands for the barrier after staging the halo in LDS
        if (tid == 0) a.done[shot * NTILE + tile] = t;
    }
    if (bad) atomicAdd(a.mismatch, bad);
}

int main(int argc, char **argv)
{
    const int nsteps = argc > 1 ? atoi(argv[1]) : 2000, nshots = argc > 2 ? atoi(argv[2]) : 8, mode = argc > 3 ? atoi(argv[3]) : 0;
    Args a;
    a.nsteps = nsteps, a.nshots = nshots, a.mode = mode;
    const size_t nbuf = std::max((size_t)2 * nshots * NTILE * TPTS, (size_t)nshots * NTILE * 4 * 2 * CHUNKS * 4);
    CK(hipMalloc(&a.buf, sizeof(float) * nbuf));
    CK(hipMemset(a.buf, 0, sizeof(float) * nbuf));
    int *flags;
    const int nflag = nshots * NTILE + 16 + 1 + 1 + 16;
    CK(hipMalloc(&flags, sizeof(int) * nflag));
    a.done = flags, a.counter = flags + nshots * NTILE, a.abort_flag = a.counter + 16, a.mismatch = a.abort_flag + 1,
    a.xcc_seen = a.mismatch + 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // more blocks than tiles per XCD so that an uneven hand-out still fills every shot; the surplus leaves at once
    const int per_xcd = 112, nblocks = 8 * per_xcd;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(flags, 0, sizeof(int) * nflag));
        CK(hipMemset(a.buf, 0, sizeof(float) * nbuf));
        CK(hipEventRecord(e0));
        if (mode == 1000)
            hipLaunchKernelGGL(k_mailbox, dim3(nblocks), dim3(256), 0, 0, a);
        else
            hipLaunchKernelGGL(k_persist, dim3(nblocks), dim3(256), 0, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<int> h(nflag);
        CK(hipMemcpy(h.data(), flags, sizeof(int) * nflag, hipMemcpyDeviceToHost));
        int mindone = 1 << 30;
        for (int i = 0; i < nshots * NTILE; ++i) mindone = h[i] < mindone ? h[i] : mindone;
        printf("rep %d: %d steps, %d shots, mode %d: %.3f ms = %.3f us/step  abort=%d mismatch=%d min_done=%d  blocks per xcc:", rep,
               nsteps, nshots, mode, ms, ms * 1e3 / nsteps, h[nshots * NTILE + 16], h[nshots * NTILE + 17], mindone);
        for (int i = 0; i < 8; ++i) printf(" %d", h[nshots * NTILE + 18 + i]);
        printf("\n");
    }
    return 0;
}
