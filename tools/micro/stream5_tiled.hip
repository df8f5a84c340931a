// The access pattern of the 2.5-D step without its arithmetic: blocks of 256 threads own a TY x TZ tile of the (y, z)
// plane (TY * TZ = 1024, four z points per thread) and march along x over a chunk of planes, reading NR streams and
// writing one, 16 bytes per lane.  What the memory system delivers for this pattern bounds the step from below.
// Usage: stream5_tiled [n=592] [halo=8] [TZ=64] [nchunk=6] [nread=4] [depth=1 planes in flight] [blocks per CU=3 (by LDS)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int NR, int DEPTH>
__global__ __launch_bounds__(256) void tiled_kernel(const float *a, const float *b, const float *c, const float *d, float *out,
                                                    int n, long s0, long s1, long off0, int TZ, int nty, int ntz, int nchunk, int xchunk)
{
    extern __shared__ float pad[];
    if (threadIdx.x == 1023) pad[0] = 0.f;   // keeps the allocation that limits the blocks per CU
    const int per_xcd = (gridDim.x + 7) / 8;
    int work = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (work >= nty * ntz * nchunk) return;
    const int tz = work % ntz; work /= ntz;
    const int ty = work % nty; const int chunk = work / nty;
    const int TY = 1024 / TZ, TQ = TZ / 4;
    const int y = ty * TY + threadIdx.x / TQ, z = tz * TZ + 4 * (threadIdx.x % TQ);
    if (y >= n || z >= n) return;
    const long col = off0 + (long)y * s1 + z;
    const int xa = chunk * xchunk, xb = min(n, xa + xchunk);
    vf4 va[DEPTH], vb[DEPTH], vc[DEPTH], vd[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
        const long o = col + (long)min(xa + k, xb - 1) * s0;
        va[k] = *(const vf4 *)(a + o);
        if (NR > 1) vb[k] = __builtin_nontemporal_load((const vf4 *)(b + o));
        if (NR > 2) vc[k] = __builtin_nontemporal_load((const vf4 *)(c + o));
        if (NR > 3) vd[k] = __builtin_nontemporal_load((const vf4 *)(d + o));
    }
    for (int x = xa; x < xb; x += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            if (x + k >= xb) break;
            vf4 v = va[k];
            if (NR > 1) v += vb[k];
            if (NR > 2) v += vc[k];
            if (NR > 3) v += vd[k];
            const long o = col + (long)min(x + k + DEPTH, xb - 1) * s0;
            va[k] = *(const vf4 *)(a + o);
            if (NR > 1) vb[k] = __builtin_nontemporal_load((const vf4 *)(b + o));
            if (NR > 2) vc[k] = __builtin_nontemporal_load((const vf4 *)(c + o));
            if (NR > 3) vd[k] = __builtin_nontemporal_load((const vf4 *)(d + o));
            __builtin_nontemporal_store(v, (vf4 *)(out + col + (long)(x + k) * s0));
        }
    }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 592, halo = argc > 2 ? atoi(argv[2]) : 8;
    const int TZ = argc > 3 ? atoi(argv[3]) : 64, nchunk = argc > 4 ? atoi(argv[4]) : 6;
    const int nread = argc > 5 ? atoi(argv[5]) : 4, depth = argc > 6 ? atoi(argv[6]) : 1, bpc = argc > 7 ? atoi(argv[7]) : 3;
    const int lds = bpc >= 8 ? 0 : (160 * 1024 / bpc - 1024) / 16 * 16;
    const long np = n + 2 * halo, s1 = np, s0 = np * np, off0 = halo * s0 + halo * s1 + halo, field = np * np * np;
    float *buf[5];
    for (int i = 0; i < 5; ++i) { CK(hipMalloc(&buf[i], field * 4)); CK(hipMemset(buf[i], 0, field * 4)); }
    const int TY = 1024 / TZ, nty = (n + TY - 1) / TY, ntz = (n + TZ - 1) / TZ, xchunk = (n + nchunk - 1) / nchunk;
    const int nblocks = (nty * ntz * nchunk + 7) / 8 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 8; ++rep) {
        CK(hipEventRecord(e0));
#define L(NR, D) hipLaunchKernelGGL((tiled_kernel<NR, D>), dim3(nblocks), dim3(256), lds, 0, buf[0], buf[1], buf[2], buf[3], buf[4], n, s0, s1, off0, TZ, nty, ntz, nchunk, xchunk)
        if (nread == 4 && depth == 1) L(4, 1); else if (nread == 4 && depth == 2) L(4, 2); else if (nread == 4 && depth == 4) L(4, 4);
        else if (nread == 1 && depth == 1) L(1, 1); else if (nread == 1 && depth == 4) L(1, 4); else { printf("unsupported\n"); return 1; }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (nread + 1) * 4.0 * n * (double)n * n;
    printf("n=%d tile %dx%d chunks=%d reads=%d depth=%d blocks=%d per CU %d: %.4f ms  %.3f TB/s\n", n, TY, TZ, nchunk, nread, depth, nblocks, bpc, best, bytes / best / 1e9);
    return 0;
}
