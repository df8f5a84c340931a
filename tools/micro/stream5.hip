// What the memory system delivers for the access mix of the 3-D time step, without any stencil: NR read streams and
// one written stream of N floats each, 16 bytes per lane, grid-stride over float4.  Prints TB/s (all streams).
// Usage: stream5 [n_floats=207474688 (592^3)] [nread=4] [nt=1: non-temporal hints on all but the first read stream]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int NR, bool NT>
__global__ __launch_bounds__(256) void mix_kernel(const vf4 *__restrict__ a, const vf4 *__restrict__ b, const vf4 *__restrict__ c,
                                                  const vf4 *__restrict__ d, vf4 *__restrict__ out, long n4)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        vf4 v = a[i];
        if (NR > 1) v += NT ? __builtin_nontemporal_load(&b[i]) : b[i];
        if (NR > 2) v += NT ? __builtin_nontemporal_load(&c[i]) : c[i];
        if (NR > 3) v += NT ? __builtin_nontemporal_load(&d[i]) : d[i];
        if (NT) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
    }
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 207474688L;
    const int nread = argc > 2 ? atoi(argv[2]) : 4;
    const int nt = argc > 3 ? atoi(argv[3]) : 1;
    const long n4 = n / 4;
    float *buf[5];
    for (int i = 0; i < 5; ++i) { CK(hipMalloc(&buf[i], n * 4)); CK(hipMemset(buf[i], 0, n * 4)); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64, (int)((n4 + 255) / 256)}) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
#define LAUNCH(NR, NTF) hipLaunchKernelGGL((mix_kernel<NR, NTF>), dim3(blocks), dim3(256), 0, 0, (const vf4 *)buf[0], (const vf4 *)buf[1], (const vf4 *)buf[2], (const vf4 *)buf[3], (vf4 *)buf[4], n4)
            if (nread == 1) { if (nt) LAUNCH(1, true); else LAUNCH(1, false); }
            else if (nread == 2) { if (nt) LAUNCH(2, true); else LAUNCH(2, false); }
            else if (nread == 3) { if (nt) LAUNCH(3, true); else LAUNCH(3, false); }
            else { if (nt) LAUNCH(4, true); else LAUNCH(4, false); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("n=%ld reads=%d nt=%d blocks=%d: %.4f ms  %.3f TB/s\n", n, nread, nt, blocks, best, (nread + 1) * n * 4.0 / best / 1e9);
    }
    return 0;
}
