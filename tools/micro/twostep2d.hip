// Micro-benchmark: one launch per time step against one launch per TWO time steps (overlapped tiles, the first
// step recomputed on a halo of R) for the 2-D 16th-order forward update on the benchmark grid: 8 shots of
// 420 x 220 points.  No sources or receivers; both variants start from the same random state and the program
// checks that they agree.  Chains are replayed from a HIP graph, as the library does.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/twostep2d.hip -o twostep2d && ./twostep2d
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int R = 8, H = 16;              // stencil radius, allocated halo (two radii)
constexpr int N1 = 420, N2 = 220;         // grid (x, z), z contiguous
constexpr int PITCH = 256, ROWS = N1 + 2 * H;
constexpr long FIELD = (long)ROWS * PITCH;
constexpr int TX = 32, TZ = 32, NTX = (N1 + TX - 1) / TX, NTZ = (N2 + TZ - 1) / TZ;

struct Args {
    const float *cur, *old, *m, *eta;
    float *out1, *out2;
    float c0, c[R + 1], dt, dt2;
    int nshots;
};

__device__ inline long gidx(int x, int z) { return (long)(x + H) * PITCH + (z + H); }

__device__ inline float update(float lap, float uc, float uo, float m, float eta, float dt, float dt2)
{
    const float ed = eta * dt;
    return (dt2 * lap + 2.0f * m * uc - m * uo + ed * uc) / (m + ed);
}

// ---- one step per launch: 32 x 32 tile + halo 8 in LDS, four z points per thread ---------------------------
__global__ __launch_bounds__(256) void step1(Args a)
{
    constexpr int W = TZ + 2 * R, HT = TX + 2 * R, P = 96;       // pitch 96: conflict-free 16-byte reads
    __shared__ __attribute__((aligned(16))) float s[HT * P];
    const int tid = threadIdx.x, shot = blockIdx.x % a.nshots, tile = blockIdx.x / a.nshots;
    const int x0 = (tile / NTZ) * TX, z0 = (tile % NTZ) * TZ;
    const float *cur = a.cur + shot * FIELD, *old = a.old + shot * FIELD;
    float *out = a.out1 + shot * FIELD;
    for (int e = tid; e < HT * (W / 4); e += 256) {
        const int r = e / (W / 4), q = e - r * (W / 4);
        float4 v = make_float4(0, 0, 0, 0);
        if (x0 - R + r < N1 + H) v = *reinterpret_cast<const float4 *>(cur + gidx(x0 - R + r, z0 - R + 4 * q));
        *reinterpret_cast<float4 *>(&s[r * P + 4 * q]) = v;
    }
    const int tx = tid / 8, tq = tid % 8, x = x0 + tx, z = z0 + 4 * tq;
    const bool live = x < N1 && z < N2;
    float4 m4 = make_float4(1, 1, 1, 1), e4 = make_float4(0, 0, 0, 0), o4 = e4;
    if (live) {
        m4 = *reinterpret_cast<const float4 *>(a.m + gidx(x, z));
        e4 = *reinterpret_cast<const float4 *>(a.eta + gidx(x, z));
        o4 = *reinterpret_cast<const float4 *>(old + gidx(x, z));
    }
    __syncthreads();
    if (!live) return;
    const float *ctr = &s[(tx + R) * P + R + 4 * tq];
    float f[4 + 2 * R], lap[4];
    for (int q = 0; q < (4 + 2 * R) / 4; ++q) *reinterpret_cast<float4 *>(&f[4 * q]) = *reinterpret_cast<const float4 *>(ctr - R + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) lap[i] = a.c0 * f[R + i];
#pragma unroll
    for (int k = 1; k <= R; ++k) {
        const float4 lo = *reinterpret_cast<const float4 *>(ctr - k * P), hi = *reinterpret_cast<const float4 *>(ctr + k * P);
        lap[0] += a.c[k] * (f[R - k] + f[R + k] + lo.x + hi.x);
        lap[1] += a.c[k] * (f[R + 1 - k] + f[R + 1 + k] + lo.y + hi.y);
        lap[2] += a.c[k] * (f[R + 2 - k] + f[R + 2 + k] + lo.z + hi.z);
        lap[3] += a.c[k] * (f[R + 3 - k] + f[R + 3 + k] + lo.w + hi.w);
    }
    const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w}, oo[4] = {o4.x, o4.y, o4.z, o4.w};
    float res[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) res[i] = (z + i < N2) ? update(lap[i], f[R + i], oo[i], mm[i], ee[i], a.dt, a.dt2) : 0.f;
    *reinterpret_cast<float4 *>(out + gidx(x, z)) = make_float4(res[0], res[1], res[2], res[3]);
}

// ---- two steps per launch -----------------------------------------------------------------------------
// LDS A: u[t] on the tile + halo 2R (64 x 64); LDS B: u[t+1] on the tile + halo R (48 x 48), recomputed here
__global__ __launch_bounds__(256) void step2(Args a)
{
    constexpr int WA = TZ + 4 * R, HA = TX + 4 * R, PA = 96;     // 64 x 64, pitch 96
    constexpr int WB = TZ + 2 * R, HB = TX + 2 * R, PB = 96;     // 48 x 48
    __shared__ __attribute__((aligned(16))) float sa[HA * PA];
    __shared__ __attribute__((aligned(16))) float sb[HB * PB];
    const int tid = threadIdx.x, shot = blockIdx.x % a.nshots, tile = blockIdx.x / a.nshots;
    const int x0 = (tile / NTZ) * TX, z0 = (tile % NTZ) * TZ;
    const float *cur = a.cur + shot * FIELD, *old = a.old + shot * FIELD;
    float *out1 = a.out1 + shot * FIELD, *out2 = a.out2 + shot * FIELD;
    for (int e = tid; e < HA * (WA / 4); e += 256) {
        const int r = e / (WA / 4), q = e - r * (WA / 4);
        const int gx = x0 - 2 * R + r;
        float4 v = make_float4(0, 0, 0, 0);
        if (gx < N1 + H) v = *reinterpret_cast<const float4 *>(cur + gidx(gx, z0 - 2 * R + 4 * q));
        *reinterpret_cast<float4 *>(&sa[r * PA + 4 * q]) = v;
    }
    // coefficients and old level of this thread's first-step points: 48 rows x 12 float4 = 576 -> 3 passes
    constexpr int NP = (HB * (WB / 4) + 255) / 256;
    float4 m4[NP], e4[NP], o4[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int e = tid + 256 * p, r = e / (WB / 4), q = e - r * (WB / 4);
        const int x = x0 - R + r, z = z0 - R + 4 * q;
        m4[p] = make_float4(1, 1, 1, 1), e4[p] = o4[p] = make_float4(0, 0, 0, 0);
        if (e < HB * (WB / 4) && x >= 0 && x < N1 && z + 3 >= 0 && z < N2) {
            m4[p] = *reinterpret_cast<const float4 *>(a.m + gidx(x, z));
            e4[p] = *reinterpret_cast<const float4 *>(a.eta + gidx(x, z));
            o4[p] = *reinterpret_cast<const float4 *>(old + gidx(x, z));
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int e = tid + 256 * p, r = e / (WB / 4), q = e - r * (WB / 4);
        if (e >= HB * (WB / 4)) continue;
        const int x = x0 - R + r, z = z0 - R + 4 * q;
        const float *ctr = &sa[(r + R) * PA + R + 4 * q];
        float f[4 + 2 * R], lap[4];
        for (int qq = 0; qq < (4 + 2 * R) / 4; ++qq) *reinterpret_cast<float4 *>(&f[4 * qq]) = *reinterpret_cast<const float4 *>(ctr - R + 4 * qq);
#pragma unroll
        for (int i = 0; i < 4; ++i) lap[i] = a.c0 * f[R + i];
#pragma unroll
        for (int k = 1; k <= R; ++k) {
            const float4 lo = *reinterpret_cast<const float4 *>(ctr - k * PA), hi = *reinterpret_cast<const float4 *>(ctr + k * PA);
            lap[0] += a.c[k] * (f[R - k] + f[R + k] + lo.x + hi.x);
            lap[1] += a.c[k] * (f[R + 1 - k] + f[R + 1 + k] + lo.y + hi.y);
            lap[2] += a.c[k] * (f[R + 2 - k] + f[R + 2 + k] + lo.z + hi.z);
            lap[3] += a.c[k] * (f[R + 3 - k] + f[R + 3 + k] + lo.w + hi.w);
        }
        const float mm[4] = {m4[p].x, m4[p].y, m4[p].z, m4[p].w}, ee[4] = {e4[p].x, e4[p].y, e4[p].z, e4[p].w},
                    oo[4] = {o4[p].x, o4[p].y, o4[p].z, o4[p].w};
        float res[4];
        const bool row_in = x >= 0 && x < N1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            res[i] = (row_in && z + i >= 0 && z + i < N2) ? update(lap[i], f[R + i], oo[i], mm[i], ee[i], a.dt, a.dt2) : 0.f;
        const float4 v = make_float4(res[0], res[1], res[2], res[3]);
        *reinterpret_cast<float4 *>(&sb[r * PB + 4 * q]) = v;
        // the tile's own part of u[t+1] goes to memory as well
        if (r >= R && r < R + TX && q >= R / 4 && q < R / 4 + TZ / 4 && x < N1 && z < N2)
            *reinterpret_cast<float4 *>(out1 + gidx(x, z)) = v;
    }
    __syncthreads();
    const int tx = tid / 8, tq = tid % 8, x = x0 + tx, z = z0 + 4 * tq;
    if (!(x < N1 && z < N2)) return;
    const float4 mq = *reinterpret_cast<const float4 *>(a.m + gidx(x, z)), eq = *reinterpret_cast<const float4 *>(a.eta + gidx(x, z));
    const float4 uo = *reinterpret_cast<const float4 *>(&sa[(tx + 2 * R) * PA + 2 * R + 4 * tq]);   // u[t] is the old level now
    const float *ctr = &sb[(tx + R) * PB + R + 4 * tq];
    float f[4 + 2 * R], lap[4];
    for (int q = 0; q < (4 + 2 * R) / 4; ++q) *reinterpret_cast<float4 *>(&f[4 * q]) = *reinterpret_cast<const float4 *>(ctr - R + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) lap[i] = a.c0 * f[R + i];
#pragma unroll
    for (int k = 1; k <= R; ++k) {
        const float4 lo = *reinterpret_cast<const float4 *>(ctr - k * PB), hi = *reinterpret_cast<const float4 *>(ctr + k * PB);
        lap[0] += a.c[k] * (f[R - k] + f[R + k] + lo.x + hi.x);
        lap[1] += a.c[k] * (f[R + 1 - k] + f[R + 1 + k] + lo.y + hi.y);
        lap[2] += a.c[k] * (f[R + 2 - k] + f[R + 2 + k] + lo.z + hi.z);
        lap[3] += a.c[k] * (f[R + 3 - k] + f[R + 3 + k] + lo.w + hi.w);
    }
    const float mm[4] = {mq.x, mq.y, mq.z, mq.w}, ee[4] = {eq.x, eq.y, eq.z, eq.w}, oo[4] = {uo.x, uo.y, uo.z, uo.w};
    float res[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) res[i] = (z + i < N2) ? update(lap[i], f[R + i], oo[i], mm[i], ee[i], a.dt, a.dt2) : 0.f;
    *reinterpret_cast<float4 *>(out2 + gidx(x, z)) = make_float4(res[0], res[1], res[2], res[3]);
}

int main(int argc, char **argv)
{
    const int shots = argc > 1 ? atoi(argv[1]) : 8, steps = 1500;   // batch size: 8 fills the GPU, 1-4 do not
    const size_t fb = (size_t)shots * FIELD * sizeof(float);
    float *lv[4], *ref[3], *m, *eta;
    for (auto &p : lv) { CK(hipMalloc(&p, fb)); }
    for (auto &p : ref) { CK(hipMalloc(&p, fb)); }
    CK(hipMalloc(&m, FIELD * 4)); CK(hipMalloc(&eta, FIELD * 4));
    std::vector<float> hm(FIELD, 1.f), he(FIELD, 0.f), u0((size_t)shots * FIELD, 0.f), u1((size_t)shots * FIELD, 0.f);
    srand(1);
    for (int x = 0; x < N1; ++x)
        for (int z = 0; z < N2; ++z) {
            const long i = (long)(x + H) * PITCH + z + H;
            const float v = 1.5f + 3.0f * z / N2;
            hm[i] = 1.0f / (v * v);
            const int d = std::min(std::min(x, N1 - 1 - x), std::min(z, N2 - 1 - z));
            he[i] = d < 40 ? 0.02f * (40 - d) / 40.f : 0.f;
            for (int s = 0; s < shots; ++s) {
                u0[s * FIELD + i] = (rand() % 2001 - 1000) * 1e-3f;
                u1[s * FIELD + i] = u0[s * FIELD + i] * 0.99f;
            }
        }
    CK(hipMemcpy(m, hm.data(), FIELD * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(eta, he.data(), FIELD * 4, hipMemcpyHostToDevice));
    Args a{};
    a.m = m, a.eta = eta, a.nshots = shots;
    // 16th-order second-derivative weights on h = 30 m, dt = 3 ms
    const double w[9] = {-3.05484410430839, 1.77777777777778, -0.311111111111111, 0.0754208754208754, -0.0176767676767677,
                         0.00348096348096348, -0.000518000518000518, 5.07429078857650e-05, -2.42812742812743e-06};
    a.c0 = (float)(2 * w[0] / 900.0);
    for (int k = 1; k <= R; ++k) a.c[k] = (float)(w[k] / 900.0);
    a.dt = 3.0f, a.dt2 = 9.0f;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nb = NTX * NTZ * shots;
    std::vector<float> r1((size_t)shots * FIELD), r2((size_t)shots * FIELD);
    for (int variant = 0; variant < 2; ++variant) {
        float **L = variant == 0 ? ref : lv;
        const int nl = variant == 0 ? 3 : 4;
        for (int i = 0; i < nl; ++i) CK(hipMemset(L[i], 0, fb));
        CK(hipMemcpy(L[0], u0.data(), fb, hipMemcpyHostToDevice));     // level t-1
        CK(hipMemcpy(L[1], u1.data(), fb, hipMemcpyHostToDevice));     // level t
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        if (variant == 0) {
            for (int t = 1; t <= steps; ++t) {
                a.cur = L[t % 3], a.old = L[(t + 2) % 3], a.out1 = L[(t + 1) % 3], a.out2 = nullptr;
                hipLaunchKernelGGL(step1, dim3(nb), dim3(256), 0, s, a);
            }
        } else {
            for (int t = 1; t <= steps; t += 2) {
                a.cur = L[t % 4], a.old = L[(t + 3) % 4], a.out1 = L[(t + 1) % 4], a.out2 = L[(t + 2) % 4];
                hipLaunchKernelGGL(step2, dim3(nb), dim3(256), 0, s, a);
            }
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            for (int i = 0; i < nl; ++i) CK(hipMemsetAsync(L[i], 0, fb, s));
            CK(hipMemcpyAsync(L[0], u0.data(), fb, hipMemcpyHostToDevice, s));
            CK(hipMemcpyAsync(L[1], u1.data(), fb, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        // final level = index (steps + 1)
        CK(hipMemcpy((variant == 0 ? r1 : r2).data(), L[(steps + 1) % nl], fb, hipMemcpyDeviceToHost));
        printf("%d shots, %s: %.3f ms for %d steps = %.3f us per time step\n", shots, variant == 0 ? "one step per launch " : "two steps per launch",
               best, steps, best * 1e3 / steps);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    double num = 0, den = 0;
    for (size_t i = 0; i < r1.size(); ++i) num += (double)(r1[i] - r2[i]) * (r1[i] - r2[i]), den += (double)r1[i] * r1[i];
    printf("relative difference of the final level: %.3e (norm %.3e)\n", std::sqrt(num / (den > 0 ? den : 1)), std::sqrt(den));
    return 0;
}
