// Probe (no torch): what does the DATA phase of the decode-step attention cost?  Kernels that only read the K and V rows of a
// 40-head x S x 128 fp16 cache (Llama-2-13B decode, 21 MB at S = 1025) with the attention kernel's geometry -- 256 threads,
// 16 lanes per 256-byte row, a workgroup instruction = 16 rows = 4 KiB -- in different work assignments, chain-timed over 20
// rotating caches (443 MB > Infinity Cache) like tools/attn_bench.py times the real kernel.
//   mode 0  interleaved 16-row blocks (block split + u * splits), buffer loads, all blocks requested at once   (the round-6 kernel)
//   mode 1  the same with nt loads
//   mode 2  contiguous chunks of ceil(S / splits) rows, all requested at once
//   mode 3  contiguous chunks, trips of 4 blocks, one trip ahead (the round 2-5 kernel's request pattern)
//   mode 4  mode 0 + the per-position VALU work of the real kernel (dot, 16-lane sum, exp, fp32 accumulation)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/attn_probe.hip -o tools/attn_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kU = 12;

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, moved);
}

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void probe_kernel(const uint8_t* __restrict__ kc, const uint8_t* __restrict__ vc, int S,
                                                        unsigned* __restrict__ sink)
{
    constexpr int BLK = THREADS / 16;  // rows per workgroup instruction
    const int split = blockIdx.x, splits = gridDim.x, h = blockIdx.y;
    const int tid = threadIdx.x, rib = tid >> 4, d0b = (tid & 15) * 16;
    const size_t head = (size_t)h * S * 256;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(kc + head), 0, S * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(vc + head), 0, S * 256, 0x00020000);
    constexpr int AUX = (MODE == 1 || MODE >= 5) ? 2 : 0;
    const int nblk = (S + BLK - 1) / BLK;
    u32 acc = 0;
    if constexpr (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 4 || MODE >= 5) {
        const int chunk_blocks = (nblk + splits - 1) / splits;
        u32x4     k[kU], v[kU];
        unsigned  off[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int blk = MODE == 2 ? split * chunk_blocks + u : split + u * splits;
            const int j   = blk * BLK + rib;
            const bool in = j < S && (MODE != 2 || u < chunk_blocks);
            off[u]        = in ? (unsigned)j * 256u + d0b : 0x80000000u;
        }
        u32x4 small[7];
        if constexpr (MODE == 6 || MODE == 7) {
            // seven 16-byte loads of L2-resident lines first in the queue (q, k, v of the new token, the pairs, cos | sin)
#pragma unroll
            for (int i = 0; i < 7; ++i) small[i] = *reinterpret_cast<const u32x4*>(kc + (size_t)(h * 7 + i) * 256 + d0b);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) k[u] = __builtin_amdgcn_raw_buffer_load_b128(rk, (int)off[u], 0, AUX);
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rv, (int)off[u], 0, AUX);
        if constexpr (MODE == 5 || MODE == 7) {
            // the real kernel's out-of-range loads: one 2-byte mask load per block from an EMPTY descriptor, and a further trip
            // of 4 + 4 16-byte loads beyond the chunk
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(kc), 0, 0, 0x00020000);
            unsigned short mm[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) mm[u] = __builtin_amdgcn_raw_buffer_load_b16(rm, (int)(off[u] >> 7), 0, 0);
            u32x4 ex[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ex[u] = __builtin_amdgcn_raw_buffer_load_b128(u < 4 ? rk : rv, (int)(0x80000000u + u * 16), 0, AUX);
#pragma unroll
            for (int u = 0; u < kU; ++u) acc ^= mm[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= ex[u].x ^ ex[u].w;
        }
        if constexpr (MODE == 6 || MODE == 7) {
#pragma unroll
            for (int i = 0; i < 7; ++i) acc ^= small[i].x ^ small[i].y;
        }
        if constexpr (MODE == 4) {
            float m = -INFINITY, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const f16x2 q2 = {(f16)0.01f, (f16)0.02f};
            float       sc[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const f16x8 kk = __builtin_bit_cast(f16x8, k[u]);
                float       a  = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) a = __builtin_amdgcn_fdot2(f16x2{kk[2 * i], kk[2 * i + 1]}, q2, a, false);
                a = dpp_add<0xB1>(a);
                a = dpp_add<0x4E>(a);
                a = dpp_add<0x141>(a);
                a = dpp_add<0x140>(a);
                sc[u] = a * 0.088f;
                m     = fmaxf(m, sc[u]);
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const float p  = __expf(sc[u] - m);
                const f16x8 vv = __builtin_bit_cast(f16x8, v[u]);
                l += p;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = fmaf(p, (float)vv[i], o[i]);
            }
            float s = l;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += o[i];
            acc = __builtin_bit_cast(u32, s);
        } else {
#pragma unroll
            for (int u = 0; u < kU; ++u) acc ^= k[u].x ^ k[u].y ^ k[u].z ^ k[u].w ^ v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    } else {
        // contiguous chunk in trips of 4 blocks, next trip requested before the current one is consumed
        const int chunk = (S + splits - 1) / splits, j0 = split * chunk, j1 = min(S, j0 + chunk);
        u32x4     k[4], v[4], kn[4], vn[4];
        auto load = [&](u32x4(&kk)[4], u32x4(&vv)[4], int jb) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = jb + u * BLK + rib;
                const unsigned o = j < j1 ? (unsigned)j * 256u + d0b : 0x80000000u;
                kk[u] = __builtin_amdgcn_raw_buffer_load_b128(rk, (int)o, 0, 0);
                vv[u] = __builtin_amdgcn_raw_buffer_load_b128(rv, (int)o, 0, 0);
            }
        };
        load(k, v, j0);
        for (int jb = j0;;) {
            const int jn = jb + 4 * BLK;
            load(kn, vn, jn);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc ^= k[u].x ^ k[u].y ^ k[u].z ^ k[u].w ^ v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
            if (jn >= j1) break;
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = kn[u], v[u] = vn[u];
            jb = jn;
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

static double time_graph(const std::function<void(int, hipStream_t)>& plain, int iters, int reps = 5)
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipGraph_t     g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < iters; ++i) plain(i, s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        auto t0 = std::chrono::high_resolution_clock::now();
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::high_resolution_clock::now();
        best    = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(s));
    return best;
}

int main(int argc, char** argv)
{
    const int H = 40, S = argc > 1 ? atoi(argv[1]) : 1025, L = 20, CAP = 1082;
    std::vector<uint8_t*> kc(L), vc(L);
    std::vector<uint8_t>  host((size_t)H * CAP * 256);
    srand(1);
    for (auto& b : host) b = (uint8_t)(rand() >> 7) & 0x3f;
    for (int i = 0; i < L; ++i) {
        CK(hipMalloc(&kc[i], host.size()));
        CK(hipMalloc(&vc[i], host.size()));
        CK(hipMemcpy(kc[i], host.data(), host.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(vc[i], host.data(), host.size(), hipMemcpyHostToDevice));
    }
    unsigned* sink;
    CK(hipMalloc(&sink, 64));
    const double mb = 2.0 * H * S * 256 / 1e6;
    printf("decode-attention data phase probe: %d heads x %d rows x 256 B, K + V = %.1f MB per step, %d rotating caches\n", H, S, mb, L);
    auto run = [&](const char* name, auto kern, int splits, int threads) {
        // NOTE: the caches were allocated with CAP rows per head; the probe reads them as S-row heads (addresses stay inside)
        const double us = time_graph(
            [&](int i, hipStream_t s) { hipLaunchKernelGGL(kern, dim3(splits, H), dim3(threads), 0, s, kc[i % L], vc[i % L], S, sink); },
            400);
        printf("%-58s splits %2d (%4d workgroups x %3d thr) %6.2f us/step  %5.0f GB/s\n", name, splits, splits * H, threads, us, mb / us * 1e3);
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (int sp : {6, 8, 10, 12}) run("0 interleaved blocks, all at once", probe_kernel<0, 256>, sp, 256);
        for (int sp : {6, 8}) run("1 interleaved blocks, all at once, nt", probe_kernel<1, 256>, sp, 256);
        for (int sp : {6, 8, 10}) run("2 contiguous chunk, all at once", probe_kernel<2, 256>, sp, 256);
        for (int sp : {6, 8, 12}) run("3 contiguous chunk, trips of 4 blocks one ahead", probe_kernel<3, 256>, sp, 256);
        for (int sp : {6, 8, 10}) run("4 = 0 + per-position VALU work", probe_kernel<4, 256>, sp, 256);
        for (int sp : {6, 8}) run("5 = 1 + out-of-range mask / further-trip loads", probe_kernel<5, 256>, sp, 256);
        for (int sp : {6, 8}) run("6 = 1 + seven small L2-resident loads first", probe_kernel<6, 256>, sp, 256);
        for (int sp : {6, 8}) run("7 = 1 + both", probe_kernel<7, 256>, sp, 256);
    }
    return 0;
}
