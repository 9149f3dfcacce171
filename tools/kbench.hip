// Kernel micro-benchmark harness (no torch): times kernel variants on one MI355X with
//   (a) per-dispatch begin/end timestamps (hipExtLaunchKernelGGL start/stop events == what rocprof reports),
//   (b) wall time per step of a HIP graph holding a chain of dependent launches.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/kbench.hip -o tools/kbench
//   (tools/kbench_stamps: the same with -DEETQ_KBENCH_STAMPS -- device-clock stamps inside the kernels)
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../eetq_amd/csrc/gemv_kernel.hpp"
#include "../eetq_amd/csrc/gemm_kernel.hpp"
#include "../eetq_amd/csrc/streamk_kernel.hpp"
#include "../eetq_amd/csrc/gemm_mid_kernel.hpp"
#include "../eetq_amd/csrc/gemm_splitk_kernel.hpp"

namespace eetq {  // stubs for the error plumbing declared in common.hpp (unused by the kernels)
void set_error(const std::string&) {}
int  fail(int c, const std::string&) { return c; }
int  check_hip(hipError_t e, const char*) { return e == hipSuccess ? 0 : -2; }
ProfEvents next_prof_events() { return {}; }
}  // namespace eetq

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

using eetq::u32x4;

struct Stats {
    double mean, med, mn, p10, p90;
};

static Stats stats_of(std::vector<float>& v)
{
    std::sort(v.begin(), v.end());
    double s = 0;
    for (float f : v) s += f;
    size_t n = v.size();
    return {s / n, v[n / 2], v[0], v[n / 10], v[n * 9 / 10]};
}

// launch(i, start, stop): enqueue launch i with dispatch timestamps into (start, stop)
static Stats time_dispatch(const std::function<void(int, hipEvent_t, hipEvent_t)>& launch, int iters, int warm = 50)
{
    std::vector<hipEvent_t> a(iters), b(iters);
    for (int i = 0; i < iters; ++i) {
        CK(hipEventCreate(&a[i]));
        CK(hipEventCreate(&b[i]));
    }
    hipEvent_t wa, wb;
    CK(hipEventCreate(&wa));
    CK(hipEventCreate(&wb));
    for (int i = 0; i < warm; ++i) launch(i, wa, wb);
    CK(hipDeviceSynchronize());
    for (int i = 0; i < iters; ++i) launch(i, a[i], b[i]);
    CK(hipDeviceSynchronize());
    std::vector<float> us(iters);
    for (int i = 0; i < iters; ++i) {
        float ms;
        CK(hipEventElapsedTime(&ms, a[i], b[i]));
        us[i] = ms * 1e3f;
        CK(hipEventDestroy(a[i]));
        CK(hipEventDestroy(b[i]));
    }
    return stats_of(us);
}

// plain(i, stream): enqueue launch i on stream (capturable).  Returns us per step of a graph replay.
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
        auto   t1 = std::chrono::high_resolution_clock::now();
        double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
        best      = std::min(best, us);
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(s));
    return best;
}

// ---- pure streaming read: the floor for "read 16 MiB once" on this chip -------------------------------
template <int LOADS, bool NT>
__global__ void stream_read_kernel(const u32x4* __restrict__ p, unsigned* __restrict__ out)
{
    const u32x4* q = p + (size_t)blockIdx.x * blockDim.x * LOADS + threadIdx.x;
    u32x4        v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) v[i] = eetq::gemv::load_w<NT>(q + (size_t)i * blockDim.x);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x9e3779b9u) out[0] = acc;  // practically never: keeps the loads alive
}

// ---- GEMV ablation ladder (round 6): the shipping M = 1 kernel (16 waves x 4 tiles, exact fit, x in registers) rebuilt one
// ingredient at a time on top of the load-only kernel of the same geometry.  LEVEL:
//   0  weight loads only (xor of the bytes kept alive by a store that never executes)
//   1  + the scale and activation loads queued ahead of the weights
//   2  + dequant and v_dot2 (the accumulator kept alive the same way)
//   3  + the wave's xor-16 / xor-32 butterflies
//   4  + LDS write, barrier, wave 0 sums the 16 waves (+ its butterflies)
//   5  + the 32-byte store                        == gemv_kernel<1, 16, 4, true, true, 1, 8>'s instruction stream
//   6  level 5 with the store issued write-through (sc0 sc1): nothing dirty in L2 at the end of the kernel
//   7  level 5 with the cross-wave sum finished by 4 waves (one per 4 columns) instead of wave 0
//   8  level 5 with the activations staged per WAVE through LDS: one 16-byte load per lane (the wave's 512 bytes of x, lanes
//      32..63 duplicate) instead of eight, a ds_write_b128, eight ds_read_b128 -- no workgroup barrier in front of the math
//   9  level 1 with the scale load only (no activation loads)
//  10  level 8 with the 512 bytes fetched by lanes 0..31 only (exec-masked load)
//  11  level 5 with the activations fetched ONCE per wave (8 bytes per lane: the 16 lanes of a k-group hold the group's 32 dwords
//      of x, two each) and handed to the dot products by DPP row broadcast on the v_dot2c operand: no LDS, no extra instruction
// acc += wq[i] . x dword (tile d, i), the x dword read from lane d * 4 + (i >> 1) of this lane's 16-lane row (row_newbcast)
template <int SRC>
__device__ __forceinline__ float dot2_bcast(eetq::f16x2 wq, eetq::u32 xreg, float acc)
{
    asm volatile("v_dot2c_f32_f16_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(xreg), "v"(wq), "n"(SRC));
    return acc;
}
__device__ __forceinline__ void ladder_dot_bcast(const eetq::f16x2 (&wq)[8], eetq::u32 xp0, eetq::u32 xp1, int d, float& acc)
{
#define EETQ_LD(dd)                                                  \
    case dd:                                                         \
        acc = dot2_bcast<dd * 4 + 0>(wq[0], xp0, acc);               \
        acc = dot2_bcast<dd * 4 + 0>(wq[1], xp1, acc);               \
        acc = dot2_bcast<dd * 4 + 1>(wq[2], xp0, acc);               \
        acc = dot2_bcast<dd * 4 + 1>(wq[3], xp1, acc);               \
        acc = dot2_bcast<dd * 4 + 2>(wq[4], xp0, acc);               \
        acc = dot2_bcast<dd * 4 + 2>(wq[5], xp1, acc);               \
        acc = dot2_bcast<dd * 4 + 3>(wq[6], xp0, acc);               \
        acc = dot2_bcast<dd * 4 + 3>(wq[7], xp1, acc);               \
        break;
    switch (d) { EETQ_LD(0) EETQ_LD(1) EETQ_LD(2) EETQ_LD(3) }
#undef EETQ_LD
}

template <int LEVEL>
__global__ __launch_bounds__(1024, 8) void gemv_ladder_kernel(const eetq::f16* __restrict__ x, const uint8_t* __restrict__ w,
                                                               const eetq::f16* __restrict__ scales, eetq::f16* __restrict__ y,
                                                               int N, int K, unsigned* __restrict__ sink)
{
    using namespace eetq;
    constexpr int WAVES = 16, D = 4;
    __shared__ float red[WAVES * 16];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int g = lane >> 4, c = lane & 15, ntile = blockIdx.x, KT = K / 64;
    u32   sraw = 0;
    u32x4 xr[D * 2];
    __shared__ __attribute__((aligned(16))) u32x4 xw[WAVES * 32];  // levels 8 / 10: 512 bytes of x per wave
    u32x4 xstage = {};
    if constexpr (LEVEL >= 1) sraw = reinterpret_cast<const uint16_t*>(scales)[ntile * 16 + c];
    u32 xp0 = 0, xp1 = 0;
    if constexpr (LEVEL == 11 || LEVEL == 12) {
        // lane (g, c): dwords 2c, 2c + 1 of row g's list [tile d][dword i] = [d * 8 + i]
        const u32* p = reinterpret_cast<const u32*>(x + (wave + (c >> 2) * WAVES) * 64 + 16 * g + 4 * (c & 3));
        const auto v = *reinterpret_cast<const __attribute__((ext_vector_type(2))) u32*>(p);
        xp0 = v.x, xp1 = v.y;
        if constexpr (LEVEL == 12) __builtin_amdgcn_sched_barrier(0);  // the activation load stays ahead of the weight stream
    } else if constexpr (LEVEL == 8 || LEVEL == 10) {
        // lane L (mod 32): 16-byte piece L & 7 of the wave's tile d = (L >> 3) & 3
        const int L = lane & 31;
        const u32x4* p = reinterpret_cast<const u32x4*>(x + (wave + (L >> 3) * WAVES) * 64) + (L & 7);
        if (LEVEL == 8 || lane < 32) xstage = *p;
    } else if constexpr (LEVEL >= 1 && LEVEL != 9) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const u32x4* p = reinterpret_cast<const u32x4*>(x + (wave + d * WAVES) * 64 + 16 * g);
            xr[d * 2]      = p[0];
            xr[d * 2 + 1]  = p[1];
        }
    }
    const u32x4* wp = reinterpret_cast<const u32x4*>(w + (size_t)ntile * KT * 1024) + wave * 64 + lane;
    u32x4        buf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) buf[d] = eetq::gemv::load_w<true>(wp + (size_t)d * WAVES * 64);
    if constexpr (LEVEL == 8 || LEVEL == 10) {
        // wave-private staging: LDS operations of one wave execute in order, no barrier needed
        if (lane < 32) xw[wave * 32 + lane] = xstage;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int d = 0; d < D; ++d) {
            xr[d * 2]     = xw[wave * 32 + d * 8 + 2 * g];
            xr[d * 2 + 1] = xw[wave * 32 + d * 8 + 2 * g + 1];
        }
    }
    if constexpr (LEVEL <= 1 || LEVEL == 9) {
        unsigned a = sraw;
#pragma unroll
        for (int d = 0; d < D; ++d) a ^= buf[d].x ^ buf[d].y ^ buf[d].z ^ buf[d].w;
        if constexpr (LEVEL == 1) {
#pragma unroll
            for (int d = 0; d < 2 * D; ++d) a ^= xr[d].x ^ xr[d].y ^ xr[d].z ^ xr[d].w;
        }
        if (a == 0x9e3779b9u) sink[0] = a;
        return;
    } else {
        asm volatile("" : "+v"(sraw));
        const f16x2 scale2 = as_f16x2(sraw | (sraw << 16));
        float       acc    = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            f16x2 wq[8];
            dequant_16(buf[d], scale2, wq);
            if constexpr (LEVEL == 11 || LEVEL == 12) {
                ladder_dot_bcast(wq, xp0, xp1, d, acc);
                continue;
            }
            const u32x4 xa = xr[d * 2], xb = xr[d * 2 + 1];
            const u32   xd[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_fdot2(wq[i], as_f16x2(xd[i]), acc, false);
        }
        if constexpr (LEVEL == 2) {
            if (__builtin_bit_cast(u32, acc) == 0x9e3779b9u) sink[0] = 1;
            return;
        }
        acc = sum_xor32(sum_xor16(acc));
        if constexpr (LEVEL == 3) {
            if (__builtin_bit_cast(u32, acc) == 0x9e3779b9u) sink[0] = 1;
            return;
        }
        if (lane < 16) red[wave * 16 + lane] = acc;
        __syncthreads();
        if constexpr (LEVEL == 7) {
            // waves 0..3: wave v finishes columns 4v..4v+3; lane = 4 * (wave index being summed) + column-in-quad
            if (wave < 4) {
                const int cq = lane & 3, ws = lane >> 2;   // 16 waves x 4 columns = 64 lanes
                float     s  = red[ws * 16 + wave * 4 + cq];
                s += __shfl_xor(s, 4, 64);
                s += __shfl_xor(s, 8, 64);
                s = sum_xor32(sum_xor16(s));
                if (lane < 4) y[ntile * 16 + wave * 4 + lane] = (f16)s;
            }
            return;
        }
        if (wave == 0) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) s += red[(g + 4 * wv) * 16 + c];
            s = sum_xor32(sum_xor16(s));
            if constexpr (LEVEL == 4) {
                if (__builtin_bit_cast(u32, s) == 0x9e3779b9u) sink[0] = 1;
                return;
            }
            if (lane < 16) {
                const f16 v = (f16)s;
                if constexpr (LEVEL == 6) {
                    const unsigned short b = __builtin_bit_cast(unsigned short, v);
                    asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(y + ntile * 16 + c), "v"((u32)b) : "memory");
                } else {
                    y[ntile * 16 + c] = v;
                }
            }
        }
    }
}

// L2 prefetch probe (round 6): every lane touches ONE dword of its own 64-byte sector (STRIDE = 64) or 128-byte line (STRIDE = 128)
// with a plain load; 256 threads, grid sized by the caller
template <int STRIDE>
__global__ void touch_lines_kernel(const uint8_t* __restrict__ p, size_t bytes, unsigned* __restrict__ out)
{
    // workgroup b touches the b-th 1 / gridDim of the bytes: with as many workgroups as the reader has, a toucher and the reader
    // of the same bytes have the same block id, i.e. (as the dispatcher deals blocks today) the same XCD and the same L2
    unsigned     acc   = 0;
    const size_t chunk = bytes / gridDim.x, base = (size_t)blockIdx.x * chunk;
    for (size_t off = (size_t)threadIdx.x * STRIDE; off < chunk; off += (size_t)blockDim.x * STRIDE)
        acc ^= *reinterpret_cast<const unsigned*>(p + base + off);
    if (acc == 0x9e3779b9u) out[0] = acc;
}

// dispatch-overhead probes: what a kernel costs that touches no memory / one cache line per wave
__global__ void empty_kernel_b(unsigned* out, int never)
{
    if (never == 7) out[1] = 2;
}
// the same with ~200 live registers (accumulators that are never stored unless `never` says so)
__global__ __launch_bounds__(256) void fat_kernel(unsigned* out, int never)
{
    float acc[192];
#pragma unroll
    for (int i = 0; i < 192; ++i) acc[i] = (float)(threadIdx.x + i);
    if (never == 7) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 192; ++i) acc[i] = acc[i] * acc[(i + 1) % 192] + (float)r;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 192; ++i) t += acc[i];
        out[2] = (unsigned)t;
    }
}
__global__ void empty_kernel(unsigned* out, int never)
{
    if (never == 12345) out[0] = threadIdx.x;
}
__global__ void touch_kernel(const unsigned* __restrict__ src, unsigned* out)
{
    const unsigned v = src[(size_t)blockIdx.x * 16384 + (threadIdx.x >> 6) * 32];  // one 128-byte line per wave
    if (v == 0x12345678u) out[0] = v;
}

static void bench_overhead(const std::vector<uint8_t*>& bufs, unsigned* out)
{
    for (int threads : {64, 1024}) {
        for (int grid : {1, 256, 1024}) {
            auto st = time_dispatch(
                [&](int, hipEvent_t a, hipEvent_t b) {
                    hipExtLaunchKernelGGL(empty_kernel, dim3(grid), dim3(threads), 0, 0, a, b, 0, out, 0);
                },
                400);
            printf("empty kernel      grid=%5d thr=%4d | disp mean %6.2f med %6.2f min %6.2f us\n", grid, threads, st.mean,
                   st.med, st.mn);
        }
    }
    for (int grid : {256}) {
        auto st = time_dispatch(
            [&](int i, hipEvent_t a, hipEvent_t b) {
                hipExtLaunchKernelGGL(touch_kernel, dim3(grid), dim3(1024), 0, 0, a, b, 0,
                                      (const unsigned*)bufs[i % bufs.size()], out);
            },
            400);
        printf("one HBM line/wave grid=%5d thr=1024 | disp mean %6.2f med %6.2f min %6.2f us\n", grid, st.mean, st.med,
               st.mn);
    }
}


// ---- same FLOPs on fewer CUs (round 4): does the chip pay idle CUs back as clock? -----------------------------------------
// one wave per workgroup records {XCC_ID, s_memtime (shader cycles), s_memrealtime (100 MHz)}; two launches around a chain give
// the average shader clock per XCD over the chain (eetq_amd/csrc/diag.hip has the library's copy)
__global__ __launch_bounds__(64) void kb_clock_stamp(unsigned long long* out)
{
    if (threadIdx.x != 0) return;
    unsigned long long tm, tr;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm), "=s"(tr)::"memory");
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    out[blockIdx.x * 3 + 0] = xcc;
    out[blockIdx.x * 3 + 1] = tm;
    out[blockIdx.x * 3 + 2] = tr;
}

struct ChainResult {
    double us, mhz;
};
// `iters` back-to-back launches between two stamp launches, captured as one graph, best of `reps` replays
static ChainResult time_chain_clock(const std::function<void(int, hipStream_t)>& plain, int iters, int reps = 4)
{
    const int           nwg = 512;
    unsigned long long *sa, *sb;
    CK(hipMalloc(&sa, nwg * 24));
    CK(hipMalloc(&sb, nwg * 24));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipGraph_t     g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(kb_clock_stamp, dim3(nwg), dim3(64), 0, s, sa);
    for (int i = 0; i < iters; ++i) plain(i, s);
    hipLaunchKernelGGL(kb_clock_stamp, dim3(nwg), dim3(64), 0, s, sb);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    ChainResult best{1e30, 0};
    std::vector<unsigned long long> ha(nwg * 3), hb(nwg * 3);
    for (int r = 0; r < reps; ++r) {
        auto t0 = std::chrono::high_resolution_clock::now();
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto         t1 = std::chrono::high_resolution_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
        CK(hipMemcpy(ha.data(), sa, nwg * 24, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), sb, nwg * 24, hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        for (unsigned x = 0; x < 8; ++x) {
            int ia = -1, ib = -1;
            for (int i = 0; i < nwg && (ia < 0 || ib < 0); ++i) {
                if (ia < 0 && ha[i * 3] == x) ia = i;
                if (ib < 0 && hb[i * 3] == x) ib = i;
            }
            if (ia >= 0 && ib >= 0 && hb[ib * 3 + 2] > ha[ia * 3 + 2])
                mhz.push_back((double)(hb[ib * 3 + 1] - ha[ia * 3 + 1]) / (double)(hb[ib * 3 + 2] - ha[ia * 3 + 2]) * 100.0);
        }
        std::sort(mhz.begin(), mhz.end());
        if (us < best.us) best = ChainResult{us, mhz.empty() ? 0.0 : mhz[mhz.size() / 2]};
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(s));
    CK(hipFree(sa));
    CK(hipFree(sb));
    return best;
}

template <int ABLATE, int J = 2>
static void bench_gemm_cu(const char* name, int M, int N, int K, const std::vector<uint8_t*>& bufs, const eetq::f16* x,
                          const eetq::f16* scales, eetq::f16* y)
{
    using namespace eetq::gemm;
    auto kern = gemm_tile_kernel<ABLATE, J, false, 2>;
    constexpr int SMEM_BYTES = TileCfg<J, 2>::SMEM_BYTES, BN = TileCfg<J, 2>::BN;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    const int    tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const double flops = 2.0 * M * N * K;
    auto r = time_chain_clock(
        [&](int i, hipStream_t s) {
            hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), SMEM_BYTES, s, x, (const uint8_t*)bufs[i % bufs.size()], scales, y, M, N,
                               K, N, eetq::Epilogue{});
        },
        100);
    const double mfma_cycles = (double)(K / 64) * 8 * J * 36.0;  // per wave: MFMAs x ~36 cycles each (32 passes + issue)
    printf("%-34s M=%5d N=%5d K=%5d wgs=%4d | %7.2f us/launch %6.0f MHz -> %6.1f k cycles (MFMA floor %5.1f k) %7.1f TF\n", name, M, N, K,
           tiles, r.us, r.mhz, r.us * r.mhz / 1e3, mfma_cycles / 1e3, flops / r.us / 1e6);
}

template <int LOADS, bool NT>
static void bench_stream(const char* name, int threads, const std::vector<uint8_t*>& bufs, size_t bytes, unsigned* out)
{
    const int grid = (int)(bytes / 16 / threads / LOADS);
    auto      st   = time_dispatch(
        [&](int i, hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL((stream_read_kernel<LOADS, NT>), dim3(grid), dim3(threads), 0, 0, a, b, 0,
                                  (const u32x4*)bufs[i % bufs.size()], out);
        },
        400);
    double g = time_graph(
        [&](int i, hipStream_t s) {
            hipLaunchKernelGGL((stream_read_kernel<LOADS, NT>), dim3(grid), dim3(threads), 0, s,
                               (const u32x4*)bufs[i % bufs.size()], out);
        },
        400);
    printf("%-34s grid=%5d thr=%4d | disp mean %6.2f med %6.2f min %6.2f p90 %6.2f us -> %6.0f GB/s(med) | graph %6.2f us/step\n",
           name, grid, threads, st.mean, st.med, st.mn, st.p90, bytes / st.med / 1e3, g);
}

template <int M, int WAVES, int D, bool EXACT, bool XREG, int XV, int OCC>
static void bench_gemv(const char* name, int N, int K, const std::vector<uint8_t*>& bufs, const eetq::f16* x,
                       const eetq::f16* scales, eetq::f16* y)
{
    const int    grid  = N / 16;
    const double bytes = (double)K * N + 2.0 * M * K + 2.0 * N + 2.0 * M * N;
    auto         kern  = eetq::gemv::gemv_kernel<M, WAVES, D, EXACT, XREG, XV, OCC>;
    const size_t smem  = eetq::gemv::gemv_smem_bytes(M, K, WAVES, XREG);
    if (smem > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    auto st = time_dispatch(
        [&](int i, hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), (unsigned)smem, 0, a, b, 0, x,
                                  (const uint8_t*)bufs[i % bufs.size()], scales, y, N, K, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, 0, 0.f);
        },
        400);
    double g = time_graph(
        [&](int i, hipStream_t s) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), (unsigned)smem, s, x,
                               (const uint8_t*)bufs[i % bufs.size()], scales, y, N, K, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, 0, 0.f);
        },
        400);
    printf("%-30s N=%5d K=%5d M=%d | disp mean %6.2f med %6.2f min %6.2f p90 %6.2f us -> %6.0f GB/s(med) | graph %6.2f us/step -> %6.0f GB/s\n",
           name, N, K, M, st.mean, st.med, st.mn, st.p90, bytes / st.med / 1e3, g, bytes / g / 1e3);
}

template <int ABLATE, int J = 2, int CW = 2>
static void bench_gemm(const char* name, int M, int N, int K, const std::vector<uint8_t*>& bufs, const eetq::f16* x,
                        const eetq::f16* scales, eetq::f16* y)
{
    using namespace eetq::gemm;
    auto kern = gemm_tile_kernel<ABLATE, J, false, CW>;
    constexpr int SMEM_BYTES = TileCfg<J, CW>::SMEM_BYTES, BN = TileCfg<J, CW>::BN;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    const int    tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const double flops = 2.0 * M * N * K;
    auto         st    = time_dispatch(
        [&](int i, hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(kern, dim3(tiles), dim3(128 * CW), SMEM_BYTES, 0, a, b, 0, x,
                                  (const uint8_t*)bufs[i % bufs.size()], scales, y, M, N, K, N, eetq::Epilogue{});
        },
        60, 10);
    printf("%-22s M=%5d N=%5d K=%5d | disp mean %7.2f med %7.2f min %7.2f us -> %7.1f TF(med)\n", name, M, N, K, st.mean,
           st.med, st.mn, flops / st.med / 1e6);
}

template <int MT, int NT, int WAVES, int D, int OCC>
static void bench_streamk(const char* name, int M, int N, int K, const std::vector<uint8_t*>& bufs, const eetq::f16* x,
                          const eetq::f16* scales, eetq::f16* y)
{
    const int    grid  = N / (16 * NT);
    const double bytes = (double)K * N + 2.0 * M * K + 2.0 * N + 2.0 * M * N;
    auto         kern  = eetq::streamk::streamk_kernel<MT, NT, WAVES, D, OCC>;
    const size_t smem  = eetq::streamk::streamk_smem_bytes(MT, NT, WAVES);
    if (smem > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    auto st = time_dispatch(
        [&](int i, hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), (unsigned)smem, 0, a, b, 0, x,
                                  (const uint8_t*)bufs[i % bufs.size()], scales, y, M, N, K, eetq::Epilogue{});
        },
        200);
    printf("%-30s N=%5d K=%5d M=%3d | disp mean %6.2f med %6.2f min %6.2f us -> %6.0f GB/s(med) %7.1f TF\n", name, N, K, M,
           st.mean, st.med, st.mn, bytes / st.med / 1e3, 2.0 * M * N * K / st.med / 1e6);
}

template <int MT, int STAGES>
static void bench_mid(const char* name, int M, int N, int K, const std::vector<uint8_t*>& bufs, const eetq::f16* x,
                      const eetq::f16* scales, eetq::f16* y)
{
    using namespace eetq::gemm_mid;
    using C   = Cfg<MT, STAGES>;
    auto kern = gemm_mid_kernel<MT, STAGES, true>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::kSmem));
    const int    tiles = ((N + kBN - 1) / kBN) * ((M + C::kRows - 1) / C::kRows);
    const double bytes = (double)K * N + 2.0 * M * K + 2.0 * N + 2.0 * M * N;
    auto st = time_dispatch(
        [&](int i, hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(kern, dim3(tiles), dim3(kThreads), C::kSmem, 0, a, b, 0, x,
                                  (const uint8_t*)bufs[i % bufs.size()], scales, y, M, N, K, eetq::Epilogue{});
        },
        200);
    printf("%-22s N=%5d K=%5d M=%3d | disp mean %6.2f med %6.2f min %6.2f us -> %6.0f GB/s(med) %7.1f TF\n", name, N, K, M,
           st.mean, st.med, st.mn, bytes / st.med / 1e3, 2.0 * M * N * K / st.med / 1e6);
}


template <int MT, int NB, int STAGES, int SB = STAGES, bool KFULL = true, int W = 4, bool INTER = false>
static void bench_splitk(const char* name, int M, int N, int K, int S, const std::vector<uint8_t*>& bufs, const eetq::f16* x,
                              const eetq::f16* scales, eetq::f16* y, float* slabs, unsigned* tickets)
{
    using namespace eetq::gemm_splitk;
    using C   = Cfg<MT, NB, STAGES, SB, W>;
    auto kern = gemm_splitk_kernel<MT, NB, STAGES, SB, KFULL, W, INTER>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int    tiles = (N + C::kBN - 1) / C::kBN;
    const double bytes = (double)K * N + 2.0 * M * K + 2.0 * N + 2.0 * M * N;
    auto st = time_dispatch(
        [&](int i, hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(kern, dim3(tiles * S), dim3(C::kThreads), C::kSmem, 0, a, b, 0, x,
                                  (const uint8_t*)bufs[i % bufs.size()], scales, y, M, N, K, S, slabs, tickets, eetq::Epilogue{});
        },
        200);
    const double g = time_graph(
        [&](int i, hipStream_t s) {
            hipLaunchKernelGGL(kern, dim3(tiles * S), dim3(C::kThreads), C::kSmem, s, x, (const uint8_t*)bufs[i % bufs.size()],
                               scales, y, M, N, K, S, slabs, tickets, eetq::Epilogue{});
        },
        200);
    printf("%-26s N=%5d K=%5d M=%3d BN=%2d S=%d ring %dx%d w%d wg=%4d | disp mean %6.2f med %6.2f min %6.2f | graph step %6.2f us -> %6.0f GB/s %6.1f TF (graph)\n",
           name, N, K, M, 32 * NB, S, STAGES, SB, W, tiles * S, st.mean, st.med, st.mn, g, bytes / g / 1e3,
           2.0 * M * N * K / g / 1e6);
}

// ---- data-path probe: how many bytes per second can one CU pull from L2 (a) into LDS by LDS-DMA, (b) into registers?
// Every workgroup streams PER_WG bytes out of a `region`-byte window (window <= 4 MiB: L2-resident after the first pass;
// 8 MiB: the M=1024 GEMM's activation matrix, served by L2 + Infinity Cache).  PATTERN 0: contiguous 1 KiB per wave
// instruction (weight tiles);  PATTERN 1: 8 rows x 128 B at an 8 KiB row stride (activation tile of the tiled GEMM).
template <int MODE, int PATTERN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void datapath_kernel(const uint8_t* __restrict__ src, unsigned* __restrict__ out,
                                                             int region, int per_wg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, region, 0x00020000);
    const int pieces = per_wg / 1024 / WAVES;  // per wave
    // workgroups of one XCD start at different offsets of the window, like tiles of different rows / columns
    unsigned pos = (blockIdx.x * 37u + wave) * 1024u;
    const int voff = PATTERN == 0 ? lane * 16 : (lane >> 3) * 8192 + (lane & 7) * 16;
    u32x4 acc = {0, 0, 0, 0};
    constexpr int RING = 8;
    for (int i = 0; i < pieces; i += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            unsigned base;
            if (PATTERN == 0) {
                base = pos % (unsigned)region;
                pos += WAVES * 1024u;
            } else {  // piece = 8 rows x 128 B: walk 128-B column blocks of an 8-row band, then the next band
                const unsigned pc = pos >> 10, band = pc >> 6, col = pc & 63;
                base = (band * 65536u + col * 128u) % (unsigned)(region - 65536);
                pos += WAVES * 1024u;
            }
            if (MODE == 0) {
                eetq::gemm::dma16(rsrc, voff, (int)base, smem + (wave * RING + r) * 1024);
            } else {
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (int)base, 0));
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) acc.x = reinterpret_cast<unsigned*>(smem)[tid];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int MODE, int PATTERN, int WAVES>
static void bench_datapath(const char* name, const uint8_t* src, unsigned* out, int region, int per_wg, int grid = 256)
{
    auto kern = datapath_kernel<MODE, PATTERN, WAVES>;
    const unsigned smem = MODE == 0 ? WAVES * 8 * 1024 : 1024 * 4;
    if (smem > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    auto st = time_dispatch(
        [&](int, hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), smem, 0, a, b, 0, src, out, region, per_wg);
        },
        40, 10);
    const double bytes = (double)per_wg * grid;
    printf("%-34s grid %3d region %5d KiB %5d KiB/WG | med %7.2f us -> %6.1f GB/s per CU, %6.2f TB/s total\n", name, grid,
           region >> 10, per_wg >> 10, st.med, per_wg / st.med / 1e3, bytes / st.med / 1e6);
}


#ifdef EETQ_KBENCH_STAMPS
// load-only kernel with device-clock stamps: geometry study for the 16 MiB stream (which launch shape gets the bytes on
// chip soonest after the first wave starts?)
template <int LOADS, bool NT>
__global__ void stream_read_stamped(const u32x4* __restrict__ p, unsigned* __restrict__ out, unsigned long long* __restrict__ st)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const u32x4* q = p + (size_t)blockIdx.x * blockDim.x * LOADS + threadIdx.x;
    u32x4        v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) v[i] = eetq::gemv::load_w<NT>(q + (size_t)i * blockDim.x);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    unsigned long long t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc) : "memory");
    if (acc == 0x9e3779b9u) out[0] = acc;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        st[w * 2]     = t0;
        st[w * 2 + 1] = t1;
    }
}

template <int LOADS, bool NT>
static void bench_geometry(const char* name, int threads, const std::vector<uint8_t*>& bufs, size_t bytes, unsigned* out,
                           unsigned long long* st)
{
    const int grid  = (int)(bytes / ((size_t)threads * LOADS * 16));
    const int waves = grid * threads / 64;
    std::vector<unsigned long long> h((size_t)waves * 2);
    std::vector<float> span, ramp, ev;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int r = -10; r < 200; ++r) {
        // a short burst of plain launches keeps the clocks up; the last one of the burst is the measured one
        for (int k = 0; k < 4; ++k)
            hipLaunchKernelGGL((stream_read_stamped<LOADS, NT>), dim3(grid), dim3(threads), 0, 0,
                               (const u32x4*)bufs[(r + 10 + k) % bufs.size()], out, st);
        hipExtLaunchKernelGGL((stream_read_stamped<LOADS, NT>), dim3(grid), dim3(threads), 0, 0, a, b, 0,
                              (const u32x4*)bufs[(r + 17) % bufs.size()], out, st);
        CK(hipDeviceSynchronize());
        if (r < 0) continue;
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t0max = 0, t1 = 0;
        for (int w = 0; w < waves; ++w) {
            t0    = std::min(t0, h[w * 2]);
            t0max = std::max(t0max, h[w * 2]);
            t1    = std::max(t1, h[w * 2 + 1]);
        }
        span.push_back((t1 - t0) * 0.01f);
        ramp.push_back((t0max - t0) * 0.01f);
        ev.push_back(ms * 1e3f);
    }
    Stats ss = stats_of(span), sr = stats_of(ramp), se = stats_of(ev);
    printf("%-34s grid %5d x %4d thr x %2d loads: device span med %5.2f min %5.2f p90 %5.2f | ramp med %4.2f | events med %5.2f us | %5.0f GB/s over the span\n",
           name, grid, threads, LOADS, ss.med, ss.mn, ss.p90, sr.med, se.med, bytes / ss.med / 1e3);
}
#endif

int main(int argc, char** argv)
{
    const char* what = argc > 1 ? argv[1] : "all";
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz  L2=%d KiB\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate / 1000, prop.l2CacheSize / 1024);
    const size_t W4K  = 4096ull * 4096;
    const size_t WBIG = 4096ull * 11008;
    const int    NBUF = 40;
    std::vector<uint8_t*> bufs(NBUF), bufs_big(16);
    std::vector<uint8_t>  host(WBIG);
    srand(1);
    for (auto& b : host) b = (uint8_t)(rand() >> 7);
    for (auto& p : bufs) {
        CK(hipMalloc(&p, W4K));
        CK(hipMemcpy(p, host.data(), W4K, hipMemcpyHostToDevice));
    }
    for (auto& p : bufs_big) {
        CK(hipMalloc(&p, WBIG));
        CK(hipMemcpy(p, host.data(), WBIG, hipMemcpyHostToDevice));
    }
    eetq::f16 *x, *scales, *y;
    CK(hipMalloc(&x, 16 * 11008 * 2));
    CK(hipMalloc(&scales, 11008 * 2));
    CK(hipMalloc(&y, 16 * 11008 * 2));
    std::vector<uint16_t> hx(16 * 11008, 0x3800), hs(11008, 0x1c00);  // x = 0.5, s = 2^-8
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0x3000 + (rand() & 0x7ff);
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(scales, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    unsigned* out;
    CK(hipMalloc(&out, 64));

    if (!strcmp(what, "all") || !strcmp(what, "stream")) {
        printf("--- dispatch overhead ---\n");
        bench_overhead(bufs, out);
        printf("--- streaming read floor, 16 MiB per launch, %d rotating buffers ---\n", NBUF);
        bench_stream<4, true>("read nt 1024thr x4", 1024, bufs, W4K, out);
        bench_stream<4, false>("read    1024thr x4", 1024, bufs, W4K, out);
        bench_stream<2, true>("read nt 1024thr x2", 1024, bufs, W4K, out);
        bench_stream<1, true>("read nt 1024thr x1", 1024, bufs, W4K, out);
        bench_stream<4, true>("read nt 512thr x4", 512, bufs, W4K, out);
        bench_stream<8, true>("read nt 512thr x8", 512, bufs, W4K, out);
        bench_stream<4, true>("read nt 256thr x4", 256, bufs, W4K, out);
        bench_stream<8, true>("read nt 256thr x8", 256, bufs, W4K, out);
        bench_stream<16, true>("read nt 256thr x16", 256, bufs, W4K, out);
        bench_stream<16, false>("read    256thr x16", 256, bufs, W4K, out);
        printf("--- streaming read floor, 43 MiB per launch ---\n");
        bench_stream<4, true>("read nt 1024thr x4 (43MiB)", 1024, bufs_big, WBIG, out);
        bench_stream<8, true>("read nt 256thr x8 (43MiB)", 256, bufs_big, WBIG, out);
    }
    if (!strcmp(what, "all") || !strcmp(what, "gemv")) {
        printf("--- GEMV variants ---\n");
        bench_gemv<1, 16, 4, true, true, 1, 4>("M1 exact xreg 16x4 o4", 4096, 4096, bufs, x, scales, y);
        bench_gemv<1, 16, 4, true, true, 1, 8>("M1 exact xreg 16x4 o8", 4096, 4096, bufs, x, scales, y);
        bench_gemv<1, 16, 4, true, false, 1, 4>("M1 exact lds 16x4 o4", 4096, 4096, bufs, x, scales, y);
        bench_gemv<1, 8, 8, true, true, 1, 2>("M1 exact xreg 8x8 o2", 4096, 4096, bufs, x, scales, y);
        bench_gemv<1, 16, 2, false, false, 1, 4>("M1 loop lds 16x2 o4", 4096, 4096, bufs, x, scales, y);
        bench_gemv<2, 16, 4, true, true, 1, 4>("M2 exact xreg 16x4", 4096, 4096, bufs, x, scales, y);
        bench_gemv<2, 16, 4, true, false, 1, 4>("M2 exact lds 16x4", 4096, 4096, bufs, x, scales, y);
        bench_gemv<4, 16, 4, true, false, 2, 4>("M4 exact lds 16x4", 4096, 4096, bufs, x, scales, y);
        printf("--- N=11008 K=4096 ---\n");
        bench_gemv<1, 16, 4, true, true, 1, 4>("M1 exact xreg 16x4 o4", 11008, 4096, bufs_big, x, scales, y);
        bench_gemv<1, 16, 4, true, true, 1, 8>("M1 exact xreg 16x4 o8", 11008, 4096, bufs_big, x, scales, y);
        bench_gemv<1, 8, 8, true, true, 1, 4>("M1 exact xreg 8x8 o4", 11008, 4096, bufs_big, x, scales, y);
        bench_gemv<1, 8, 8, true, true, 1, 2>("M1 exact xreg 8x8 o2", 11008, 4096, bufs_big, x, scales, y);
        printf("--- N=4096 K=11008 ---\n");
        bench_gemv<1, 16, 4, false, false, 2, 4>("M1 loop lds 16x4 o4", 4096, 11008, bufs_big, x, scales, y);
        bench_gemv<1, 16, 4, false, false, 2, 8>("M1 loop lds 16x4 o8", 4096, 11008, bufs_big, x, scales, y);
        bench_gemv<1, 16, 2, false, false, 2, 8>("M1 loop lds 16x2 o8", 4096, 11008, bufs_big, x, scales, y);
        bench_gemv<1, 16, 8, false, false, 2, 4>("M1 loop lds 16x8 o4", 4096, 11008, bufs_big, x, scales, y);
        bench_gemv<4, 16, 4, false, false, 8, 4>("M4 loop lds 16x4 o4", 4096, 11008, bufs_big, x, scales, y);
    }
    if (!strcmp(what, "gaps")) {
        // What does a dependent dispatch cost when CONSECUTIVE launches differ?  Graph chains of 1 000 empty launches (256 workgroups):
        // one kernel repeated / two kernels alternating / one kernel alternating its dynamic LDS size / its workgroup size / its
        // register footprint (a decode step alternates five kernels of different LDS and register needs)
        auto chain = [&](const char* name, std::function<void(int, hipStream_t)> f) {
            printf("%-64s %6.3f us per launch\n", name, time_graph(f, 1000));
        };
        CK(hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)empty_kernel_b, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int pass = 0; pass < 2; ++pass) {
            chain("same kernel, 256 x 256 threads, no LDS", [&](int, hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, out, 0); });
            chain("two kernels alternating", [&](int i, hipStream_t s) {
                if (i & 1) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, out, 0);
                else hipLaunchKernelGGL(empty_kernel_b, dim3(256), dim3(256), 0, s, out, 0);
            });
            chain("same kernel, LDS 0 / 64 KiB alternating", [&](int i, hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), (i & 1) ? 65536 : 0, s, out, 0); });
            chain("same kernel, LDS 64 KiB always", [&](int, hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 65536, s, out, 0); });
            chain("same kernel, 256 / 1024 threads alternating", [&](int i, hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3((i & 1) ? 1024 : 256), 0, s, out, 0); });
            chain("same kernel, 1024 threads always", [&](int, hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(1024), 0, s, out, 0); });
            chain("small / big register footprint alternating", [&](int i, hipStream_t s) {
                if (i & 1) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, out, 0);
                else hipLaunchKernelGGL(fat_kernel, dim3(256), dim3(256), 0, s, out, 0);
            });
            chain("big register footprint always", [&](int, hipStream_t s) { hipLaunchKernelGGL(fat_kernel, dim3(256), dim3(256), 0, s, out, 0); });
            chain("five kernels of different LDS / threads / registers", [&](int i, hipStream_t s) {
                switch (i % 5) {
                case 0: hipLaunchKernelGGL(empty_kernel, dim3(960), dim3(512), 20480, s, out, 0); break;
                case 1: hipLaunchKernelGGL(fat_kernel, dim3(200), dim3(256), 0, s, out, 0); break;
                case 2: hipLaunchKernelGGL(empty_kernel_b, dim3(768), dim3(512), 10240, s, out, 0); break;
                case 3: hipLaunchKernelGGL(empty_kernel, dim3(1728), dim3(512), 10240, s, out, 0); break;
                default: hipLaunchKernelGGL(empty_kernel_b, dim3(768), dim3(512), 27648, s, out, 0); break;
                }
            });
            chain("... the same five shapes, one kernel, one LDS size", [&](int i, hipStream_t s) {
                const int grids[5] = {960, 200, 768, 1728, 768};
                hipLaunchKernelGGL(empty_kernel, dim3(grids[i % 5]), dim3(512), 27648, s, out, 0);
            });
        }
        return 0;
    }
    if (!strcmp(what, "gemvladder")) {
        // one ablation ladder from the load-only kernel to the shipping GEMV, every rung chain-timed (one graph of 1200 dependent
        // launches over the 40 rotating weight sets) and dispatch-timed on the same box, three passes in alternating order
        const int  N = 4096, K = 4096, ITERS = 1200;
        const char* names[13] = {"0 weight loads only", "1 + scale / x loads first", "2 + dequant, dot2", "3 + wave butterflies",
                                 "4 + LDS, barrier, wave-0 sum", "5 + 32-byte store (= GEMV)", "6 = 5, store sc0 sc1",
                                 "7 = 5, sum by 4 waves", "8 = 5, x staged per wave in LDS", "9 = 1, scale load only",
                                 "10 = 8, 32-lane x load", "11 = 5, x by DPP row broadcast", "12 = 11, x load pinned first"};
        constexpr int NL = 13;
        {   // level 11 must give level 5's bits
            std::vector<uint16_t> y5(N), y11(N);
            hipLaunchKernelGGL(gemv_ladder_kernel<5>, dim3(N / 16), dim3(1024), 0, 0, x, (const uint8_t*)bufs[3], scales, y, N, K, out);
            CK(hipMemcpy(y5.data(), y, N * 2, hipMemcpyDeviceToHost));
            CK(hipMemset(y, 0, N * 2));
            hipLaunchKernelGGL(gemv_ladder_kernel<11>, dim3(N / 16), dim3(1024), 0, 0, x, (const uint8_t*)bufs[3], scales, y, N, K, out);
            CK(hipMemcpy(y11.data(), y, N * 2, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int i = 0; i < N; ++i) bad += y5[i] != y11[i];
            printf("level 11 vs level 5: %d of %d outputs differ (y[0] = 0x%04x / 0x%04x)\n", bad, N, y5[0], y11[0]);
        }
        double chain[3][NL + 1], disp[3][NL + 1], plain_chain[3];
        auto run = [&](auto kern, int pass, int idx) {
            chain[pass][idx] = time_graph(
                [&](int i, hipStream_t s) {
                    hipLaunchKernelGGL(kern, dim3(N / 16), dim3(1024), 0, s, x, (const uint8_t*)bufs[i % bufs.size()], scales, y, N, K, out);
                },
                ITERS);
            auto st = time_dispatch(
                [&](int i, hipEvent_t a, hipEvent_t b) {
                    hipExtLaunchKernelGGL(kern, dim3(N / 16), dim3(1024), 0, 0, a, b, 0, x, (const uint8_t*)bufs[i % bufs.size()], scales, y,
                                          N, K, out);
                },
                400);
            disp[pass][idx] = st.med;
        };
        for (int pass = 0; pass < 3; ++pass) {
            auto body = [&](int l) {
                switch (l) {
                case 0: run(gemv_ladder_kernel<0>, pass, 0); break;
                case 1: run(gemv_ladder_kernel<1>, pass, 1); break;
                case 2: run(gemv_ladder_kernel<2>, pass, 2); break;
                case 3: run(gemv_ladder_kernel<3>, pass, 3); break;
                case 4: run(gemv_ladder_kernel<4>, pass, 4); break;
                case 5: run(gemv_ladder_kernel<5>, pass, 5); break;
                case 6: run(gemv_ladder_kernel<6>, pass, 6); break;
                case 7: run(gemv_ladder_kernel<7>, pass, 7); break;
                case 8: run(gemv_ladder_kernel<8>, pass, 8); break;
                case 9: run(gemv_ladder_kernel<9>, pass, 9); break;
                case 10: run(gemv_ladder_kernel<10>, pass, 10); break;
                case 11: run(gemv_ladder_kernel<11>, pass, 11); break;
                case 12: run(gemv_ladder_kernel<12>, pass, 12); break;
                }
            };
            if (pass & 1) for (int l = NL - 1; l >= 0; --l) body(l);
            else for (int l = 0; l < NL; ++l) body(l);
            // the library kernel itself, same harness
            auto gk = eetq::gemv::gemv_kernel<1, 16, 4, true, true, 1, 8>;
            const unsigned gsm = (unsigned)eetq::gemv::gemv_smem_bytes(1, K, 16, true);
            chain[pass][NL] = time_graph(
                [&](int i, hipStream_t s) {
                    hipLaunchKernelGGL(gk, dim3(N / 16), dim3(1024), gsm, s, x, (const uint8_t*)bufs[i % bufs.size()], scales, y, N, K,
                                       (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, 0, 0.f);
                },
                ITERS);
            auto st = time_dispatch(
                [&](int i, hipEvent_t a, hipEvent_t b) {
                    hipExtLaunchKernelGGL(gk, dim3(N / 16), dim3(1024), gsm, 0, a, b, 0, x, (const uint8_t*)bufs[i % bufs.size()], scales, y,
                                          N, K, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, 0, 0.f);
                },
                400);
            disp[pass][NL] = st.med;
            auto gp = eetq::gemv::gemv_kernel<1, 16, 4, true, true, 1, 8, 0, 8, true>;
            plain_chain[pass] = time_graph(
                [&](int i, hipStream_t s) {
                    hipLaunchKernelGGL(gp, dim3(N / 16), dim3(1024), gsm, s, x, (const uint8_t*)bufs[i % bufs.size()], scales, y, N, K,
                                       (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, 0, 0.f);
                },
                ITERS);
        }
        printf("--- GEMV ladder, M=1 N=K=4096, 256 x 1024 threads, 16 waves x 4 tiles; chain = us per step of a %d-launch graph (best of 5"
               " replays), disp = median dispatch begin->end; three passes (up, down, up) ---\n", ITERS);
        for (int l = 0; l <= NL; ++l) {
            const double c = std::min(chain[0][l], std::min(chain[1][l], chain[2][l]));
            const double p = l && l < 6 ? std::min(chain[0][l - 1], std::min(chain[1][l - 1], chain[2][l - 1])) : c;
            printf("%-30s chain %5.2f %5.2f %5.2f  best %5.2f (%+5.2f vs previous rung) | disp med %5.2f %5.2f %5.2f\n",
                   l < NL ? names[l] : "library gemv_kernel", chain[0][l], chain[1][l], chain[2][l], c, l && l < 6 ? c - p : 0.0, disp[0][l],
                   disp[1][l], disp[2][l]);
        }
        printf("%-30s chain %5.2f %5.2f %5.2f  (no run-time epilogue: what a plain projection launches)\n", "library gemv_kernel, PLAIN", plain_chain[0],
               plain_chain[1], plain_chain[2]);
        return 0;
    }
    if (!strcmp(what, "mall")) {
        // Is a weight stream served from the 256 MB Infinity Cache faster than from HBM?  A load-only kernel over a 128 MiB region:
        // the same region every launch (cache-resident after the first) vs four rotating regions (512 MiB), nt and plain loads.
        const size_t REG = 128ull << 20;
        uint8_t*     big;
        CK(hipMalloc(&big, 4 * REG));
        CK(hipMemset(big, 1, 4 * REG));
        auto run = [&](const char* name, auto kern, int nreg, int threads, int loads) {
            const int grid = (int)(REG / 16 / threads / loads);
            double    g    = time_graph(
                [&](int i, hipStream_t s) {
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, s, (const u32x4*)(big + (size_t)(i % nreg) * REG), out);
                },
                200);
            printf("%-44s regions %d | %7.2f us per 128 MiB launch -> %6.0f GB/s\n", name, nreg, g, (double)REG / g / 1e3);
        };
        // ... and from the per-XCD L2s (32 MiB in all)?  The same 16 MiB / 24 MiB every launch: does L2 content survive a kernel boundary?
        auto run_small = [&](const char* name, auto kern, size_t bytes, int nreg, int threads, int loads) {
            const int grid = (int)(bytes / 16 / threads / loads);
            double    g    = time_graph(
                [&](int i, hipStream_t s) {
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, s, (const u32x4*)(big + (size_t)(i % nreg) * REG), out);
                },
                400);
            printf("%-44s %2zu MiB, regions %d | %7.2f us per launch -> %6.0f GB/s\n", name, bytes >> 20, nreg, g, (double)bytes / g / 1e3);
        };
        for (int rep = 0; rep < 2; ++rep) {
            run_small("plain loads, 1024 thr x 4", stream_read_kernel<4, false>, 16ull << 20, 1, 1024, 4);
            run_small("plain loads, 1024 thr x 4", stream_read_kernel<4, false>, 16ull << 20, 4, 1024, 4);
            run_small("nt loads, 1024 thr x 4", stream_read_kernel<4, true>, 16ull << 20, 1, 1024, 4);
            run_small("plain loads, 1024 thr x 4", stream_read_kernel<4, false>, 24ull << 20, 1, 1024, 4);
            run_small("plain loads, 1024 thr x 4", stream_read_kernel<4, false>, 24ull << 20, 4, 1024, 4);
            run_small("plain loads, 1024 thr x 4", stream_read_kernel<4, false>, 8ull << 20, 1, 1024, 4);
            run_small("plain loads, 1024 thr x 4", stream_read_kernel<4, false>, 8ull << 20, 4, 1024, 4);
        }
        // ... and can a CHEAP kernel put them there for the next one?  Pairs (touch one dword per sector / line of region i, then read
        // region i with the GEMV's load pattern, nt or plain), four rotating 16 / 24 MiB regions; the touch kernel alone for its cost
        auto run_pair = [&](const char* name, auto touch, auto kern, size_t bytes, bool with_read) {
            const int grid = (int)(bytes / 16 / 1024 / 4);
            double    g    = time_graph(
                [&](int i, hipStream_t s) {
                    const uint8_t* reg = big + (size_t)(i % 4) * REG;
                    hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, s, reg, bytes, out);
                    if (with_read) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, s, (const u32x4*)reg, out);
                },
                400);
            printf("%-62s %2zu MiB | %7.2f us per %s\n", name, bytes >> 20, g, with_read ? "pair" : "touch launch");
        };
        for (size_t mb : {16, 24}) {
            run_pair("touch 1 dword / 64 B alone", touch_lines_kernel<64>, stream_read_kernel<4, true>, mb << 20, false);
            run_pair("touch 1 dword / 128 B alone", touch_lines_kernel<128>, stream_read_kernel<4, true>, mb << 20, false);
            run_pair("touch / 64 B, then nt read", touch_lines_kernel<64>, stream_read_kernel<4, true>, mb << 20, true);
            run_pair("touch / 64 B, then plain read", touch_lines_kernel<64>, stream_read_kernel<4, false>, mb << 20, true);
            run_pair("touch / 128 B, then nt read", touch_lines_kernel<128>, stream_read_kernel<4, true>, mb << 20, true);
            run_pair("touch / 128 B, then plain read", touch_lines_kernel<128>, stream_read_kernel<4, false>, mb << 20, true);
        }
        for (int rep = 0; rep < 2; ++rep) {
            run("nt loads, 1024 thr x 4", stream_read_kernel<4, true>, 4, 1024, 4);
            run("nt loads, 1024 thr x 4", stream_read_kernel<4, true>, 1, 1024, 4);
            run("plain loads, 1024 thr x 4", stream_read_kernel<4, false>, 4, 1024, 4);
            run("plain loads, 1024 thr x 4", stream_read_kernel<4, false>, 1, 1024, 4);
            run("plain loads, 256 thr x 16", stream_read_kernel<16, false>, 4, 256, 16);
            run("plain loads, 256 thr x 16", stream_read_kernel<16, false>, 1, 256, 16);
        }
        return 0;
    }
    if (!strcmp(what, "gemv_geom")) {  // wave count x tiles-in-flight at equal bytes in flight (64 KiB per workgroup)
        bench_gemv<1, 16, 4, true, true, 1, 8>("M1 xreg 16 waves x 4 (shipping)", 4096, 4096, bufs, x, scales, y);
        bench_gemv<1, 8, 8, true, true, 1, 2>("M1 xreg  8 waves x 8", 4096, 4096, bufs, x, scales, y);
        bench_gemv<1, 4, 16, true, true, 1, 1>("M1 xreg  4 waves x 16", 4096, 4096, bufs, x, scales, y);
        bench_gemv<1, 16, 4, true, true, 1, 8>("M1 xreg 16 waves x 4 (again)", 4096, 4096, bufs, x, scales, y);
        bench_gemv<1, 16, 4, true, false, 1, 4>("M1 lds  16 waves x 4", 4096, 4096, bufs, x, scales, y);
    }
    if (!strcmp(what, "all") || !strcmp(what, "gemm")) {
        printf("--- MFMA dequant-GEMM (128 x 128 x 64 tile, 4 waves) ---\n");
        eetq::f16 *xg, *yg;
        CK(hipMalloc(&xg, 8192ull * 4096 * 2));
        CK(hipMalloc(&yg, 8192ull * 11008 * 2));
        {
            std::vector<uint16_t> h(8192ull * 4096);
            for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));  // +-[0.125, 0.5)
            CK(hipMemcpy(xg, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        for (int rep = 0; rep < 3; ++rep) bench_gemm<0>("gemm", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<0>("gemm", 4096, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<0>("gemm", 8192, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<0>("gemm", 1024, 11008, 4096, bufs_big, xg, scales, yg);
        bench_gemm<0>("gemm", 256, 4096, 4096, bufs, xg, scales, yg);
        printf("ablations (numerically wrong on purpose; they locate the time)\n");
        bench_gemm<1>("gemm -dma", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<2>("gemm -dequant", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<4>("gemm -ldsread", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<8>("gemm -mfma", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<6>("gemm mfma+dma", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<7>("gemm mfma+barrier", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<23>("gemm mfma only", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<14>("gemm dma+barrier only", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<30>("gemm dma only", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<9>("gemm ldsread+dequant", 1024, 4096, 4096, bufs, xg, scales, yg);
    }
    if (!strcmp(what, "streamk13b")) {  // batched decode at Llama-13B shapes: tile rows per workgroup
        uint8_t* huge;
        CK(hipMalloc(&huge, 13824ull * 5120 * 8));
        CK(hipMemset(huge, 0x5a, 13824ull * 5120 * 8));
        std::vector<uint8_t*> b70;
        for (int i = 0; i < 8; ++i) b70.push_back(huge + (size_t)i * 13824 * 5120);
        eetq::f16 *xs, *ys;
        CK(hipMalloc(&xs, 16ull * 13824 * 2));
        CK(hipMalloc(&ys, 16ull * 27648 * 2));
        CK(hipMemset(xs, 0x30, 16ull * 13824 * 2));
        for (int M : {4, 8, 16}) {
            printf("M=%d\n", M);
            bench_streamk<1, 1, 16, 2, 4>("NT1 16x2 N=5120 K=5120", M, 5120, 5120, b70, xs, scales, ys);
            bench_streamk<1, 2, 16, 2, 4>("NT2 16x2 N=5120 K=5120", M, 5120, 5120, b70, xs, scales, ys);
            bench_streamk<1, 1, 16, 2, 4>("NT1 16x2 N=5120 K=13824", M, 5120, 13824, b70, xs, scales, ys);
            bench_streamk<1, 2, 16, 2, 4>("NT2 16x2 N=5120 K=13824", M, 5120, 13824, b70, xs, scales, ys);
            bench_streamk<1, 1, 16, 4, 4>("NT1 16x4 N=5120 K=13824", M, 5120, 13824, b70, xs, scales, ys);
            bench_streamk<1, 2, 16, 2, 4>("NT2 16x2 N=13824 K=5120", M, 13824, 5120, b70, xs, scales, ys);
            bench_streamk<1, 1, 16, 2, 4>("NT1 16x2 N=4096 K=4096", M, 4096, 4096, bufs, xs, scales, ys);
            bench_streamk<1, 1, 16, 2, 4>("NT1 16x2 N=4096 K=11008", M, 4096, 11008, bufs_big, xs, scales, ys);
            bench_streamk<1, 2, 16, 2, 4>("NT2 16x2 N=4096 K=11008", M, 4096, 11008, bufs_big, xs, scales, ys);
        }
    }
    if (!strcmp(what, "gemv13b")) {  // Llama-13B shapes: workgroup-count quantisation and kernel variants
        uint8_t* huge;
        CK(hipMalloc(&huge, 13824ull * 5120 * 8));
        CK(hipMemset(huge, 0x5a, 13824ull * 5120 * 8));
        std::vector<uint8_t*> b70;
        for (int i = 0; i < 8; ++i) b70.push_back(huge + (size_t)i * 13824 * 5120);
        eetq::f16* xl;
        CK(hipMalloc(&xl, 4 * 13824 * 2));
        CK(hipMemset(xl, 0x30, 4 * 13824 * 2));
        printf("--- K=5120, N sweep (generic loop kernel 16x2 o8) ---\n");
        for (int N : {4096, 5120, 6144, 8192, 13824})
            bench_gemv<1, 16, 2, false, false, 2, 8>("loop lds 16x2 o8", N, 5120, b70, xl, scales, y);
        printf("--- N=5120 K=5120 variants ---\n");
        bench_gemv<1, 16, 4, false, false, 2, 8>("loop lds 16x4 o8", 5120, 5120, b70, xl, scales, y);
        bench_gemv<1, 16, 5, true, false, 2, 8>("exact lds 16x5 o8", 5120, 5120, b70, xl, scales, y);
        bench_gemv<1, 16, 5, true, true, 1, 8>("exact xreg 16x5 o8", 5120, 5120, b70, xl, scales, y);
        bench_gemv<1, 16, 5, true, true, 1, 4>("exact xreg 16x5 o4", 5120, 5120, b70, xl, scales, y);
        bench_gemv<1, 8, 5, false, false, 2, 8>("loop lds 8x5 o8", 5120, 5120, b70, xl, scales, y);
        bench_gemv<1, 8, 10, true, true, 1, 4>("exact xreg 8x10 o4", 5120, 5120, b70, xl, scales, y);
        printf("--- N=5120 K=13824 variants ---\n");
        bench_gemv<1, 16, 2, false, false, 4, 8>("loop lds 16x2 o8", 5120, 13824, b70, xl, scales, y);
        bench_gemv<1, 16, 4, false, false, 4, 8>("loop lds 16x4 o8", 5120, 13824, b70, xl, scales, y);
        bench_gemv<1, 16, 4, false, false, 4, 4>("loop lds 16x4 o4", 5120, 13824, b70, xl, scales, y);
        bench_gemv<1, 8, 4, false, false, 4, 8>("loop lds 8x4 o8", 5120, 13824, b70, xl, scales, y);
        bench_gemv<1, 8, 8, false, false, 4, 4>("loop lds 8x8 o4", 5120, 13824, b70, xl, scales, y);
        printf("--- 8-column units (gemv_half_kernel) ---\n");
        {
            auto bench_half = [&](const char* name, auto kern, int N, int K) {
                const size_t smem = eetq::gemv::gemv_half_smem_bytes(K, 8);
                const double bytes = (double)K * N + 2.0 * K + 2.0 * N + 2.0 * N;
                auto st = time_dispatch(
                    [&](int i, hipEvent_t a, hipEvent_t b) {
                        hipExtLaunchKernelGGL(kern, dim3(N / 8), dim3(512), (unsigned)smem, 0, a, b, 0, xl,
                                              (const uint8_t*)b70[i % b70.size()], scales, y, N, K, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, 0, 0.f);
                    },
                    400);
                printf("%-30s N=%5d K=%5d M=1 | disp mean %6.2f med %6.2f min %6.2f us -> %6.0f GB/s(med)\n", name, N, K,
                       st.mean, st.med, st.mn, bytes / st.med / 1e3);
            };
            bench_half("half 8x2 o8 xv2", eetq::gemv::gemv_half_kernel<8, 2, 2, 8>, 5120, 5120);
            bench_half("half 8x4 o8 xv2", eetq::gemv::gemv_half_kernel<8, 4, 2, 8>, 5120, 5120);
            bench_half("half 8x2 o4 xv2", eetq::gemv::gemv_half_kernel<8, 2, 2, 4>, 5120, 5120);
            bench_half("half 8x2 o8 xv4", eetq::gemv::gemv_half_kernel<8, 2, 4, 8>, 5120, 13824);
            bench_half("half 8x4 o8 xv4", eetq::gemv::gemv_half_kernel<8, 4, 4, 8>, 5120, 13824);
            bench_half("half 8x2 o8 xv2", eetq::gemv::gemv_half_kernel<8, 2, 2, 8>, 6144, 5120);
            bench_half("half 8x2 o8 xv2", eetq::gemv::gemv_half_kernel<8, 2, 2, 8>, 13824, 5120);
            bench_half("half 8x2 o8 xv2", eetq::gemv::gemv_half_kernel<8, 2, 2, 8>, 4096, 5120);
        }
        printf("--- N=13824 K=5120 ---\n");
        bench_gemv<1, 16, 2, false, false, 2, 8>("loop lds 16x2 o8", 13824, 5120, b70, xl, scales, y);
        bench_gemv<1, 16, 5, true, true, 1, 8>("exact xreg 16x5 o8", 13824, 5120, b70, xl, scales, y);
    }
    if (!strcmp(what, "tiles")) {  // 128 x 128 (J = 2) vs 128 x 64 (J = 1) tiles
        eetq::f16 *xg, *yg;
        CK(hipMalloc(&xg, 4096ull * 13824 * 2));
        CK(hipMalloc(&yg, 4096ull * 13824 * 2));
        {
            std::vector<uint16_t> h(4096ull * 13824);
            for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));
            CK(hipMemcpy(xg, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        uint8_t* huge;
        CK(hipMalloc(&huge, 13824ull * 5120 * 6));
        CK(hipMemset(huge, 0x5a, 13824ull * 5120 * 6));
        std::vector<uint8_t*> b70;
        for (int i = 0; i < 6; ++i) b70.push_back(huge + (size_t)i * 13824 * 5120);
        for (int M : {64, 96, 128}) {
            bench_gemm<0, 1>("J=1 128x64", M, 4096, 4096, bufs, xg, scales, yg);
            bench_gemm<0, 1>("J=1 128x64 N=11008", M, 11008, 4096, bufs_big, xg, scales, yg);
            bench_gemm<0, 1>("J=1 128x64 K=11008", M, 4096, 11008, bufs_big, xg, scales, yg);
        }
        for (int M : {192, 256, 384, 512, 768, 1024}) {
            bench_gemm<0, 2>("J=2 128x128", M, 4096, 4096, bufs, xg, scales, yg);
            bench_gemm<0, 1>("J=1 128x64", M, 4096, 4096, bufs, xg, scales, yg);
        }
        bench_gemm<0, 2>("J=2 128x128", 1024, 5120, 5120, b70, xg, scales, yg);
        bench_gemm<0, 1>("J=1 128x64", 1024, 5120, 5120, b70, xg, scales, yg);
        bench_gemm<0, 2>("J=2 128x128", 1024, 5120, 13824, b70, xg, scales, yg);
        bench_gemm<0, 1>("J=1 128x64", 1024, 5120, 13824, b70, xg, scales, yg);
        bench_gemm<0, 2>("J=2 128x128", 1024, 13824, 5120, b70, xg, scales, yg);
        bench_gemm<0, 1>("J=1 128x64", 1024, 13824, 5120, b70, xg, scales, yg);
        bench_gemm<0, 2>("J=2 128x128", 1024, 11008, 4096, bufs_big, xg, scales, yg);
        bench_gemm<0, 1>("J=1 128x64", 1024, 11008, 4096, bufs_big, xg, scales, yg);
        bench_gemm<0, 2>("J=2 128x128", 2048, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<0, 1>("J=1 128x64", 2048, 4096, 4096, bufs, xg, scales, yg);
    }
    if (!strcmp(what, "overhead1")) {  // plain launches for rocprofv3 --kernel-trace: empty / read-only / GEMV
        for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, out, 0);
        CK(hipDeviceSynchronize());
        for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(1024), 0, 0, out, 0);
        CK(hipDeviceSynchronize());
        for (int i = 0; i < 300; ++i)
            hipLaunchKernelGGL((stream_read_kernel<4, true>), dim3(256), dim3(1024), 0, 0, (const u32x4*)bufs[i % bufs.size()], out);
        CK(hipDeviceSynchronize());
        auto gk = eetq::gemv::gemv_kernel<1, 16, 4, true, true, 1, 8>;
        for (int i = 0; i < 300; ++i)
            hipLaunchKernelGGL(gk, dim3(256), dim3(1024), (unsigned)eetq::gemv::gemv_smem_bytes(1, 4096, 16, true), 0, x,
                               (const uint8_t*)bufs[i % bufs.size()], scales, y, 4096, 4096, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, 0, 0.f);
        CK(hipDeviceSynchronize());
    }
    if (!strcmp(what, "gemmcu")) {
        // 34.36 GFLOP every time; 256 / 128 / 64 workgroups of the 128 x 128 tile (one per CU), the K loop 1x / 2x / 4x as long.
        // If idle CUs came back as clock, the full kernel on 128 CUs would approach its 256-CU time.
        printf("--- same FLOPs on fewer CUs: 128 x 128 tile, full kernel and MFMA stream alone, random vs zero operands ---\n");
        const size_t WB = 4096ull * 8192;
        setvbuf(stdout, nullptr, _IOLBF, 0);
        eetq::f16 *xg, *yg, *xz;
        CK(hipMalloc(&xg, 1024ull * 16384 * 2));
        CK(hipMalloc(&xz, 1024ull * 16384 * 2));
        CK(hipMalloc(&yg, 1024ull * 4096 * 2));
        CK(hipMemset(xz, 0, 1024ull * 16384 * 2));
        {
            std::vector<uint16_t> h(1024ull * 16384);
            for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));  // +-[0.125, 0.5)
            CK(hipMemcpy(xg, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        std::vector<uint8_t*> zb(4);
        for (auto& p : zb) {
            CK(hipMalloc(&p, WB));
            CK(hipMemset(p, 0x80, WB));  // q = 0
        }
        struct Shape { int M, N, K; };
        for (Shape sh : {Shape{1024, 4096, 4096}, Shape{1024, 2048, 8192}, Shape{512, 4096, 8192}, Shape{1024, 1024, 16384}}) {
            for (int rep = 0; rep < 2; ++rep) {
                bench_gemm_cu<0>("full kernel, random operands", sh.M, sh.N, sh.K, bufs_big, xg, scales, yg);
                bench_gemm_cu<0>("full kernel, zero operands", sh.M, sh.N, sh.K, zb, xz, scales, yg);
                bench_gemm_cu<23>("MFMA stream only, random regs", sh.M, sh.N, sh.K, bufs_big, xg, scales, yg);
            }
        }
        bench_gemm_cu<0, 1>("128 x 64 tile, random operands", 1024, 4096, 4096, bufs_big, xg, scales, yg);
        bench_gemm_cu<0, 1>("128 x 64 tile, random operands", 1024, 2048, 8192, bufs_big, xg, scales, yg);
        return 0;
    }
#ifdef EETQ_KBENCH_STAMPS
    if (!strcmp(what, "geometry")) {
        unsigned long long* st;
        CK(hipMalloc(&st, (size_t)65536 * 2 * 8));
        printf("--- 16 MiB load-only stream, launch geometry vs device span (100 MHz device clock) ---\n");
        bench_geometry<4, true>("256 WG x 16 waves x 4", 1024, bufs, W4K, out, st);
        bench_geometry<8, true>("256 WG x 8 waves x 8", 512, bufs, W4K, out, st);
        bench_geometry<16, true>("256 WG x 4 waves x 16", 256, bufs, W4K, out, st);
        bench_geometry<8, true>("512 WG x 4 waves x 8", 256, bufs, W4K, out, st);
        bench_geometry<4, true>("1024 WG x 4 waves x 4", 256, bufs, W4K, out, st);
        bench_geometry<4, true>("512 WG x 8 waves x 4", 512, bufs, W4K, out, st);
        bench_geometry<2, true>("512 WG x 16 waves x 2", 1024, bufs, W4K, out, st);
        bench_geometry<16, true>("1024 WG x 1 wave x 16", 64, bufs, W4K, out, st);
        bench_geometry<4, false>("256 WG x 16 waves x 4 (no nt)", 1024, bufs, W4K, out, st);
        bench_geometry<16, false>("256 WG x 4 waves x 16 (no nt)", 256, bufs, W4K, out, st);
        printf("--- 43 MiB ---\n");
        bench_geometry<4, true>("688 WG x 16 waves x 4", 1024, bufs_big, WBIG, out, st);
        bench_geometry<16, true>("688 WG x 4 waves x 16", 256, bufs_big, WBIG, out, st);
    }
#endif
#ifdef EETQ_KBENCH_STAMPS

    if (!strcmp(what, "gemmstamps")) {
        // Where do the tile GEMM's microseconds go at M = 1024, N = K = 4096 (one tile per workgroup, 256 workgroups)?
        // Device-clock stamps of wave 0 of every workgroup: 0 entry, 1 first stage landed, 2 steady loop done, 3 drain steps
        // done, 4 second K half parked in LDS, 5 output stored.
        using namespace eetq::gemm;
        const int M = 1024, N = 4096, K = 4096, NWG = (M / 128) * (N / 128), ROUNDS = 40;
        eetq::f16 *xg, *yg;
        CK(hipMalloc(&xg, (size_t)M * K * 2));
        CK(hipMalloc(&yg, (size_t)M * N * 2));
        {
            std::vector<uint16_t> hx((size_t)M * K);
            for (auto& v : hx) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));  // +-[0.125, 0.5)
            CK(hipMemcpy(xg, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        }
        auto kern = gemm_tile_kernel<0, 2>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TileCfg<2>::SMEM_BYTES));
        unsigned long long* stamps;
        CK(hipMalloc(&stamps, (size_t)NWG * 8 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(eetq::gemm::g_gemm_stamps), &stamps, sizeof(stamps)));
        std::vector<unsigned long long> h((size_t)NWG * 8);
        double sum[6] = {0, 0, 0, 0, 0, 0}, mx[6] = {0, 0, 0, 0, 0, 0}, span = 0;
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        double disp = 0;
        for (int r = -5; r < ROUNDS; ++r) {
            hipExtLaunchKernelGGL(kern, dim3(NWG), dim3(256), TileCfg<2>::SMEM_BYTES, 0, a, b, 0, xg,
                                  (const uint8_t*)bufs[(r + 5) % bufs.size()], scales, yg, M, N, K, N, eetq::Epilogue{});
            CK(hipDeviceSynchronize());
            if (r < 0) continue;
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            disp += ms * 1e3;
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, tend = 0;
            for (int w = 0; w < NWG; ++w) t0 = std::min(t0, h[w * 8]);
            for (int i = 0; i < 6; ++i) {
                double s = 0, m = 0;
                for (int w = 0; w < NWG; ++w) {
                    const double v = (double)(h[w * 8 + i] - t0) / 100.0;
                    s += v;
                    m = std::max(m, v);
                }
                sum[i] += s / NWG;
                mx[i] += m;
            }
            for (int w = 0; w < NWG; ++w) tend = std::max(tend, h[w * 8 + 5]);
            span += (double)(tend - t0) / 100.0;
        }
        const char* names[6] = {"entry", "first stage landed", "steady loop done", "drain steps done", "K halves parked in LDS",
                                "output stored"};
        printf("tile GEMM M=%d N=%d K=%d, %d workgroups, %d launches: dispatch begin->end %.2f us, device span %.2f us\n", M, N, K,
               NWG, ROUNDS, disp / ROUNDS, span / ROUNDS);
        for (int i = 0; i < 6; ++i)
            printf("  %-24s mean %6.2f us  latest workgroup %6.2f us   (+%.2f since the previous stamp)\n", names[i], sum[i] / ROUNDS,
                   mx[i] / ROUNDS, i ? (sum[i] - sum[i - 1]) / ROUNDS : 0.0);
        return 0;
    }
    if (!strcmp(what, "decompose")) {
        // Where do the GEMV's microseconds go?  Same process, same buffers, interleaved rounds of
        //   E  empty kernel, GEMV launch geometry (256 x 1024)      -> what a dispatch costs with no work
        //   R  load-only kernel, 16 MiB (the read floor)
        //   G  the shipping GEMV (M = 1, N = K = 4096) with device-clock stamps from every workgroup
        // each with begin/end dispatch timestamps (the quantity rocprofv3 --kernel-trace reports; run this mode under
        // rocprofv3 --kernel-trace to get the tool's own numbers for the same dispatches).  Device clock: 100 MHz
        // s_memrealtime, one tick = 10 ns.
        const int NWG = 256, ROUNDS = 400;
        unsigned long long* stamps;
        CK(hipMalloc(&stamps, (size_t)NWG * 2 * 4 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(eetq::gemv::g_gemv_stamps), &stamps, sizeof(stamps)));
        auto gk = eetq::gemv::gemv_kernel<1, 16, 4, true, true, 1, 8>;
        const unsigned gsm = (unsigned)eetq::gemv::gemv_smem_bytes(1, 4096, 16, true);
        std::vector<float> dE, dR, dG, span, ramp, wgdur, tail, first_done, last_start;
        std::vector<unsigned long long> h((size_t)NWG * 8);
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        auto timed = [&](auto&& launch) {
            launch(a, b);
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            return ms * 1e3f;
        };
        for (int r = -20; r < ROUNDS; ++r) {
            const uint8_t* wbuf = bufs[(r + 20) % bufs.size()];
            float e = timed([&](hipEvent_t s0, hipEvent_t s1) {
                hipExtLaunchKernelGGL(empty_kernel, dim3(256), dim3(1024), 0, 0, s0, s1, 0, out, 0);
            });
            float rd = timed([&](hipEvent_t s0, hipEvent_t s1) {
                hipExtLaunchKernelGGL((stream_read_kernel<4, true>), dim3(256), dim3(1024), 0, 0, s0, s1, 0, (const u32x4*)wbuf, out);
            });
            const uint8_t* wbuf2 = bufs[(r + 33) % bufs.size()];
            float g = timed([&](hipEvent_t s0, hipEvent_t s1) {
                hipExtLaunchKernelGGL(gk, dim3(256), dim3(1024), gsm, 0, s0, s1, 0, x, wbuf2, scales, y, 4096, 4096, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, 0, 0.f);
            });
            if (r < 0) continue;
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t0max = 0, t1min = ~0ull, t2 = 0;
            double             wsum = 0;
            for (int w = 0; w < NWG; ++w) {
                const unsigned long long s0 = std::min(h[(w * 2) * 4], h[(w * 2 + 1) * 4]);
                const unsigned long long e2 = h[(w * 2) * 4 + 2];  // wave 0 writes y last
                const unsigned long long e1 = std::max(h[(w * 2) * 4 + 1], h[(w * 2 + 1) * 4 + 1]);
                t0    = std::min(t0, s0);
                t0max = std::max(t0max, s0);
                t1min = std::min(t1min, e1);
                t2    = std::max(t2, e2);
                wsum += (double)(e2 - s0);
            }
            dE.push_back(e);
            dR.push_back(rd);
            dG.push_back(g);
            span.push_back((t2 - t0) * 0.01f);
            ramp.push_back((t0max - t0) * 0.01f);
            wgdur.push_back((float)(wsum / NWG) * 0.01f);
            first_done.push_back((t1min - t0) * 0.01f);
        }
        auto pr = [&](const char* name, std::vector<float>& v) {
            Stats st = stats_of(v);
            printf("%-58s mean %6.2f  med %6.2f  min %6.2f  p10 %6.2f  p90 %6.2f us\n", name, st.mean, st.med, st.mn, st.p10, st.p90);
        };
        printf("--- GEMV M=1 N=K=4096 decomposition, %d interleaved rounds, un-profiled dispatch timestamps ---\n", ROUNDS);
        pr("E  empty kernel 256x1024: dispatch begin->end", dE);
        pr("R  load-only 16 MiB: dispatch begin->end", dR);
        pr("G  GEMV: dispatch begin->end", dG);
        pr("G  device span: first wave in -> last store out", span);
        pr("G  launch ramp: first workgroup in -> last workgroup in", ramp);
        pr("G  first workgroup's data complete (since first wave in)", first_done);
        pr("G  mean workgroup residency (entry -> final store)", wgdur);
        std::vector<float> cp(dG.size());
        for (size_t i = 0; i < dG.size(); ++i) cp[i] = dG[i] - span[i];
        pr("G  dispatch begin->end MINUS device span (CP / fences)", cp);
        Stats sg = stats_of(dG), ss = stats_of(span), sr = stats_of(dR);
        printf("bytes/launch 16801792: begin->end %.0f GB/s (%.3f of 8 TB/s), device span %.0f GB/s (%.3f), load-only floor %.0f GB/s;"
               " GEMV / floor = %.3f\n",
               16801792.0 / sg.mean / 1e3, 16801792.0 / sg.mean / 1e3 / 8000, 16801792.0 / ss.mean / 1e3,
               16801792.0 / ss.mean / 1e3 / 8000, 16777216.0 / sr.mean / 1e3, sg.mean / sr.mean);
    }
#endif
    if (!strcmp(what, "gemm8")) {  // the 128 x 128 tile on 4 waves (J = 2, one per SIMD) vs 8 waves (J = 1, two per SIMD)
        eetq::f16 *xg, *yg;
        CK(hipMalloc(&xg, 8192ull * 4096 * 2));
        CK(hipMalloc(&yg, 8192ull * 11008 * 2));
        {
            std::vector<uint16_t> h(8192ull * 4096);
            for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));
            CK(hipMemcpy(xg, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        for (int rep = 0; rep < 3; ++rep) {
            bench_gemm<0, 2, 2>("gemm 4 waves", 1024, 4096, 4096, bufs, xg, scales, yg);
            bench_gemm<0, 1, 4>("gemm 8 waves", 1024, 4096, 4096, bufs, xg, scales, yg);
        }
        bench_gemm<0, 2, 2>("gemm 4 waves", 4096, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<0, 1, 4>("gemm 8 waves", 4096, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<0, 2, 2>("gemm 4 waves", 1024, 11008, 4096, bufs_big, xg, scales, yg);
        bench_gemm<0, 1, 4>("gemm 8 waves", 1024, 11008, 4096, bufs_big, xg, scales, yg);
        bench_gemm<0, 2, 2>("gemm 4 waves", 512, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<0, 1, 4>("gemm 8 waves", 512, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<8, 1, 4>("gemm 8 waves -mfma", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<1, 1, 4>("gemm 8 waves -dma", 1024, 4096, 4096, bufs, xg, scales, yg);
        bench_gemm<23, 1, 4>("gemm 8 waves mfma only", 1024, 4096, 4096, bufs, xg, scales, yg);
    }
    if (!strcmp(what, "gemm1")) {  // single configuration for PMC runs
        eetq::f16 *xg, *yg;
        CK(hipMalloc(&xg, 1024ull * 4096 * 2));
        CK(hipMalloc(&yg, 1024ull * 4096 * 2));
        std::vector<uint16_t> h(1024ull * 4096);
        for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));
        CK(hipMemcpy(xg, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        using namespace eetq::gemm;
        auto kern = gemm_tile_kernel<0, 2>;
        constexpr int THREADS8 = 256;
        constexpr int SMEM_BYTES = TileCfg<2>::SMEM_BYTES;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        for (int i = 0; i < 20; ++i)
            hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS8), SMEM_BYTES, 0, xg, (const uint8_t*)bufs[i % bufs.size()], scales,
                               yg, 1024, 4096, 4096, 4096, eetq::Epilogue{});
        CK(hipDeviceSynchronize());
        auto gk = eetq::gemv::gemv_kernel<1, 16, 4, true, true, 1, 8>;
        for (int i = 0; i < 20; ++i)
            hipLaunchKernelGGL(gk, dim3(256), dim3(1024), (unsigned)eetq::gemv::gemv_smem_bytes(1, 4096, 16, true), 0, x,
                               (const uint8_t*)bufs[i % bufs.size()], scales, y, 4096, 4096, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, (const eetq::f16*)nullptr, 0, 0.f);
        CK(hipDeviceSynchronize());
    }
    if (!strcmp(what, "splitk")) {
        printf("--- split-K medium-batch kernel vs the unsplit tile ---\n");
        eetq::f16 *xs, *ys;
        CK(hipMalloc(&xs, 128ull * 13824 * 2));
        CK(hipMalloc(&ys, 128ull * 13824 * 2));
        {
            std::vector<uint16_t> h(128ull * 13824);
            for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));
            CK(hipMemcpy(xs, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        float*    slabs;
        unsigned* tickets;
        CK(hipMalloc(&slabs, 64ull << 20));
        CK(hipMalloc(&tickets, 4096 * 4));
        CK(hipMemset(tickets, 0, 4096 * 4));
        bench_mid<2, 3>("mid MT2 (round 1)", 64, 4096, 4096, bufs, xs, scales, ys);
        bench_splitk<2, 1, 3>("ring", 64, 4096, 4096, 1, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 1, 3>("ring", 64, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 1, 2>("ring", 64, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3>("ring", 64, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3>("ring", 64, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2>("ring", 64, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<1, 1, 3>("ring", 32, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<1, 2, 3>("ring", 32, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<4, 1, 2>("ring", 128, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<4, 2, 2>("ring", 128, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3>("ring K=11008", 64, 4096, 11008, 4, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3>("ring N=11008", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2>("ring N=11008", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<4, 2, 2>("ring N=11008 M=128", 128, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
    }
    if (!strcmp(what, "deepk")) {  // round 3: separate ring depths for activations (SA) and weights (SB)
        printf("--- split-K kernel: shared ring (SA == SB) vs a deep weight ring (SB > SA); graph step is the figure of merit ---\n");
        eetq::f16 *xs, *ys;
        CK(hipMalloc(&xs, 128ull * 13824 * 2));
        CK(hipMalloc(&ys, 128ull * 13824 * 2));
        {
            std::vector<uint16_t> h(128ull * 13824);
            for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));
            CK(hipMemcpy(xs, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        float*    slabs;
        unsigned* tickets;
        CK(hipMalloc(&slabs, 64ull << 20));
        CK(hipMalloc(&tickets, 4096 * 4));
        CK(hipMemset(tickets, 0, 4096 * 4));
        std::vector<uint8_t*> bufs13(8);   // 5120 x 13824 (= 13824 x 5120 bytes): Llama-13B gate / up / down
        for (auto& p : bufs13) {
            CK(hipMalloc(&p, 5120ull * 13824));
            CK(hipMemcpy(p, host.data(), 4096ull * 11008, hipMemcpyHostToDevice));
            CK(hipMemcpy(p + 4096ull * 11008, host.data(), 5120ull * 13824 - 4096ull * 11008, hipMemcpyHostToDevice));
        }
        for (int M : {32, 17}) {
            bench_splitk<1, 1, 3, 3>("M<=32 shared", M, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 1, 3, 8>("M<=32 deep", M, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 1, 3, 8>("M<=32 deep", M, 4096, 4096, 1, bufs, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 2, 3, 6>("M<=32 deep", M, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 2, 3, 6>("M<=32 deep", M, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        }
        bench_splitk<2, 1, 3, 3>("M=64 shared", 64, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 1, 3, 8>("M=64 deep", 64, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 1, 2, 8>("M=64 deep", 64, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 1, 3, 8>("M=64 deep", 64, 4096, 4096, 1, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3, 4>("M=64 deep", 64, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3, 4>("M=64 deep", 64, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2, 6>("M=64 deep", 64, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2, 6>("M=64 deep", 64, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<3, 1, 2, 2>("M=96 shared", 96, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<3, 1, 2, 8>("M=96 deep", 96, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<3, 2, 2, 4>("M=96 deep", 96, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<4, 2, 2, 2>("M=128 shared", 128, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<4, 1, 2, 4>("M=128 deep", 128, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
        bench_splitk<4, 1, 2, 4>("M=128 deep", 128, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
        if (argc > 2 && !strcmp(argv[2], "w8")) {  // four vs eight waves per workgroup, shared 3x3 ring
            printf("-- 4 waves (one per k tile) vs 8 waves (two per k tile, one 32-deep half each) --\n");
#define W48(MT_, NB_, M_, N_, K_, S_, BUFS)                                                                            \
    bench_splitk<MT_, NB_, 3, 3, true, 4>("4 waves", M_, N_, K_, S_, BUFS, xs, scales, ys, slabs, tickets);            \
    bench_splitk<MT_, NB_, 3, 3, true, 8>("8 waves", M_, N_, K_, S_, BUFS, xs, scales, ys, slabs, tickets);
            W48(1, 1, 17, 4096, 4096, 2, bufs)
            W48(1, 1, 32, 4096, 4096, 2, bufs)
            W48(1, 2, 32, 4096, 4096, 4, bufs)
            W48(2, 1, 64, 4096, 4096, 2, bufs)
            W48(2, 2, 64, 4096, 4096, 4, bufs)
            W48(2, 2, 64, 11008, 4096, 1, bufs_big)
            W48(1, 2, 32, 11008, 4096, 1, bufs_big)
            W48(2, 2, 64, 4096, 11008, 4, bufs_big)
            W48(1, 1, 32, 4096, 11008, 2, bufs_big)
            W48(2, 2, 64, 13824, 5120, 1, bufs13)
            W48(1, 2, 32, 13824, 5120, 1, bufs13)
            W48(1, 2, 17, 13824, 5120, 1, bufs13)
            W48(2, 2, 64, 5120, 13824, 2, bufs13)
            W48(1, 2, 32, 5120, 13824, 2, bufs13)
#undef W48
            return 0;
        }
        if (argc > 2 && !strcmp(argv[2], "inter")) {  // DMA pieces issued between the MFMA groups vs all before the first dequant
            printf("-- DMA issue: block (before the dequant) vs interleaved with the MFMA groups; 4 and 8 waves --\n");
#define INTERCMP(MT_, NB_, M_, N_, K_, S_, BUFS)                                                                                 \
    bench_splitk<MT_, NB_, 3, 3, true, 4, false>("w4 block", M_, N_, K_, S_, BUFS, xs, scales, ys, slabs, tickets);               \
    bench_splitk<MT_, NB_, 3, 3, true, 4, true>("w4 interleaved", M_, N_, K_, S_, BUFS, xs, scales, ys, slabs, tickets);          \
    bench_splitk<MT_, NB_, 3, 3, true, 8, true>("w8 interleaved", M_, N_, K_, S_, BUFS, xs, scales, ys, slabs, tickets);
            INTERCMP(1, 1, 32, 4096, 4096, 2, bufs)
            INTERCMP(2, 1, 64, 4096, 4096, 2, bufs)
            INTERCMP(2, 2, 64, 4096, 4096, 4, bufs)
            INTERCMP(2, 2, 64, 11008, 4096, 1, bufs_big)
            INTERCMP(1, 2, 32, 11008, 4096, 1, bufs_big)
            INTERCMP(2, 2, 64, 4096, 11008, 4, bufs_big)
            INTERCMP(2, 2, 64, 13824, 5120, 1, bufs13)
            INTERCMP(1, 2, 32, 13824, 5120, 1, bufs13)
            INTERCMP(2, 2, 64, 5120, 13824, 2, bufs13)
#undef INTERCMP
            return 0;
        }
        if (argc > 2 && !strcmp(argv[2], "ring4")) {  // shared ring of 3 vs 4 stages (activation lookahead 2 vs 3 steps)
            printf("-- shared ring 3x3 vs 4x4 --\n");
#define R34(MT_, NB_, M_, N_, K_, S_, BUFS)                                                                        \
    bench_splitk<MT_, NB_, 3, 3>("ring 3x3", M_, N_, K_, S_, BUFS, xs, scales, ys, slabs, tickets);                 \
    bench_splitk<MT_, NB_, 4, 4>("ring 4x4", M_, N_, K_, S_, BUFS, xs, scales, ys, slabs, tickets);
            R34(1, 1, 32, 4096, 4096, 2, bufs)
            R34(1, 2, 32, 4096, 4096, 4, bufs)
            R34(2, 1, 64, 4096, 4096, 2, bufs)
            R34(1, 2, 32, 11008, 4096, 1, bufs_big)
            R34(1, 1, 32, 4096, 11008, 2, bufs_big)
            R34(2, 1, 64, 4096, 11008, 2, bufs_big)
            R34(1, 2, 32, 13824, 5120, 1, bufs13)
            R34(1, 2, 17, 13824, 5120, 1, bufs13)
            R34(2, 1, 64, 13824, 5120, 1, bufs13)
            R34(1, 2, 32, 5120, 13824, 2, bufs13)
#undef R34
            return 0;
        }
        if (argc > 2 && !strcmp(argv[2], "bn128")) {  // 128-column blocks (NB = 4): the activations re-read half as often
            printf("-- BN = 64 (shipping plans) vs BN = 128 --\n");
            bench_splitk<2, 2, 3, 3>("M=64 BN=64", 64, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 4, 2, 3>("M=64 BN=128", 64, 13824, 5120, 2, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 4, 2, 3>("M=64 BN=128", 64, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 4, 2, 3>("M=64 BN=128", 64, 13824, 5120, 4, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 2, 3, 3>("M=32 BN=64", 32, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 4, 3, 3>("M=32 BN=128", 32, 13824, 5120, 2, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 4, 3, 3>("M=32 BN=128", 32, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 2, 3, 3>("M=64 BN=64", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 4, 2, 3>("M=64 BN=128", 64, 11008, 4096, 2, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 2, 3, 3>("M=32 BN=64", 32, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 4, 3, 3>("M=32 BN=128", 32, 11008, 4096, 2, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 2, 3, 3>("M=64 BN=64", 64, 4096, 11008, 4, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 4, 2, 3>("M=64 BN=128", 64, 4096, 11008, 4, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 2, 3, 3, false>("M=64 BN=64", 64, 5120, 13824, 2, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 4, 2, 3, false>("M=64 BN=128", 64, 5120, 13824, 4, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 1, 3, 3>("M=64 BN=32", 64, 4096, 4096, 2, bufs, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 4, 2, 3>("M=64 BN=128", 64, 4096, 4096, 4, bufs, xs, scales, ys, slabs, tickets);
            return 0;
        }
        if (argc > 2 && !strcmp(argv[2], "percu")) {  // unsplit plans small enough for two or three workgroups per CU
            printf("-- unsplit (S = 1) plans: one workgroup per CU (rings > 80 KiB) vs two / three per CU (ring 2x2) --\n");
            bench_splitk<2, 2, 3, 3>("M=64 BN=64 3x3 1/CU", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 1, 2, 2>("M=64 BN=32 2x2 2/CU", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 1, 3, 3>("M=64 BN=32 3x3 1/CU", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 2, 3, 3>("M=32 BN=64 3x3 1/CU", 32, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 2, 2, 2>("M=32 BN=64 2x2 2/CU", 32, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 1, 2, 2>("M=32 BN=32 2x2 3/CU", 32, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 2, 3, 3>("M=64 BN=64 3x3 1/CU", 64, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<2, 1, 2, 2>("M=64 BN=32 2x2 2/CU", 64, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 2, 3, 3>("M=32 BN=64 3x3 1/CU", 32, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 2, 2, 2>("M=32 BN=64 2x2 2/CU", 32, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 1, 2, 2>("M=32 BN=32 2x2 3/CU", 32, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 1, 2, 2>("M=17 BN=32 2x2 3/CU", 17, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            bench_splitk<1, 2, 3, 3>("M=17 BN=64 3x3 1/CU", 17, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
            return 0;
        }
        printf("-- 4096 x 11008 / 11008 x 4096 --\n");
        bench_splitk<2, 2, 3, 3>("N=11008 shared", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3, 4>("N=11008 deep", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2, 6>("N=11008 deep", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 1, 3, 8>("N=11008 deep", 64, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<1, 2, 3, 6>("N=11008 deep", 32, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<1, 1, 3, 8>("N=11008 deep", 32, 11008, 4096, 1, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3, 3>("K=11008 shared", 64, 4096, 11008, 4, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3, 4>("K=11008 deep", 64, 4096, 11008, 4, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2, 6>("K=11008 deep", 64, 4096, 11008, 4, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 1, 3, 8>("K=11008 deep", 64, 4096, 11008, 2, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2, 6>("K=11008 deep", 64, 4096, 11008, 2, bufs_big, xs, scales, ys, slabs, tickets);
        bench_splitk<1, 1, 3, 8>("K=11008 deep", 32, 4096, 11008, 2, bufs_big, xs, scales, ys, slabs, tickets);
        printf("-- Llama-13B: 5120 x 13824 (N = 13824) and 13824 x 5120 --\n");
        bench_splitk<2, 2, 3, 3>("N=13824 shared", 64, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3, 4>("N=13824 deep", 64, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2, 6>("N=13824 deep", 64, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
        bench_splitk<1, 2, 3, 6>("N=13824 deep", 32, 13824, 5120, 1, bufs13, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3, 3, false>("K=13824 shared", 64, 5120, 13824, 2, bufs13, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2, 6, false>("K=13824 deep", 64, 5120, 13824, 2, bufs13, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 3, 4, false>("K=13824 deep", 64, 5120, 13824, 2, bufs13, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 1, 3, 8, false>("K=13824 deep", 64, 5120, 13824, 1, bufs13, xs, scales, ys, slabs, tickets);
        bench_splitk<2, 2, 2, 6, false>("K=13824 deep", 64, 5120, 13824, 4, bufs13, xs, scales, ys, slabs, tickets);
        bench_splitk<1, 2, 3, 6, false>("K=13824 deep", 32, 5120, 13824, 2, bufs13, xs, scales, ys, slabs, tickets);
        return 0;
    }
    if (!strcmp(what, "all") || !strcmp(what, "mid")) {
        printf("--- medium-batch tile kernel (32-column tiles, 256-deep K steps) ---\n");
        eetq::f16 *xs, *ys;
        CK(hipMalloc(&xs, 128ull * 13824 * 2));
        CK(hipMalloc(&ys, 128ull * 13824 * 2));
        {
            std::vector<uint16_t> h(128ull * 13824);
            for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));
            CK(hipMemcpy(xs, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        bench_mid<1, 3>("mid MT1 S3", 32, 4096, 4096, bufs, xs, scales, ys);
        bench_mid<2, 3>("mid MT2 S3", 64, 4096, 4096, bufs, xs, scales, ys);
        bench_mid<3, 2>("mid MT3 S2", 96, 4096, 4096, bufs, xs, scales, ys);
        bench_mid<4, 2>("mid MT4 S2", 128, 4096, 4096, bufs, xs, scales, ys);
        bench_mid<2, 2>("mid MT2 S2 N=11008", 64, 11008, 4096, bufs_big, xs, scales, ys);
        bench_mid<2, 3>("mid MT2 S3 K=11008", 64, 4096, 11008, bufs_big, xs, scales, ys);
        bench_mid<4, 2>("mid MT4 S2 N=11008", 128, 11008, 4096, bufs_big, xs, scales, ys);
        bench_mid<1, 2>("mid MT1 S2 N=11008", 32, 11008, 4096, bufs_big, xs, scales, ys);
    }
    if (!strcmp(what, "all") || !strcmp(what, "streamk")) {
        printf("--- stream MFMA kernel with register-resident activations ---\n");
        eetq::f16* xs;
        CK(hipMalloc(&xs, 128ull * 13824 * 2));
        {
            std::vector<uint16_t> h(128ull * 13824);
            for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));
            CK(hipMemcpy(xs, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        eetq::f16* ys;
        CK(hipMalloc(&ys, 128ull * 11008 * 2));
        uint8_t* huge;
        CK(hipMalloc(&huge, 4096ull * 22016 * 12));
        std::vector<uint8_t*> bufs_huge;
        for (int i = 0; i < 12; ++i) bufs_huge.push_back(huge + (size_t)i * 4096 * 22016);
        CK(hipMemset(huge, 0x5a, 4096ull * 22016 * 12));
        for (int M : {8, 16}) {
            printf("NT sweep M=%d\n", M);
            bench_streamk<1, 1, 16, 2, 4>("NT1 16x2 o4 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 2, 16, 2, 4>("NT2 16x2 o4 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 2, 16, 2, 2>("NT2 16x2 o2 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 2, 8, 4, 4>("NT2 8x4 o4 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 4, 16, 2, 2>("NT4 16x2 o2 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 4, 8, 4, 2>("NT4 8x4 o2 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 1, 16, 2, 4>("NT1 16x2 o4 N=11008", M, 11008, 4096, bufs_big, xs, scales, ys);
            bench_streamk<1, 2, 16, 2, 4>("NT2 16x2 o4 N=11008", M, 11008, 4096, bufs_big, xs, scales, ys);
            bench_streamk<1, 2, 8, 4, 4>("NT2 8x4 o4 N=11008", M, 11008, 4096, bufs_big, xs, scales, ys);
            bench_streamk<1, 1, 16, 2, 4>("NT1 16x2 o4 K=11008", M, 4096, 11008, bufs_big, xs, scales, ys);
            bench_streamk<1, 1, 16, 4, 4>("NT1 16x4 o4 K=11008", M, 4096, 11008, bufs_big, xs, scales, ys);
            bench_streamk<1, 1, 16, 4, 2>("NT1 16x4 o2 K=11008", M, 4096, 11008, bufs_big, xs, scales, ys);
        }
        if (argc > 2)
        for (int M : {4, 8}) {
            printf("M=%d\n", M);
            bench_streamk<1, 1, 16, 2, 4>("MT1 16x2 o4 (shipping)", M, 4096, 4096, bufs, xs, scales, ys);
            bench_streamk<1, 1, 16, 4, 4>("MT1 16x4 o4", M, 4096, 4096, bufs, xs, scales, ys);
            bench_streamk<1, 1, 16, 2, 8>("MT1 16x2 o8", M, 4096, 4096, bufs, xs, scales, ys);
            bench_streamk<1, 1, 16, 4, 8>("MT1 16x4 o8", M, 4096, 4096, bufs, xs, scales, ys);
            bench_streamk<1, 1, 8, 4, 4>("MT1 8x4 o4", M, 4096, 4096, bufs, xs, scales, ys);
            bench_streamk<1, 1, 16, 2, 4>("MT1 16x2 o4 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 1, 16, 4, 4>("MT1 16x4 o4 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 1, 16, 2, 8>("MT1 16x2 o8 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 1, 16, 4, 8>("MT1 16x4 o8 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
            bench_streamk<1, 1, 8, 4, 8>("MT1 8x4 o8 N=22016", M, 22016, 4096, bufs_huge, xs, scales, ys);
        }
    }
    if (!strcmp(what, "all") || !strcmp(what, "streamk")) {
        printf("--- stream MFMA kernel with register-resident activations ---\n");
        eetq::f16* xs;
        CK(hipMalloc(&xs, 128ull * 13824 * 2));
        {
            std::vector<uint16_t> h(128ull * 13824);
            for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));
            CK(hipMemcpy(xs, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        eetq::f16* ys;
        CK(hipMalloc(&ys, 128ull * 11008 * 2));
        bench_streamk<1, 1, 16, 2, 4>("MT1 NT1 16x2", 8, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<1, 2, 16, 2, 4>("MT1 NT2 16x2", 8, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<1, 1, 16, 2, 4>("MT1 NT1 16x2", 16, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<1, 1, 16, 2, 4>("MT1 NT1 16x2 K=11008", 8, 4096, 11008, bufs_big, xs, scales, ys);
        bench_streamk<1, 2, 16, 2, 4>("MT1 NT2 16x2 K=11008", 8, 4096, 11008, bufs_big, xs, scales, ys);
        bench_streamk<1, 1, 16, 4, 4>("MT1 NT1 16x4 K=11008", 8, 4096, 11008, bufs_big, xs, scales, ys);
        bench_streamk<1, 1, 16, 2, 4>("MT1 NT1 16x2 N=11008", 8, 11008, 4096, bufs_big, xs, scales, ys);
        bench_streamk<2, 1, 16, 2, 4>("MT2 NT1 16x2", 32, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<2, 2, 16, 2, 4>("MT2 NT2 16x2", 32, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<4, 1, 16, 2, 4>("MT4 NT1 16x2", 64, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<4, 2, 16, 2, 4>("MT4 NT2 16x2", 64, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<4, 2, 16, 1, 4>("MT4 NT2 16x1", 64, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<4, 2, 8, 2, 2>("MT4 NT2 8x2", 64, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<4, 2, 16, 2, 4>("MT4 NT2 16x2 N=11008", 64, 11008, 4096, bufs_big, xs, scales, ys);
        bench_streamk<4, 2, 16, 2, 4>("MT4 NT2 16x2 K=11008", 64, 4096, 11008, bufs_big, xs, scales, ys);
        bench_streamk<8, 1, 16, 1, 4>("MT8 NT1 16x1", 128, 4096, 4096, bufs, xs, scales, ys);
        bench_streamk<8, 1, 16, 2, 4>("MT8 NT1 16x2", 128, 4096, 4096, bufs, xs, scales, ys);
    }
    return 0;
}
