// Probe (no torch; round 6, verdict item 1c): the four DEPENDENT weight streams of a Llama-2-13B decoder layer's decode step
//   q|k|v 5120 x 15360  ->  o 5120 x 5120  ->  gate|up 5120 x 27648  ->  down 13824 x 5120      (K x N, int8 tiles, 317 MB per layer)
// as (A) one launch per step -- what the graph decoder replays today -- and (P) ONE persistent launch per chain of layers with the
// next step's weights prefetched DEEP INTO LDS across the step boundary (per-wave LDS-DMA rings, 96 KiB per CU = 24 MB on the chip =
// ~4 us of stream) and the activation vector handed over inside the launch (write-through stores, per-XCD arrival counters, L1-
// bypassing reads).  The probe measures TIME, not results: both forms move the same bytes, run the same dequantise + v_dot2
// arithmetic per weight byte and the same per-step reductions, but the persistent form cuts a step's weight tiles evenly over the
// 256 workgroups regardless of tile rows (its outputs are not a GEMV's).  What it answers: is a step boundary inside a launch,
// with its all-to-all hand-over hidden behind prefetched weights, cheaper than a dependent launch boundary -- by how much per
// step -- at step sizes of 26 - 142 MB (the 16 MiB chain of tools/chain_probe.hip lost: too short for any prefetch credit).
// Every spin is bounded (a stuck wait sets an error word and falls through).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/layer_probe.hip -o tools/layer_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../eetq_amd/csrc/gemv_kernel.hpp"

namespace eetq {
void set_error(const std::string&) {}
int  fail(int c, const std::string&) { return c; }
int  check_hip(hipError_t e, const char*) { return e == hipSuccess ? 0 : -2; }
ProfEvents next_prof_events() { return {}; }
}  // namespace eetq

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

using namespace eetq;

constexpr int      kWaves     = 16;
constexpr int      kRing      = 6;       // 1 KiB slots per wave
constexpr int      kShards    = 8;
constexpr int      kShardStep = 32;      // dwords between arrival counters (one 128-byte line each)
constexpr unsigned kSpinLimit = 400000;
constexpr int      kMaxK      = 13824;

struct Step {
    const uint8_t* w;      // K * N bytes of tiles
    int            K, N;   // N: columns streamed (outputs written: N / 256 per workgroup, enough for the probe)
};
constexpr int kMaxSteps = 64;
struct ChainArgs {
    Step        steps[kMaxSteps];   // BY VALUE: kernel arguments are read with scalar loads; a table in global memory is read
                                    // with a vector load + vmcnt(0) wherever the compiler cannot prove it constant -- in the
                                    // refill of every ring slot, which emptied the ring once per tile (3.5 TB/s)
    int         nsteps;
    const f16*  scales;
    f16*        xbuf;    // 2 x kMaxK halfs: step e reads xbuf[e & 1], writes xbuf[(e + 1) & 1]
    unsigned*   ctr;     // [nsteps][kShards * kShardStep], zeroed before the launch
    unsigned*   err;
    int         idle_every;   // > 0: after every idle_every-th step the workgroups sit out `idle_cycles` (a phase that needs no HBM: the attention)
    int         idle_cycles;
};

typedef __attribute__((address_space(3))) void        lds_void;
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;

// per-step one-launch form of the same synthetic work (flat tile split, same arithmetic): the launch-boundary baseline that moves
// exactly what the persistent form moves
__device__ __forceinline__ float tile_dot(const u32x4& wv, f16x2 scale2, const u32x4& xa, const u32x4& xb)
{
    f16x2 wq[8];
    dequant_16(wv, scale2, wq);
    const u32   xd[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
    float       acc   = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_fdot2(wq[i], as_f16x2(xd[i]), acc, false);
    return acc;
}

template <bool PERSISTENT>
__global__ __launch_bounds__(kWaves * 64, 4) void layer_kernel(ChainArgs a, int first_step)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* ring = smem;                                             // kWaves * kRing KiB
    f16*     xs   = reinterpret_cast<f16*>(smem + kWaves * kRing * 1024);   // kMaxK halfs
    float*   red  = reinterpret_cast<float*>(smem + kWaves * kRing * 1024 + kMaxK * 2);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int b = blockIdx.x, G = gridDim.x;
    const u32 sraw = reinterpret_cast<const uint16_t*>(a.scales)[c];
    const f16x2 scale2 = as_f16x2(sraw | (sraw << 16));
    uint8_t* my_ring = ring + wave * kRing * 1024;
    const int lds_ring = (int)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)my_ring;
    const int lds_xs   = (int)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(smem + kWaves * kRing * 1024);

    const int e0 = first_step, e1 = PERSISTENT ? a.nsteps : first_step + 1;
    // flat tile stream of this wave: for step e, tiles lo_e + wave, + kWaves, ... < hi_e where [lo_e, hi_e) = this workgroup's share.
    // Everything about the cursors is 32-bit, wave-uniform and division-free (G = 2^lg workgroups; host-checked): a 64-bit divide
    // or a table read from global memory in the refill path costs more than the tile it fetches.
    const int lg = 31 - __builtin_clz((unsigned)G);
    auto share = [&](int e, int& lo, int& hi, int& KT) {
        KT = a.steps[e].K >> 6;
        const unsigned T = (unsigned)(a.steps[e].N >> 4) * (unsigned)KT;
        lo = (int)(((unsigned long long)T * (unsigned)b) >> lg);
        hi = (int)(((unsigned long long)T * (unsigned)(b + 1)) >> lg);
    };
    // issue cursor (runs kRing tiles ahead of the consume cursor, across step boundaries)
    int ie = e0, ilo, ihi, it, iKT;
    share(ie, ilo, ihi, iKT);
    it = ilo + wave;
    __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.steps[ie].w), 0, a.steps[ie].K * a.steps[ie].N, 0x00020000);
    int islot = 0;
    auto issue = [&]() {
        if (it >= ihi && ie + 1 < e1) {   // next step's share (wave-uniform; a share is never empty)
            ++ie;
            share(ie, ilo, ihi, iKT);
            it  = ilo + wave;
            irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.steps[ie].w), 0, a.steps[ie].K * a.steps[ie].N, 0x00020000);
        }
        const int t = it < ihi ? it : ihi - 1;   // past the end of everything: a clamped, unused refill
        __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (lds_void*)(my_ring + islot * 1024), 16, lane * 16, t * 1024, 0, 2 /* nt */);
        it += kWaves;
        islot = islot + 1 == kRing ? 0 : islot + 1;
    };

    // the first step's activations
    {
        const Step s = a.steps[e0];
        const u32x4* xg = reinterpret_cast<const u32x4*>(a.xbuf + (e0 & 1) * kMaxK);
        for (int v = tid; v < (s.K >> 3); v += kWaves * 64) reinterpret_cast<u32x4*>(xs)[v] = xg[v];
    }
#pragma unroll
    for (int i = 0; i < kRing; ++i) issue();
    __syncthreads();

    int cslot = 0;
    for (int e = e0; e < e1; ++e) {
        int lo, hi, KT;
        share(e, lo, hi, KT);
        float acc = 0.f;
        int   kt  = (lo + wave) % KT;
        for (int t = lo + wave; t < hi; t += kWaves) {
            // the oldest of the kRing DMAs in flight has landed once kRing - 1 are outstanding.  All three LDS reads are hand-written:
            // the compiler orders every LDS read it can see behind ALL pending LDS-DMAs (vmcnt(0)), i.e. it would empty the ring
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRing - 1) : "memory");
            u32x4 wv, xa, xb;
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(wv), "=&v"(xa), "=&v"(xb)
                         : "v"(lds_ring + cslot * 1024 + lane * 16), "v"(lds_xs + (kt * 64 + 16 * g) * 2)
                         : "memory");
            issue();   // refills the slot just read
            acc += tile_dot(wv, scale2, xa, xb);
            cslot = cslot + 1 == kRing ? 0 : cslot + 1;
            kt += kWaves;
            if (kt >= KT) kt -= KT;
        }
        // ---- end of the step: wave butterflies, cross-wave sum, this workgroup's outputs ----
        acc = sum_xor32(sum_xor16(acc));
        if (lane < 16) red[wave * 16 + lane] = acc;
        __syncthreads();
        const Step s   = a.steps[e];
        f16*       yv  = a.xbuf + ((e + 1) & 1) * kMaxK;
        const int  per = (s.N + G - 1) / G;   // outputs this workgroup owns
        if (wave == 0) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < kWaves; ++wv) v += red[wv * 16 + c];
            const f16 hv = (f16)(v * 1e-3f);
            for (int o = lane; o < per; o += 64) {
                const int idx = b * per + o;
                if (idx < kMaxK) {
                    if constexpr (PERSISTENT)
                        __hip_atomic_store(reinterpret_cast<unsigned short*>(yv) + idx, __builtin_bit_cast(unsigned short, hv),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        yv[idx] = hv;
                }
            }
        }
        if constexpr (PERSISTENT) {
            if (e + 1 == e1) break;
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (drains this wave's prefetches too: they land during the wait below anyway)
                if (lane == 0)
                    __hip_atomic_fetch_add(a.ctr + (size_t)e * kShards * kShardStep + (b & (kShards - 1)) * kShardStep, 1u,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // a phase that needs no HBM (the attention between q|k|v and o): sit it out, the rings keep filling
                if (a.idle_every > 0 && (e % a.idle_every) == 0) {
                    const unsigned long long t0 = __builtin_readcyclecounter();
                    while (__builtin_readcyclecounter() - t0 < (unsigned long long)a.idle_cycles) __builtin_amdgcn_s_sleep(8);
                }
                const unsigned target = (unsigned)(G / kShards);
                const unsigned* cs    = a.ctr + (size_t)e * kShards * kShardStep;
                for (unsigned spins = 0;; ++spins) {
                    const unsigned v = lane < kShards ? __hip_atomic_load(cs + lane * kShardStep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                    if (__all(v >= target)) break;
                    if (spins > kSpinLimit) {
                        if (lane == 0) atomicOr(a.err, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            // the next step's activations: L1-bypassing loads of what the other workgroups wrote through
            const Step sn = a.steps[e + 1];
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(yv, 0, kMaxK * 2, 0x00020000);
            for (int v = tid; v < (sn.K >> 3); v += kWaves * 64)
                reinterpret_cast<u32x4*>(xs)[v] = __builtin_amdgcn_raw_buffer_load_b128(xr, v * 16, 0, 16 /* sc1 */);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // every wave is at vmcnt 0 here only if it had no DMA in flight; re-prime so that kRing are outstanding again
            // (the waits above drained them): nothing to do -- the ring slots hold landed tiles, the consume loop's
            // `vmcnt(kRing - 1)` is then satisfied at once and its issue() keeps the ring full from there on
        }
    }
}

static double time_graph(hipGraphExec_t ge, hipStream_t s, int reps)
{
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        auto t0 = std::chrono::high_resolution_clock::now();
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::high_resolution_clock::now();
        best    = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    return best;
}

int main(int argc, char** argv)
{
    const int LAYERS = argc > 1 ? atoi(argv[1]) : 8;   // layers per chain (distinct weights: 317 MB each)
    const int shapes[4][2] = {{5120, 15360}, {5120, 5120}, {5120, 27648}, {13824, 5120}};
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int G = prop.multiProcessorCount;
    if (G & (G - 1)) {
        fprintf(stderr, "the probe wants a power-of-two CU count\n");
        return 1;
    }
    printf("device: %s  CUs=%d; %d layers x 4 steps per chain\n", prop.gcnArchName, G, LAYERS);
    std::vector<Step> hsteps;
    size_t            layer_bytes = 0;
    std::vector<uint8_t> host(13824ull * 5120 + 5120ull * 27648);
    srand(1);
    for (auto& v : host) v = (uint8_t)(rand() >> 7);
    for (int l = 0; l < LAYERS; ++l)
        for (int s = 0; s < 4; ++s) {
            const size_t bytes = (size_t)shapes[s][0] * shapes[s][1];
            uint8_t*     p;
            CK(hipMalloc(&p, bytes));
            CK(hipMemcpy(p, host.data() + (l * 977 + s * 131) % 4096, bytes, hipMemcpyHostToDevice));
            hsteps.push_back(Step{p, shapes[s][0], shapes[s][1]});
            if (l == 0) layer_bytes += bytes;
        }
    const int NS = (int)hsteps.size();
    if (NS > kMaxSteps) {
        fprintf(stderr, "at most %d layers\n", kMaxSteps / 4);
        return 1;
    }
    f16 *scales, *xbuf;
    CK(hipMalloc(&scales, 64));
    CK(hipMalloc(&xbuf, 2 * kMaxK * 2));
    std::vector<_Float16> hs(16, (_Float16)0.001f), hx(2 * kMaxK);
    for (auto& v : hx) v = (_Float16)((rand() & 0xffff) / 65536.0f - 0.5f);
    CK(hipMemcpy(scales, hs.data(), 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(xbuf, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    unsigned *ctr, *err;
    const size_t ctr_bytes = (size_t)NS * kShards * kShardStep * 4;
    CK(hipMalloc(&ctr, ctr_bytes));
    CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    const size_t smem = (size_t)kWaves * kRing * 1024 + kMaxK * 2 + kWaves * 16 * 4;
    CK(hipFuncSetAttribute((const void*)layer_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute((const void*)layer_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipStream_t st;
    CK(hipStreamCreate(&st));

    auto report = [&](const char* name, double us) {
        printf("%-78s %8.2f us per layer  %6.2f us per step  (%.0f GB/s)\n", name, us / LAYERS, us / NS, layer_bytes * LAYERS / us / 1e3);
    };
    // A: one launch per step (the synthetic kernel, flat tile split)
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        ChainArgs a;
        memcpy(a.steps, hsteps.data(), NS * sizeof(Step));
        a.nsteps = NS, a.scales = scales, a.xbuf = xbuf, a.ctr = ctr, a.err = err, a.idle_every = 0, a.idle_cycles = 0;
        for (int e = 0; e < NS; ++e) hipLaunchKernelGGL(layer_kernel<false>, dim3(G), dim3(kWaves * 64), smem, st, a, e);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        report("A  one launch per step (same synthetic work, LDS-DMA rings, flat tile split)", time_graph(ge, st, 5));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // P: one persistent launch for the whole chain, with and without an HBM-idle phase after every q|k|v step
    for (int idle_us : {0, 4, 8}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(ctr, 0, ctr_bytes, st));
        ChainArgs a;
        memcpy(a.steps, hsteps.data(), NS * sizeof(Step));
        a.nsteps = NS, a.scales = scales, a.xbuf = xbuf, a.ctr = ctr, a.err = err, a.idle_every = idle_us ? 4 : 0, a.idle_cycles = idle_us * 2100;
        hipLaunchKernelGGL(layer_kernel<true>, dim3(G), dim3(kWaves * 64), smem, st, a, 0);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double us = time_graph(ge, st, 5);
        unsigned     e  = 0;
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        char name[160];
        snprintf(name, sizeof name, "P  ONE persistent launch, in-launch hand-over, %d KiB per CU prefetched%s%s", kWaves * kRing,
                 idle_us ? (idle_us == 4 ? "; + 4 us HBM-idle phase per layer" : "; + 8 us HBM-idle phase per layer") : "", e ? "  [GIVE-UP WORD SET]" : "");
        report(name, us);
        if (idle_us) printf("%-78s %8.2f us per layer\n", "   ... minus the idle phase itself", us / LAYERS - idle_us);
        CK(hipMemset(err, 0, 4));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
