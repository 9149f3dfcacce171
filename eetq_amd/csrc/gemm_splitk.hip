// Split-K medium-batch (M <= 128) MFMA dequant-GEMM launcher; kernel and design notes in gemm_splitk_kernel.hpp.
// Reference behaviour matched: the small-M preference of the CUTLASS tile heuristic (cutlass_heuristic.cc:123-206) -- the
// reference never splits K because its wrapper passes no workspace (fpA_intB_gemm_wrapper.cu:169-170); here the workspace
// is owned by the library so that the operator signature stays workspace-free.
#include <mutex>

#include "gemm_splitk_kernel.hpp"

namespace eetq {

namespace {

// ---- scratch: fp32 partial tiles + per-tile tickets --------------------------------------------------------------------
// One arena per device, cut into kRegions regions; a stream is given a region the first time it launches a split-K GEMM and
// keeps it (a HIP graph replays with the region of the stream it was captured on).  Launches that run CONCURRENTLY must use
// different regions: guaranteed for up to kRegions distinct launch streams per device; beyond that, and for graphs captured
// on one stream but replayed concurrently on several, set EETQ_AMD_SPLITK=0 (the dispatcher then uses the unsplit kernels).
constexpr int    kRegions       = 16;
constexpr size_t kRegionSlabs   = 40ull << 20;  // bytes of fp32 partial tiles per region (largest plan: ~33 MiB)
constexpr size_t kRegionTickets = 4096;         // tiles per launch (N <= 4096 * 32 columns)

struct Arena {
    uint8_t*    base = nullptr;
    hipStream_t owner[kRegions] = {};
    bool        used[kRegions]  = {};
};
std::mutex g_mutex;
Arena      g_arena[64];

int region_for(hipStream_t stream, float** slabs, unsigned** tickets)
{
    int dev = 0;
    EETQ_TRY_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mutex);
    Arena&                      a = g_arena[dev & 63];
    const size_t region_bytes = kRegionSlabs + kRegionTickets * sizeof(unsigned);
    if (!a.base) {
        // not capturable: the first split-K launch on a device must happen outside graph capture (any eager warm-up does it)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] split-K scratch cannot be created during graph capture");
        EETQ_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&a.base), region_bytes * kRegions));
        EETQ_TRY_HIP(hipMemset(a.base, 0, region_bytes * kRegions));  // tickets start at 0 and only ever grow by S per tile
    }
    int r = -1;
    for (int i = 0; i < kRegions; ++i)
        if (a.used[i] && a.owner[i] == stream) r = i;
    if (r < 0) {
        for (int i = 0; i < kRegions && r < 0; ++i)
            if (!a.used[i]) r = i;
        if (r < 0) r = (int)(((uintptr_t)stream >> 4) % kRegions);  // more streams than regions: shared (see above)
        a.used[r]  = true;
        a.owner[r] = stream;
    }
    *slabs   = reinterpret_cast<float*>(a.base + (size_t)r * region_bytes);
    *tickets = reinterpret_cast<unsigned*>(a.base + (size_t)r * region_bytes + kRegionSlabs);
    return EETQ_OK;
}

template <int MT, int NB, int D, bool KFULL>
int launch_full(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K, int S,
                hipStream_t stream)
{
    using C   = gemm_splitk::Cfg<MT, NB, D>;
    auto kern = gemm_splitk::gemm_splitk_kernel<MT, NB, D, KFULL>;
    if (C::kSmem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    const int tiles = (N + C::kBN - 1) / C::kBN;
    float*    slabs   = nullptr;
    unsigned* tickets = nullptr;
    if (S > 1) {
        if ((size_t)tiles > kRegionTickets || (size_t)tiles * S * C::kSlabFloats * 4 > kRegionSlabs) S = 1;
        else {
            int st = region_for(stream, &slabs, &tickets);
            if (st != EETQ_OK) return st;
        }
    }
    launch_kernel(kern, dim3(tiles * S), dim3(gemm_splitk::kThreads), C::kSmem, stream, x, w, scales, y, M, N, K, S, slabs,
                  tickets, ep);
    return check_hip(hipGetLastError(), "gemm_splitk_kernel launch");
}

template <int MT, int NB, int D>
int launch_inst(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K, int S,
                hipStream_t stream)
{
    return K % 256 == 0 ? launch_full<MT, NB, D, true>(x, w, scales, ep, y, M, N, K, S, stream)
                        : launch_full<MT, NB, D, false>(x, w, scales, ep, y, M, N, K, S, stream);
}

// steps in flight per wave: as many as the LDS ring (16*MT KiB per step and workgroup) and the 6-bit vmcnt allow
template <int MT>
int launch_mt(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K, int nb, int S,
              hipStream_t stream)
{
    constexpr int D = MT == 1 ? 4 : MT == 2 ? 4 : MT == 3 ? 3 : 2;
    if (nb == 2) return launch_inst<MT, 2, D>(x, w, scales, ep, y, M, N, K, S, stream);
    return launch_inst<MT, 1, D>(x, w, scales, ep, y, M, N, K, S, stream);
}

}  // namespace

// Plan (column blocks per workgroup, K slices, ring depth) for a shape.  What a compute unit must ingest through its
// ~57 B/clk vector memory path bounds these shapes, so the plan minimises the busiest CU's bytes:
//     rounds(workgroups / CUs) * ((32*MT*2 + 32*NB) * K / S)        [+ the reduction's slab traffic when S > 1]
// subject to >= 2 K steps per slice.  Measured on MI355X: profiles/r02_kbench_splitk.txt.
void splitk_plan(int M, int N, int K, int* nb_out, int* s_out)
{
    const int  MT    = (M + 31) / 32;
    const int  ncu   = device_cu_count();
    const int  steps = (K / 64 + 3) / 4;
    double     best  = 1e30;
    int        bnb = 1, bs = 1;
    for (int nb = 1; nb <= 2; ++nb) {
        if (nb == 2 && N % 64 != 0 && N < 2048) continue;
        const int tiles = (N + 32 * nb - 1) / (32 * nb);
        for (int s = 1; s <= gemm_splitk::kMaxSlices; s *= 2) {
            if (steps / s < 2 && s > 1) continue;
            const int    wgs    = tiles * s;
            const int    rounds = (wgs + ncu - 1) / ncu;
            const double bytes  = (double)(32 * MT * 2 + 32 * nb) * K / s;
            // per-workgroup fixed cost (prologue, barriers, cross-wave reduction; + publish / ticket / slab reads when split),
            // in units of ingest bytes (~120 B/ns per CU)
            const double fixed = 40e3 + (s > 1 ? 60e3 + 4096.0 * MT * nb * (s + 1) : 0.0);
            const double cost  = rounds * (bytes + fixed);
            if (cost < best) {
                best = cost;
                bnb  = nb;
                bs   = s;
            }
        }
    }
    *nb_out = bnb;
    *s_out  = bs;
}

int launch_gemm_splitk(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                       hipStream_t stream, int force_nb, int force_s)
{
    if (M < 1 || M > kMidMaxM) return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] split-K tile path supports 1 <= M <= 128");
    EETQ_REQUIRE((size_t)M * K * 2 < (1ull << 31) && (size_t)N * K < (1ull << 31),
                 "operand larger than 2 GiB is not supported by the buffer-addressed DMA path");
    int nb, s;
    splitk_plan(M, N, K, &nb, &s);
    if (force_nb) nb = force_nb;
    if (force_s) s = force_s;
    EETQ_REQUIRE((nb == 1 || nb == 2) && (s == 1 || s == 2 || s == 4), "invalid split-K plan");
    switch ((M + 31) / 32) {
        case 1: return launch_mt<1>(x, w, scales, ep, y, M, N, K, nb, s, stream);
        case 2: return launch_mt<2>(x, w, scales, ep, y, M, N, K, nb, s, stream);
        case 3: return launch_mt<3>(x, w, scales, ep, y, M, N, K, nb, s, stream);
        default: return launch_mt<4>(x, w, scales, ep, y, M, N, K, nb, s, stream);
    }
}

}  // namespace eetq
