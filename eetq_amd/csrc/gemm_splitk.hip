// Split-K medium-batch MFMA dequant-GEMM launcher (AUTO: 9 <= M <= 128 on some shapes, row-group plans to M = 1024; the forced
// path takes any M <= 1024); kernel and design notes in gemm_splitk_kernel.hpp.
// Reference behaviour matched: the small-M preference of the CUTLASS tile heuristic (cutlass_heuristic.cc:123-206) -- the
// reference never splits K because its wrapper passes no workspace (fpA_intB_gemm_wrapper.cu:169-170); here the workspace
// is owned by the library so that the operator signature stays workspace-free.
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "gemm_splitk_kernel.hpp"

namespace eetq {

namespace {

// ---- scratch: fp32 partial tiles + per-tile tickets --------------------------------------------------------------------
// Up to kMaxRegions regions per device (EETQ_AMD_SPLITK_REGIONS lowers the cap; 0 = never split), each created the first
// time a stream launches a split-K GEMM and OWNED by that stream from then on (a HIP graph replays with the region of the
// stream it was captured on).  A region is never shared: a launch that cannot get its own -- every region taken by a live
// stream, the cap reached, or the region would have to be created during graph capture -- runs unsplit (S = 1), which needs
// no scratch.  Regions whose owner stream has been destroyed are taken over (after a device synchronisation).  Everything
// is freed by eetq_release_workspace().  Not covered: one captured graph replayed CONCURRENTLY on several streams (the
// replays share the capture stream's region) -- capture once per replay stream, or set EETQ_AMD_SPLITK=0.
constexpr int    kMaxRegions    = 16;
constexpr size_t kRegionSlabs   = 40ull << 20;  // bytes of fp32 partial tiles per region (largest plan: ~33 MiB)
constexpr size_t kRegionTickets = 4096;         // tiles per launch (N <= 4096 * 32 columns); one ticket array PER slice count
// a tile's ticket grows by S per launch and "last" is (old & (S-1)) == S-1: that only works if every launch that touches a
// ticket uses the same S, so S = 2 and S = 4 launches keep separate ticket arrays
constexpr size_t kRegionBytes   = kRegionSlabs + 2 * kRegionTickets * sizeof(unsigned);

struct Region {
    uint8_t*           base  = nullptr;
    hipStream_t        owner = nullptr;
    bool               used  = false;
    unsigned long long uses  = 0;  // launches handed this region (release_splitk_region: has the owner launched since its sync?)
};
struct Arena {
    Region r[kMaxRegions];
    bool   need_spare = true;
};
std::mutex g_mutex;
Arena      g_arena[64];

hipError_t alloc_region(Region& r);

int region_cap()
{
    static const int cap = [] {
        const char* e = getenv("EETQ_AMD_SPLITK_REGIONS");
        if (!e || !*e) return kMaxRegions;
        const int v = atoi(e);
        return v < 0 ? 0 : (v > kMaxRegions ? kMaxRegions : v);
    }();
    return cap;
}

hipError_t alloc_region(Region& r)
{
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&r.base), kRegionBytes);
    if (e != hipSuccess) {
        r.base = nullptr;
        return e;
    }
    // tickets start at 0 and only ever grow by S per tile
    e = hipMemset(r.base + kRegionSlabs, 0, kRegionBytes - kRegionSlabs);
    return e;
}

// EETQ_OK with the stream's own region, or EETQ_ERR_UNSUPPORTED (no message) when it cannot have one right now
int region_for(hipStream_t stream, float** slabs, unsigned** tickets)
{
    int dev = 0;
    EETQ_TRY_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mutex);
    Arena&    a   = g_arena[dev & 63];
    const int cap = region_cap();
    Region*   reg = nullptr;
    for (int i = 0; i < cap && !reg; ++i)
        if (a.r[i].used && a.r[i].owner == stream) reg = &a.r[i];
    if (!reg) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        // a capturing stream can only TAKE a region that already exists (creating one is not capturable): every eager
        // launch below leaves one spare region allocated for exactly that (torch captures on a stream of its own, which
        // has never launched anything eagerly)
        for (int i = 0; i < cap && !reg; ++i)
            if (!a.r[i].used && (a.r[i].base || !capturing)) reg = &a.r[i];
        if (!reg && capturing) return EETQ_ERR_UNSUPPORTED;
        // every region taken: this stream runs unsplit.  Nothing is inferred about the owners (a stream that is capturing
        // on another thread, or a destroyed one whose captured graphs are still replayed, must keep its region):
        // regions come back only through eetq_release_stream_workspace / eetq_release_workspace.
        if (!reg) {
            static std::atomic<bool> warned{false};
            if (!warned.exchange(true))  // once per process: the fallback is correct but slower, and silent otherwise
                fprintf(stderr,
                        "[eetq_amd] all %d split-K scratch regions of device %d are owned by other streams: split-K launches (medium-batch GEMMs, "
                        "9 <= M <= 128 on some shapes, and K-sliced tiled launches) on further streams run unsplit.  Release a stream's region with "
                        "eetq_release_stream_workspace (eetq_amd.ops.release_stream_workspace) before destroying the stream, or raise "
                        "EETQ_AMD_SPLITK_REGIONS.\n", cap, dev);
            return EETQ_ERR_UNSUPPORTED;
        }
        if (!reg->base) EETQ_TRY_HIP(alloc_region(*reg));
        reg->used    = true;
        reg->owner   = stream;
        a.need_spare = true;  // the spare (if this was it) is gone
    }
    if (a.need_spare) {  // keep one free region allocated (never created during capture) so that a graph capture can split too
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (!(hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)) {
            bool    spare = false;
            Region* empty = nullptr;
            for (int i = 0; i < cap; ++i) {
                if (!a.r[i].used && a.r[i].base) spare = true;
                if (!a.r[i].used && !a.r[i].base && !empty) empty = &a.r[i];
            }
            if (!spare && empty) EETQ_TRY_HIP(alloc_region(*empty));
            a.need_spare = false;
        }
    }
    ++reg->uses;
    *slabs   = reinterpret_cast<float*>(reg->base);
    *tickets = reinterpret_cast<unsigned*>(reg->base + kRegionSlabs);
    return EETQ_OK;
}

template <int MT, int NB, int SA, int SB, bool KFULL, int W = 4, int BITS = 8>
int launch_full(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K, int S,
                hipStream_t stream)
{
    const int R = (M + 32 * MT - 1) / (32 * MT);  // row groups of 32*MT rows (1 unless the plan cut the batch along M)
    using C   = gemm_splitk::Cfg<MT, NB, SA, SB, W, BITS>;
    auto kern = gemm_splitk::gemm_splitk_kernel<MT, NB, SA, SB, KFULL, W, false, BITS>;
    if (C::kSmem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    const int tiles_n = (N + C::kBN - 1) / C::kBN;
    const int tiles   = tiles_n * R;  // slab / ticket units: one per (row group, column tile)
    float*    slabs   = nullptr;
    unsigned* tickets = nullptr;
    if (S > 1) {
        if ((size_t)tiles > kRegionTickets || (size_t)tiles * S * C::kSlabFloats * 4 > kRegionSlabs) S = 1;
        else {
            int st = region_for(stream, &slabs, &tickets);
            if (st == EETQ_ERR_UNSUPPORTED) S = 1;  // no region of its own for this stream right now: run this launch unsplit
            else if (st != EETQ_OK) return st;
            else if (S == 4) tickets += kRegionTickets;
        }
    }
    // Occupancy of split launches.  Round 2 shipped two workgroups per CU for rings of <= 80 KiB and returned a few wrong
    // elements per call whenever more workgroups than fit were launched that way; round 3 first kept a second workgroup off the
    // CU (by asking for 84 KiB of LDS) and then found the cause: not the hand-over protocol but a write-after-read on the
    // slab stores' data registers that hipcc does not guard (gemm_splitk_kernel.hpp, "The stores' DATA registers stay live").
    // With that fixed both occupancies are exact (tools/deepk_check.py: 812 forced plans, 0 wrong either way;
    // tools/experiments/sk_debug.py) and equally fast (tools/splitk_occupancy.py), so a split launch asks for what it uses.
    const size_t lds = C::kSmem;
    // (grid.x a multiple of 8: see the block-id mapping in the kernel)
    launch_kernel(kern, dim3((tiles_n * S + 7) & ~7, R), dim3(C::kThreads), lds, stream, x, w, scales, y, M, N, K, S, slabs,
                  tickets, ep);
    return check_hip(hipGetLastError(), "gemm_splitk_kernel launch");
}

template <int MT, int NB, int SA, int SB, int W = 4, int BITS = 8>
int launch_inst(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K, int S,
                hipStream_t stream)
{
    return K % gemm_splitk::kBK == 0 ? launch_full<MT, NB, SA, SB, true, W, BITS>(x, w, scales, ep, y, M, N, K, S, stream)
                                     : launch_full<MT, NB, SA, SB, false, W, BITS>(x, w, scales, ep, y, M, N, K, S, stream);
}

// ring = 10 * SA + SB (activation / weight ring depths, gemm_splitk_kernel.hpp).  The library instantiates the shared rings
// 22 and 33 with four waves.  Measured in round 3 and NOT instantiated here (the kernel template and tools/kbench deepk
// have them): deeper weight rings (38, 36, 34, 26, 28, 24) are 0-10 % slower at every shape
// (profiles/r03_kbench_deepk.txt); eight waves per workgroup (two per k tile) are within -5 .. +8 %
// (profiles/r03_kbench_w8.txt; exact in all 1160 forced plans); DMA pieces interleaved with the MFMA groups, a 4 x 4 ring,
// 128-column blocks and two or three unsplit workgroups per CU do not help either (r03_kbench_inter / ring4 / bn128 /
// percu.txt).  DESIGN.md section 4.2b has the table and the reading.
template <int MT, int BITS = 8>
int launch_mt(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K, int nb, int S,
              int ring, hipStream_t stream)
{
#define EETQ_RING(NB_, SA_, SB_) \
    if (nb == NB_ && ring == 10 * SA_ + SB_) return launch_inst<MT, NB_, SA_, SB_, 4, BITS>(x, w, scales, ep, y, M, N, K, S, stream);
    EETQ_RING(1, 2, 2)
    EETQ_RING(2, 2, 2)
    if constexpr (MT <= 2) {
        EETQ_RING(1, 3, 3)
        EETQ_RING(2, 3, 3)
    }
#undef EETQ_RING
    return fail(EETQ_ERR_INVALID, "[eetq_amd] split-K: no instantiation for this (column blocks, ring) plan");
}

}  // namespace

// frees every split-K region of every device (eetq_release_workspace); the caller has synchronised the devices
int splitk_region(hipStream_t stream, float** slabs, size_t* slab_bytes, unsigned** tickets2, unsigned** tickets4, size_t* max_tiles)
{
    unsigned* t  = nullptr;
    int       st = region_for(stream, slabs, &t);
    if (st != EETQ_OK) return st;
    *slab_bytes = kRegionSlabs;
    *tickets2   = t;
    *tickets4   = t + kRegionTickets;
    *max_tiles  = kRegionTickets;
    return EETQ_OK;
}

// gives the region `stream` owns back to the pool (it stays allocated and becomes the spare a graph capture can take);
// the stream is synchronised first.  EETQ_OK whether or not the stream owned one.
int release_splitk_region(hipStream_t stream)
{
    int dev = 0;
    EETQ_TRY_HIP(hipGetDevice(&dev));
    // The stream is synchronised WITHOUT the lock (a blocked mutex would stall every other stream's launches); ownership and the
    // region's launch count are re-checked after re-acquiring it: a launch that slipped in between (another host thread using the
    // same stream) means the region is busy again, so synchronise once more instead of handing out a region with work in flight.
    for (int attempt = 0; attempt < 4; ++attempt) {
        unsigned long long seen = 0;
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            bool owns = false;
            for (Region& r : g_arena[dev & 63].r)
                if (r.used && r.owner == stream) {
                    owns = true;
                    seen += r.uses;
                }
            if (!owns) return EETQ_OK;
        }
        EETQ_TRY_HIP(hipStreamSynchronize(stream));
        std::lock_guard<std::mutex> lock(g_mutex);
        unsigned long long now = 0;
        for (Region& r : g_arena[dev & 63].r)
            if (r.used && r.owner == stream) now += r.uses;
        if (now != seen) continue;  // launched again while we were waiting: not idle yet
        for (Region& r : g_arena[dev & 63].r)
            if (r.used && r.owner == stream) {
                r.used  = false;
                r.owner = nullptr;
            }
        return EETQ_OK;
    }
    return fail(EETQ_ERR_INVALID, "[eetq_amd] eetq_release_stream_workspace: the stream keeps launching split-K GEMMs from another thread");
}

int release_splitk_workspace(size_t* freed)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    int keep = 0;
    (void)hipGetDevice(&keep);
    for (int d = 0; d < 64; ++d)
        for (Region& r : g_arena[d].r) {
            if (r.base) {
                if (hipSetDevice(d) == hipSuccess) {
                    (void)hipDeviceSynchronize();
                    (void)hipFree(r.base);
                }
                if (freed) *freed += kRegionBytes;
            }
            r = Region{};
        }
    for (int d = 0; d < 64; ++d) g_arena[d].need_spare = true;
    (void)hipSetDevice(keep);
    return EETQ_OK;
}

// Plan (column blocks per workgroup, K slices, ring depth) for a shape, from the measurements in
// profiles/r02_kbench_splitk.txt (MI355X, N = K = 4096 unless noted; round-1 unsplit tile in brackets):
//   M = 32: BN 32, S 2: 7.0 us [8.3];  M = 64: BN 32, S 2: 8.8 [11.2] (BN 64, S 4: 9.4);  M = 128: BN 64, S 4: 14.0 [20.2];
//   K = 11008, M = 64: BN 64, S 4: 16.3 [28.0] (BN 32, S 2: 24.5);  N = 11008, M = 64: BN 64, S 1: 17.1 [20.0].
// Reading: a slice costs about (its K steps) x (16*MT + 8*NB KiB per step at ~70 GB/s per CU) plus ~2 us for a 2-way and
// ~3 us for a 4-way in-launch reduction, on top of ~2.3 us of launch + first-data latency; wide column blocks pay off when
// the K loop is long (fewer re-reads of x) or the row tile is tall, narrow ones when the loop is short.  (Round 5 refitted the
// constants on every plan's time: the model in splitk_plan below.)
// Row groups instead of K slices (round 5): the batch is cut along M into r groups of <= 64 rows, 64-column blocks, all of K per
// workgroup -- (N / 64) * r workgroups, NO cross-workgroup reduction, the 3-deep ring (MT <= 2).  It applies when those
// workgroups fit the chip at once and the K-sliced tiled kernel would not slice (few tiles but a shallow K: tile_splitk_slices
// == 1); 32-row groups when even that leaves half the CUs idle.  Measured on two boxes (profiles/r05_splitk_rows_scan*.jsonl,
// r05_tilesplit_forms_scan*.jsonl; us, AUTO before -> this plan): 4096^2 M = 112 14.85 -> 12.35, M = 128 15.29 -> 12.42,
// M = 256 20.76 -> 17.04 (the tiled kernel K-sliced 2 x 32 steps: 17.2-17.6); 4096 x 6144 M = 128 19.98 -> 16.86; 5120^2
// M = 160 22.38 -> 20.93, M = 192 22.63 -> 21.26; 4096^2 M = 160 / 192 16.45 / 16.92 -> 16.37 / 16.41.  Where it loses it is
// not eligible: more than one round of workgroups (4096 x 6144 M = 160: 22.2 vs 18.7; 8192^2 M = 192: 45.8 vs 41.1) or a K the
// tiled kernel slices (8192^2 M = 128: 32.4 vs 30.6 K-sliced; 11008 x 4096 M = 128: 25.4 vs 23.7).
bool splitk_rows_plan(int M, int N, int K, int* r_out)
{
    // (M <= 128: splitk_plan's search covers row groups together with K slices and both column-block widths)
    if (M <= kMidMaxM || M > kSplitkMaxM || K % 64 != 0 || tile_splitk_slices(M, N, K) != 1 || wide_tile_splitk_slices(M, N, K) != 1)
        return false;
    const int ncu = device_cu_count(), tiles2 = (N + 63) / 64;
    int       r   = (M + 63) / 64;  // 64-row groups (MT = 2)
    if (tiles2 * r > ncu) return false;
    const int r32 = (M + 31) / 32;  // 32-row groups when the 64-row ones leave half the chip idle
    if (tiles2 * r * 2 <= ncu && tiles2 * r32 <= ncu) r = r32;
    if (r_out) *r_out = r;
    return true;
}

void splitk_plan(int M, int N, int K, int* nb_out, int* s_out, int* stages_out, int* r_out)
{
    {
        int r = 1;
        if (r_out && splitk_rows_plan(M, N, K, &r)) {  // callers that cannot run row groups (no r_out) keep the K-slice plans
            *nb_out     = 2;
            *s_out      = 1;
            *stages_out = 3;
            *r_out      = r;
            return;
        }
    }
    // K slices AND row groups (round 5, second half): for M > 32 the search also cuts the batch into r <= 4 groups of
    // 32 * ceil(M / 32r) rows -- fewer rows per workgroup (a lighter K step: 16 KiB of x per 32 rows), r times the workgroups, no
    // more hand-over.  One cost model for all of it (below); us before -> after the row groups joined the search: 5120^2 M = 96
    // 18.5 -> 14.2 (2,1,33 x 3 groups), 4096^2 M = 48 8.73 -> 8.3, M = 64 9.05 -> 8.7, M = 96 11.96 -> 10.9, 4096 x 6144 M = 96
    // 15.9 -> 15.0, 4096 x 2048 M = 128 10.8 -> 8.3.
    const int r_lo = (M + 127) / 128;  // a row group holds at most 128 rows (MT <= 4)
    const int r_hi = (r_out && M > 32) ? (M + 31) / 32 : r_lo;  // (<= 4; callers without r_out cannot run row groups)
    const int ncu   = device_cu_count();
    const int steps = (K / 64 + 3) / 4;
    double    best  = 1e30;
    int       bnb = 1, bs = 1, bst = 2, br = r_lo;
    for (int r = r_lo; r <= (r_hi > r_lo ? r_hi : r_lo); ++r) {
        const int MT = (M + 32 * r - 1) / (32 * r);
        if (MT > 4 || (M + 32 * MT - 1) / (32 * MT) != r) continue;  // (this r collapses to a smaller one)
        for (int nb = 1; nb <= 2; ++nb) {
            const int tiles = (N + 32 * nb - 1) / (32 * nb);
            for (int s = 1; s <= gemm_splitk::kMaxSlices; s *= 2) {
                if (s > 1 && steps / s < 2) continue;
                const int wgs = tiles * s * r;
                // ring depth 3 (120..144 KiB: one workgroup per CU) when that many workgroups fit anyway, else depth 2 (two per CU at MT <= 2)
                const int stages  = (MT <= 2 && wgs <= ncu) ? 3 : 2;
                // workgroups that can share a CU: plans without a hand-over on the 2-deep ring, as many as the 160 KiB of LDS hold,
                // three at most (split ones time the same at one and two per CU) ...
                const int lds_kib = stages * (16 * MT + 8 * nb);
                const int cap     = ((s == 1 || r > 1) && stages == 2 && MT <= 2 && lds_kib <= 80) ? (160 / lds_kib < 3 ? 160 / lds_kib : 3) : 1;
                const int rounds  = (wgs + ncu * cap - 1) / (ncu * cap);
                // ... and that DO share one in the fullest round
                const int per_cu  = rounds == 1 ? ((wgs + ncu - 1) / ncu < cap ? (wgs + ncu - 1) / ncu : cap) : cap;
                const int my_steps = (steps + s - 1) / s;
                // microseconds per K step: 0.27 per 32 rows of x + 0.14 per 32 columns of weights, x 1.22 / 1.6 when two / three
                // workgroups share the CU (they hide each other's latencies: not x 2 / x 3), the shallower ring 30 % slower; the
                // hand-over: publish + ticket + slab reads; 0.27 per extra row group (each weight tile is pulled out of L2 r
                // times).  The constants are a least-regret fit to the measured time of EVERY plan on 27 shapes x 6-7 batch sizes
                // (profiles/r05_splitk_plan_regret*.jsonl through tools/experiments/splitk_plan_fit.py; tests/test_abi.py holds the
                // planner against the same tables): over 664 points the pick is within 0.4 % of the best measured plan on average,
                // 16 points above 5 % (the round-2 constants: 1.7 %, 88).  Out of sample, before the last refit: a table measured
                // AFTER a fit on the others -- 22 shapes (5 new) at batch sizes none of them used -- scored 1.3 % / 12 of 154.
                double t = my_steps * (0.27 * MT + 0.14 * nb) * (per_cu == 3 ? 1.6 : per_cu == 2 ? 1.22 : 1.0) * (stages == 3 ? 1.0 : 1.3);
                t += s == 1 ? 0.0 : (s == 2 ? 2.2 : 2.8) + 0.088 * MT * nb * s;
                t = 1.45 + rounds * t + 0.27 * (r - 1);
                if (t < best) {
                    best = t;
                    bnb  = nb;
                    bs   = s;
                    bst  = stages;
                    br   = r;
                }
            }
        }
    }
    *nb_out     = bnb;
    *s_out      = bs;
    *stages_out = bst;
    if (r_out) *r_out = br;
}

int launch_gemm_splitk(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                       hipStream_t stream, int force_nb, int force_s, bool env_plan)
{
    if (M < 1 || M > kSplitkMaxM) return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] split-K tile path supports 1 <= M <= 1024");
    EETQ_REQUIRE((size_t)M * K * 2 < (1ull << 31) && (size_t)N * K < (1ull << 31),
                 "operand larger than 2 GiB is not supported by the buffer-addressed DMA path");
    int nb, s, stages, r = 1;
    splitk_plan(M, N, K, &nb, &s, &stages, &r);
    if (force_nb) nb = force_nb;
    if (force_s) s = force_s;
    if (force_nb || force_s) {
        r      = 1;
        stages = ((M + 31) / 32 <= 2 && ((N + 32 * nb - 1) / (32 * nb)) * s <= device_cu_count()) ? 3 : 2;
    }
    int ring = 11 * stages;  // shared ring of round 2: SA = SB = stages
    // EETQ_AMD_SPLITK_PLAN="nb,s,ring[,r]" (ring = 10 * SA + SB; r = row groups, default 1) overrides the plan of the FORCED path
    // only (EETQ_PATH_SPLITK: tuning and tests of every instantiation); production (AUTO) launches never read the environment
    if (const char* e = env_plan ? getenv("EETQ_AMD_SPLITK_PLAN") : nullptr) {
        int a = 0, b = 0, c = 0, d = 1;
        if (sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &d) >= 3) {
            nb   = a;
            s    = b;
            ring = c;
            r    = d;
        }
    }
    EETQ_REQUIRE((nb == 1 || nb == 2) && (s == 1 || s == 2 || s == 4) && r >= 1 && r <= 32, "invalid split-K plan");
    EETQ_REQUIRE((M + 32 * r - 1) / (32 * r) <= 4, "split-K plan: a row group holds at most 128 rows");
    // every K slice must own at least one 256-deep step: the kernel's prologue requests a slice's first stage without a test
    // (gemm_splitk_kernel.hpp).  The planner never cuts finer (steps / s >= 2); a FORCED plan on a shallow K could
    // (K = 320, S = 4: two steps, slices 0 and 2 empty) -- such a plan runs with as many slices as there are steps.
    while (s > 1 && (K / 64 + 3) / 4 < s) s >>= 1;
    // r row groups of 32 * MT rows each (launch_full derives the group count back from MT): MT = ceil(M / (32 r))
    switch ((M + 32 * r - 1) / (32 * r)) {
        case 1: return launch_mt<1>(x, w, scales, ep, y, M, N, K, nb, s, ring, stream);
        case 2: return launch_mt<2>(x, w, scales, ep, y, M, N, K, nb, s, ring, stream);
        case 3: return launch_mt<3>(x, w, scales, ep, y, M, N, K, nb, s, ring, stream);
        default: return launch_mt<4>(x, w, scales, ep, y, M, N, K, nb, s, ring, stream);
    }
}


// W4A16, 17 <= M <= 128 (round 4): the same tile on int4 weight tiles (gemm_splitk_kernel<..., BITS = 4>) instead of expanding
// the nibbles to int8 tiles first (one more pass over the weights and a scratch buffer that cannot be created while a graph
// is being captured; M = 64 was 2x the W8A16 time).  The plan is the int8 one: x traffic, not the weight stream, shapes it.
int launch_gemm_splitk_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                          hipStream_t stream, bool env_plan)
{
    if (M < 1 || M > kMidMaxM) return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] split-K tile path supports 1 <= M <= 128");
    EETQ_REQUIRE(K % 128 == 0 && N % 16 == 0, "W4A16 needs K % 128 == 0 and N % 16 == 0");
    EETQ_REQUIRE((size_t)M * K * 2 < (1ull << 31) && (size_t)N * K / 2 < (1ull << 31),
                 "operand larger than 2 GiB is not supported by the buffer-addressed DMA path");
    int nb, s, stages, r = 1;
    splitk_plan(M, N, K, &nb, &s, &stages, &r);  // (incl. the row-group plan: the int4 tile has the int8 tile's geometry)
    int ring = 11 * stages;
    if (const char* e = env_plan ? getenv("EETQ_AMD_SPLITK_PLAN") : nullptr) {
        int a = 0, b = 0, c = 0, d = 1;
        if (sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &d) >= 3) {
            nb   = a;
            s    = b;
            ring = c;
            r    = d;
        }
    }
    EETQ_REQUIRE((nb == 1 || nb == 2) && (s == 1 || s == 2 || s == 4) && r >= 1 && r <= 4, "invalid split-K plan");
    while (s > 1 && (K / 128 + 3) / 4 < s) s >>= 1;  // no empty K slice (see launch_gemm_splitk); an int4 tile is 128 deep
    switch ((M + 32 * r - 1) / (32 * r)) {
        case 1: return launch_mt<1, 4>(x, w, scales, ep, y, M, N, K, nb, s, ring, stream);
        case 2: return launch_mt<2, 4>(x, w, scales, ep, y, M, N, K, nb, s, ring, stream);
        case 3: return launch_mt<3, 4>(x, w, scales, ep, y, M, N, K, nb, s, ring, stream);
        default: return launch_mt<4, 4>(x, w, scales, ep, y, M, N, K, nb, s, ring, stream);
    }
}

}  // namespace eetq
