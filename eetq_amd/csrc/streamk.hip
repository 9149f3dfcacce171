// Small-batch (AUTO: 2 <= M <= 16; by explicit path up to 64 rows) W8A16 / W4A16 stream GEMM launcher; kernel in streamk_kernel.hpp.
// Covers the reference's batched-GEMV range (m <= 4, weightOnlyBatchedGemv/kernelLauncher.cu:165-192) and the
// small-M end of its CUTLASS range, where the weight stream -- not the matrix cores -- bounds the time.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "streamk_kernel.hpp"

namespace eetq {

namespace {

template <int MT, int NT, int WAVES, int D, int OCC, int BITS = 8, int XM = 0>
int launch_inst(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                hipStream_t stream)
{
    auto         kern = streamk::streamk_kernel<MT, NT, WAVES, D, OCC, BITS, XM>;
    const size_t smem = streamk::streamk_smem_bytes(MT, NT, WAVES) +
                        (XM == 1 ? (((size_t)M * K * 2 + 1023) & ~(size_t)1023)
                                 : XM >= 2 ? (size_t)WAVES * (2 * D - 1) * (XM == 5 ? 4096 : XM == 4 ? (BITS == 4 ? 4096 : 2048) : (BITS == 4 && XM == 2) ? 2048 : 1024) : 0);
    if (smem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    launch_kernel(kern, dim3(N / (kTileN * NT)), dim3(WAVES * 64), smem, stream, x, w, scales, y, M, N, K, ep);
    return check_hip(hipGetLastError(), "streamk_kernel launch");
}

// How a small-batch launch gets its activation fragments: form 0 = registers, 1 = block copy in LDS, 2 = per-wave ring in LDS
struct StreamPlan {
    int form, nt, waves;
};

// The block copy keeps M x K fp16 (rounded up to 1 KiB) in dynamic LDS NEXT TO the cross-wave reduction area (WAVES x NT KiB for one
// row tile): the launch fits only while the SUM stays within the CU's 160 KiB -- for the NT and WAVES the plan actually uses.  The
// launchers and stream_plan_query share this check; a plan that does not fit falls back to the register form (same bits).
inline bool block_copy_fits(int M, int K, int nt, int waves)
{
    const size_t xs = ((size_t)M * K * 2 + 1023) & ~(size_t)1023;
    return xs + streamk::streamk_smem_bytes(1, nt, waves) <= (size_t)kMaxDynamicLds;
}

inline StreamPlan plan_from_env(const char* name)
{
    StreamPlan  p{-1, 0, 0};
    const char* e = tuning_env(name);
    if (!e) return p;
    char form[16] = {0};
    int  nt = 0, waves = 0;
    if (sscanf(e, "%15[^,],%d,%d", form, &nt, &waves) >= 1) {
        p.form  = !strcmp(form, "regs") ? 0 : !strcmp(form, "block") ? 1 : !strcmp(form, "ring") ? 2 : -1;
        p.nt    = nt == 1 || nt == 2 ? nt : 0;
        p.waves = waves == 8 || waves == 16 ? waves : 0;
    }
    return p;
}

// The register form's plan.  int8: every workgroup re-reads the activations (M x K fp16 from L2) for its columns; with two tile
// rows per workgroup each activation fragment feeds two weight tiles.  Cost per k tile in wave-load units: NT weight loads + 2
// activation loads, times the workgroups the busiest CU gets.  N = 5120 (320 tile rows on 256 CUs): M = 8 10.0 -> 7.5 us, K = 13824
// 23.6 -> 17.2 us; N = 22016: 24.5 -> 17.0 us; N = 4096 stays at NT = 1 (5.0 vs 5.8 us) -- profiles/r01_kbench_streamk_nt.txt.
// Workgroup size (us with 16 / 8 waves): 5120^2 M = 4 7.59 / 7.08, 13824 x 5120 M = 4 17.03 / 15.26, 5120 x 13824 M = 8 15.69 / 14.67,
// 4096^2 M = 8 5.66 / 5.32; one tile row per workgroup on <= one workgroup per CU at M <= 4 keeps 16 waves (11008 x 4096 M = 2 9.66 / 10.27).
inline StreamPlan regs_plan_i8(int M, int N, int ncu)
{
    int nt = 1;
    if (N % (2 * kTileN) == 0) {
        const long cost1 = (long)((N / kTileN + ncu - 1) / ncu) * 3;
        const long cost2 = (long)((N / (2 * kTileN) + ncu - 1) / ncu) * 4;
        if (cost2 < cost1) nt = 2;
    }
    const int wgs = N / (kTileN * nt);
    return StreamPlan{0, nt, (M > 4 || nt == 2 || wgs > ncu) ? 8 : 16};
}

// int4 (per k tile: NT weight loads + 4 activation loads): 4096 x 11008 M = 4 8.99 -> 8.34 us with 8 waves, 5120 x 13824 10.93 ->
// 10.44, 11008 x 4096 8.46 -> 8.10; N = 5120: 160 two-row workgroups, 5.84 (16 waves) vs 6.29 us
inline StreamPlan regs_plan_i4(int N, int ncu)
{
    int nt = 1;
    if (N % (2 * kTileN) == 0) {
        const long cost1 = (long)((N / kTileN + ncu - 1) / ncu) * 5;
        const long cost2 = (long)((N / (2 * kTileN) + ncu - 1) / ncu) * 6;
        if (cost2 < cost1) nt = 2;
    }
    return StreamPlan{0, nt, N / (kTileN * nt) >= ncu ? 8 : 16};
}

// The rule (int8, one row tile, K / 64 >= 32), read off profiles/r04_stream_plan_sweep.txt (21 shapes x M = 2..8 x 7 plans; "rpc" =
// 16-column tile rows per CU) and checked against the register form of rounds 1-3 in the same run
// (profiles/r04_stream_plan_rule_check.txt, us per launch, registers -> rule; geometric mean over the grid 0.943, worst +1.1 %).
// nt0 / eight0: the register form's tile rows per workgroup and workgroup size.
//   rpc <= 1          ring, 1 row, 16 waves   4096^2 M = 4 5.21 -> 4.71, M = 8 5.33 -> 5.02; 11008 x 4096 M = 4 10.07 -> 8.97;
//                                             8192 x 1024 M = 4 7.04 -> 5.79, M = 8 7.51 -> 6.71; 4096 x 1024 M = 8 4.82 -> 4.21
//   1 < rpc <= 2      registers               (5120^2, 13824 x 5120, 6144^2, 7168^2, 28672 x 8192: nothing beats them by > 2 %)
//                     except N = 32 * CUs (rpc = 2) with K <= 8192 at M <= 5: ring, 1 row, 8 waves (8192^2 M = 4 12.73 -> 11.96)
//                     and M = 2 on rpc <= 1.5: block copy, 1 row (5120^2 7.07 -> 6.74)
//   2 < rpc <= 3      block copy, 1 row, while M*K*2 <= 24 KiB (4096 x 11008 M = 2 10.91 -> 8.94; 3072 x 9216 M = 2 8.43 -> 6.77),
//                     then ring, 1 row, 16 waves (4096 x 11008 M = 4 11.10 -> 9.28, M = 6 11.19 -> 9.65, M = 8 11.41 -> 10.22; 4096 x
//                     12288 M = 4 11.31 -> 9.84; the block copy falls off a cliff once its LDS leaves two workgroups per CU: M = 6 11.64)
//   3 < rpc <= 4      ring, 2 rows, 8 waves   5120 x 13824 M = 4 14.16 -> 13.36, M = 8 14.97 -> 14.21; 4096 x 14336 M = 8 11.96 -> 11.60
//   rpc > 4           ring, 1 row, 16 waves while M <= max(4, K / 1024 - 1), else 2 rows, 8 waves
//                                             8192 x 28672 M = 2 40.5 -> 35.0, M = 4 40.7 -> 35.9; 5120 x 27648 M = 4 24.8 -> 23.7,
//                                             M = 8 27.6 -> 25.7; 4096 x 22016 M = 8 16.77 -> 16.27
// 9 <= M <= 16 (profiles/r04_stream_plan_sweep_m9to16.txt; the ring holds 16 rows, two DMAs per tile):
//   rpc <= 1          ring, 1 row, 16 waves to M = 11, 8 waves from M = 12   4096^2 M = 16 6.38 -> 5.86; 8192 x 1024 M = 16 10.1 -> 8.44;
//                                                                            11008 x 4096 M = 16 13.9 -> 12.6
//   1 < rpc <= 2      ring, 2 rows, 8 waves from M = 12 (5120^2 M = 16 9.79 -> 8.97; 13824 x 5120 M = 16 21.4 -> 19.5), registers below
//   rpc > 2           ring, 2 rows, 8 waves    4096 x 11008 M = 12 12.33 -> 11.74, M = 16 13.45 -> 12.39; 5120 x 13824 M = 16 18.5 -> 16.5;
//                                              8192 x 28672 M = 16 51.5 -> 44.9
// All forms are bit-identical at equal wave count, so a wrong pick costs time only.
inline StreamPlan pick_plan(int M, int N, int K, int ncu, int nt0, bool eight0)
{
    static const bool lds_off = [] {
        const char* e = tuning_env("EETQ_AMD_I8_STREAM_XLDS");
        return e && atoi(e) == 0;
    }();
    StreamPlan p{0, nt0, eight0 ? 8 : 16};
    if (lds_off || M > 16) return p;
    const int  rows   = N / kTileN;
    const long xbytes = (long)M * K * 2;
    if (M > 8) {
        if (rows <= ncu) return StreamPlan{2, 1, M >= 12 ? 8 : 16};
        if (N % (2 * kTileN) != 0 || (rows <= 2 * ncu && M < 12)) return p;
        return StreamPlan{2, 2, 8};
    }
    if (rows <= ncu) return StreamPlan{2, 1, 16};
    if (rows <= 2 * ncu) {
        if (rows == 2 * ncu && K <= 8192 && M <= 5) return StreamPlan{2, 1, 8};
        // two rows on up to 1.5 tile rows per CU, K <= 6144: the block copy with one tile row per workgroup (5120^2 7.04 -> 6.73, 4096 x
        // 5120 6.05 -> 5.78, 4096 x 6144 6.35 -> 6.21; deeper K within +-1 %: 13824 x 5120 15.17 / 15.33; not at 1.75 rows per CU:
        // 7168^2 9.78 -> 10.16; not from M = 3)
        if (M == 2 && 2 * rows <= 3 * ncu && K <= 6144 && K % 128 == 0) return StreamPlan{1, 1, 8};
        return p;
    }
    if (rows <= 3 * ncu) return (K % 128 == 0 && xbytes <= 24 * 1024) ? StreamPlan{1, 1, 8} : StreamPlan{2, 1, 16};
    if (rows <= 4 * ncu) return N % (2 * kTileN) == 0 ? StreamPlan{2, 2, 8} : StreamPlan{2, 1, 16};
    // (round 5: at least up to M = 4 whatever K is -- held-out shapes, tools/experiments/stream_plan_sweep.py, profiles/r05_stream_plan_heldout.jsonl:
    // 3584 x 18944 M = 4 ring,1,16 12.69 vs ring,2,8 13.65 us; 4096 x 28672 M = 4 19.88 vs 20.50; costs 4096 x 22016 M = 4 16.02 vs 15.88)
    const int m_one_row = K / 1024 - 1 > 4 ? K / 1024 - 1 : 4;
    return (M <= m_one_row || N % (2 * kTileN) != 0) ? StreamPlan{2, 1, 16} : StreamPlan{2, 2, 8};
}

// W4A16 (K / 128 >= 32, i.e. K >= 4096), read off profiles/r04_stream_plan_sweep_i4.txt and checked against the register form in
// the same run (profiles/r04_stream_plan_rule_check_i4.txt, us per launch, registers -> rule; geometric mean 0.963 at M <= 8, worst
// +1.6 %).  The register form issues FOUR activation loads per weight load; the ring one DMA per tile at M <= 4, two at M <= 8.
//   rpc <= 1, M <= 8     ring, 1 row, 16 waves   11008 x 4096 M = 4 7.98 -> 6.92; 8192 x 1024 M = 4 6.07 -> 5.02; 4096^2 M = 8 4.73 -> 4.46
//   rpc <= 1, M >= 9     16-row ring (four DMAs per tile), 1 row, 8 waves   11008 x 4096 M = 16 13.35 -> 10.02; 8192 x 1024 9.90 -> 7.22;
//                        4096^2 M = 16 6.09 -> 5.23 (the block copy used before the ring existed: 5.32)
//   1 < rpc <= 2, M >= 9 16-row ring, 2 rows, 8 waves from M = 12 (from M = 9 at K >= 6144): 13824 x 5120 M = 16 19.07 -> 14.95; 8192^2
//                        12.14 -> 10.24; 5120^2 8.63 -> 7.67; 7168^2 11.5 -> 9.5; wider shapes stay on registers (ring 6-27 % slower at
//                        M = 9..12) except M >= 15 at K >= 8192 (8192 x 28672 M = 16 39.7 -> 35.3)
//   1 < rpc <= 2         ring, 2 rows, 16 waves above K = 8192 (13824 x 5120 M = 8 13.68 -> 12.32; 28672 x 8192 M = 4 25.5 -> 23.5)
//   2 < rpc <= 3, K 4096 block copy, 1 row, at M <= 5 (4096 x 11008 M = 2 8.36 -> 6.87, M = 5 8.48 -> 7.64; 4096 x 12288 M = 4 8.45 ->
//                        7.62), ring, 1 row, 8 waves above (M = 8 9.00 -> 8.68)
//   rpc > 4              ring, 1 row, 8 waves at M <= 4 from K = 5120 (5120 x 27648 M = 4 18.28 -> 16.57; 8192 x 28672 25.9 -> 24.5);
//                        ring, 2 rows, 8 waves at M >= 5 from K = 8192 (8192 x 28672 M = 8 27.9 -> 26.9)
// and registers everywhere else (5120^2, 5120 x 13824 / 15360, 4096 x 14336 / 22016, 6144^2 ...: nothing beats them by more than 2-3 %).
inline StreamPlan pick_plan_i4(int M, int N, int K, int ncu, int nt0, bool eight0)
{
    static const bool lds_off = [] {
        const char* e = tuning_env("EETQ_AMD_I4_STREAM_XLDS");
        return e && atoi(e) == 0;
    }();
    StreamPlan p{0, nt0, eight0 ? 8 : 16};
    if (lds_off) return p;
    const int  rows   = N / kTileN;
    const bool even   = N % (2 * kTileN) == 0;
    if (M > 8) {  // 16-row ring, 4 KiB slots, 8-wave workgroups (profiles/r04_stream_plan_sweep_i4_m9to16.txt)
        if (rows <= ncu) return StreamPlan{2, 1, 8};
        if (rows <= 2 * ncu) return (even && (M >= 12 || K >= 6144)) ? StreamPlan{2, 2, 8} : p;
        return (even && M >= 15 && K >= 8192) ? StreamPlan{2, 2, 8} : p;
    }
    if (rows <= ncu) return StreamPlan{2, 1, 16};
    if (rows <= 2 * ncu) return K > 8192 ? StreamPlan{2, even ? 2 : 1, 16} : p;
    if (rows <= 3 * ncu) return K > 4096 ? p : M <= 5 ? StreamPlan{1, 1, 8} : StreamPlan{2, 1, 8};
    if (rows <= 4 * ncu) return p;
    if (M <= 4 && K >= 5120) return StreamPlan{2, 1, 8};
    if (M >= 5 && K >= 8192 && even) return StreamPlan{2, 2, 8};
    return p;
}

// 17 <= M <= 32: which form the explicit stream path takes (AUTO sends these batch sizes to the split-K tile; abi.hip)
// Measured on the 14 seam shapes x M in {17, 24, 32} (tools/experiments/stream_ring32_seam.py -> profiles/r06_stream_ring32_seam.jsonl):
// the 32-row ring runs at 0.55 - 0.8 of the register form's time everywhere (4096^2 M = 32 10.1 -> 7.9 us, 5120 x 13824 M = 32
// 45 -> 22), one tile row per workgroup up to one workgroup per CU (N = 4096: 4096^2, 11008 x 4096, 14336 x 4096), two beyond.
// Against the split-K plans AUTO runs at these batch sizes it is ahead by >= 5 % on three points only, all at M = 17 (4096^2
// 7.05 -> 6.43, 4096 x 6144 8.67 -> 8.19, 8192^2 16.5 -> 15.0) and 15 - 37 % behind on the wide / deep shapes: AUTO keeps the
// split-K tile from M = 17 (verdict item 5: fewer than five shapes, shelved as an AUTO choice).
inline StreamPlan ring32_plan(int M, int N, int K, int ncu)
{
    (void)M, (void)K;
    return StreamPlan{2, N / kTileN <= ncu ? 1 : 2, 8};
}

template <int MT>
int launch_mt(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                hipStream_t stream)
{
    const int KT = K / kTileK;  // every wave must own >= D k tiles
    if (KT >= 32) {
        static const int forced_waves = [] {  // EETQ_AMD_I8_STREAM_WAVES=8 / 16: force the workgroup size (A/B runs)
            const char* e = tuning_env("EETQ_AMD_I8_STREAM_WAVES");
            return e ? atoi(e) : 0;
        }();
        if constexpr (MT == 1) {
            const int ncu = device_cu_count();
            // the register form's own plan (regs_plan_i8): two tile rows per workgroup where that puts fewer loads on the busiest
            // CU; 8-wave workgroups (round 4, profiles/r04_int8_stream_waves_ab.txt) wherever there are two tile rows per workgroup
            // or more workgroups than CUs, and at M > 4
            const StreamPlan regs  = regs_plan_i8(M, N, ncu);
            const int        nt    = regs.nt;
            const bool       eight = forced_waves ? forced_waves == 8 : regs.waves == 8;
            // Where the activation fragments come from (round 4; all three forms feed the same fragments to the same MFMAs in the
            // same order -- bit-identical results, tests/test_gpu_stream_xlds.py):
            //   regs  : straight from L2 into registers, 16 (clamped) rows x 128 B per weight tile: two vector loads of activations
            //           per vector load of weights, whatever M is
            //   block : the M rows copied ONCE per workgroup into LDS (LDS-DMA); needs K % 128 == 0 and M*K*2 <= 64 KiB; the per-tile
            //           cost is the weight load alone, so one tile row per workgroup is affordable where two load the CUs unevenly
            //   ring  : M <= 8: one LDS-DMA instruction per k tile and wave into the wave's own ring; any K, no copy up front
            // The rule: pick_plan above.  EETQ_AMD_I8_STREAM_PLAN=form,nt,waves (rule|regs|block|ring, 1|2, 8|16; "rule" / 0 = as
            // the rule says) overrides single fields for A/B runs and tests; EETQ_AMD_I8_STREAM_XLDS=0 switches both LDS forms off
            // (the register form with its own tile rows per workgroup and workgroup size: the kernel of rounds 1-3).
            StreamPlan plan = pick_plan(M, N, K, ncu, nt, eight);
            static const StreamPlan forced = plan_from_env("EETQ_AMD_I8_STREAM_PLAN");
            if (forced.form >= 0) plan.form = forced.form;
            if (forced.nt) plan.nt = forced.nt;
            if (forced.waves) plan.waves = forced.waves;
            if (forced_waves) plan.waves = forced_waves;
            if (plan.nt == 2 && N % (2 * kTileN) != 0) plan.nt = 1;
            if (plan.form == 1 && (K % 128 != 0 || (long)M * K * 2 > 128 * 1024 || !block_copy_fits(M, K, plan.nt, plan.waves))) plan.form = 0;
            if (plan.form == 2 && M > 16) plan.form = 0;
            const bool e8 = plan.waves == 8;
            if (plan.form == 2 && M > 8) {  // 16-row ring: two DMAs per tile
                if (plan.nt == 2)
                    return e8 ? launch_inst<MT, 2, 8, 2, 4, 8, 4>(x, w, scales, ep, y, M, N, K, stream)
                              : launch_inst<MT, 2, 16, 2, 4, 8, 4>(x, w, scales, ep, y, M, N, K, stream);
                return e8 ? launch_inst<MT, 1, 8, 2, 4, 8, 4>(x, w, scales, ep, y, M, N, K, stream)
                          : launch_inst<MT, 1, 16, 2, 4, 8, 4>(x, w, scales, ep, y, M, N, K, stream);
            }
            // (two tiles in flight per wave is the optimum of the ring forms too: one +4.8 %, three +6 %, four +9 % in geometric
            // mean over the sweep's shapes -- profiles/r04_stream_depth_ab.txt, r04_stream_depth1_ab.txt)
            if (plan.form == 2) {
                if (plan.nt == 2)
                    return e8 ? launch_inst<MT, 2, 8, 2, 4, 8, 2>(x, w, scales, ep, y, M, N, K, stream)
                              : launch_inst<MT, 2, 16, 2, 4, 8, 2>(x, w, scales, ep, y, M, N, K, stream);
                return e8 ? launch_inst<MT, 1, 8, 2, 4, 8, 2>(x, w, scales, ep, y, M, N, K, stream)
                          : launch_inst<MT, 1, 16, 2, 4, 8, 2>(x, w, scales, ep, y, M, N, K, stream);
            }
            if (plan.form == 1) {
                if (plan.nt == 2)
                    return e8 ? launch_inst<MT, 2, 8, 2, 4, 8, 1>(x, w, scales, ep, y, M, N, K, stream)
                              : launch_inst<MT, 2, 16, 2, 4, 8, 1>(x, w, scales, ep, y, M, N, K, stream);
                return e8 ? launch_inst<MT, 1, 8, 2, 4, 8, 1>(x, w, scales, ep, y, M, N, K, stream)
                          : launch_inst<MT, 1, 16, 2, 4, 8, 1>(x, w, scales, ep, y, M, N, K, stream);
            }
            if (plan.nt == 2)
                return e8 ? launch_inst<MT, 2, 8, 2, 4>(x, w, scales, ep, y, M, N, K, stream)
                          : launch_inst<MT, 2, 16, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
            return e8 ? launch_inst<MT, 1, 8, 2, 4>(x, w, scales, ep, y, M, N, K, stream)
                      : launch_inst<MT, 1, 16, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
        }
        if constexpr (MT == 2) {
            // 17 <= M <= 32 (round 6): the 32-row per-wave ring -- four LDS-DMAs per k tile and wave, two MFMA row tiles per weight
            // register tile, 8-wave workgroups (4 KiB slots: 16 waves would need 192 KiB of LDS) -- next to the register form
            // (16 clamped rows x 128 B of activations from L2 per row tile and weight tile).  Same fragments into the same MFMAs at
            // equal wave count.  EETQ_AMD_I8_STREAM_PLAN=ring|regs,nt,waves picks for A/B runs; the default: see the ring32 rule.
            static const StreamPlan forced = plan_from_env("EETQ_AMD_I8_STREAM_PLAN");
            const int ncu = device_cu_count();
            StreamPlan plan = ring32_plan(M, N, K, ncu);
            if (forced.form >= 0) plan.form = forced.form;
            if (forced.nt) plan.nt = forced.nt;
            if (forced.waves) plan.waves = forced.waves;
            if (plan.nt == 2 && N % (2 * kTileN) != 0) plan.nt = 1;
            if (plan.form == 2) {
                if (plan.nt == 2) return launch_inst<MT, 2, 8, 2, 2, 8, 5>(x, w, scales, ep, y, M, N, K, stream);
                return launch_inst<MT, 1, 8, 2, 2, 8, 5>(x, w, scales, ep, y, M, N, K, stream);
            }
            if (plan.waves == 8) {
                if (plan.nt == 2) return launch_inst<MT, 2, 8, 2, 2>(x, w, scales, ep, y, M, N, K, stream);
                return launch_inst<MT, 1, 8, 2, 2>(x, w, scales, ep, y, M, N, K, stream);
            }
        }
        return launch_inst<MT, 1, 16, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
    }
    if (KT >= 16) return launch_inst<MT, 1, 8, 2, 2>(x, w, scales, ep, y, M, N, K, stream);
    if (KT >= 4) return launch_inst<MT, 1, 4, 1, 1>(x, w, scales, ep, y, M, N, K, stream);
    return launch_inst<MT, 1, 1, 1, 1>(x, w, scales, ep, y, M, N, K, stream);
}

// W4A16, 2 <= M <= 16 (one row tile): int4 tiles carry 128 k, so a wave needs K / 128 >= WAVES * D tiles.  Two tile rows per
// workgroup when that puts fewer loads on the busiest CU (per k tile: NT weight loads + 4 activation loads).
int launch_mt_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K, hipStream_t stream)
{
    const int KT = K / 128;
    static const int forced_waves = [] {  // EETQ_AMD_I4_STREAM_WAVES=8 / 16: force the workgroup size (A/B runs)
        const char* e = tuning_env("EETQ_AMD_I4_STREAM_WAVES");
        return e ? atoi(e) : 0;
    }();
    if (KT >= 32) {
        const int ncu = device_cu_count();
        // the register form's own plan (regs_plan_i4): 8-wave workgroups when there are enough of them to give every CU one (round 4,
        // profiles/r04_int4_stream_waves_ab.txt), 16 waves when two tile rows per workgroup leave fewer workgroups than CUs
        const StreamPlan regs  = regs_plan_i4(N, ncu);
        const int        nt    = regs.nt;
        const bool       eight = forced_waves ? forced_waves == 8 : regs.waves == 8;
        // Where the activation fragments come from (see launch_mt / pick_plan): four activation loads per weight load in the
        // register form.  EETQ_AMD_I4_STREAM_PLAN=form,nt,waves overrides (A/B runs), EETQ_AMD_I4_STREAM_XLDS=0: registers only.
        StreamPlan plan = pick_plan_i4(M, N, K, ncu, nt, eight);
        static const StreamPlan forced = plan_from_env("EETQ_AMD_I4_STREAM_PLAN");
        if (forced.form >= 0) plan.form = forced.form;
        if (forced.nt) plan.nt = forced.nt;
        if (forced.waves) plan.waves = forced.waves;
        if (forced_waves) plan.waves = forced_waves;
        if (plan.nt == 2 && N % (2 * kTileN) != 0) plan.nt = 1;
        if (plan.form == 2 && M > 8) plan.waves = 8;
        if (plan.form == 1 && ((long)M * K * 2 > 144 * 1024 || !block_copy_fits(M, K, plan.nt, plan.waves))) plan.form = 0;
        const bool e8 = plan.waves == 8;
        if (plan.form == 2 && M > 8) {  // 16-row ring: 4 KiB slots, 8-wave workgroups only (16 waves would need 192 KiB of LDS)
            if (plan.nt == 2) return launch_inst<1, 2, 8, 2, 2, 4, 4>(x, w, scales, ep, y, M, N, K, stream);
            return launch_inst<1, 1, 8, 2, 2, 4, 4>(x, w, scales, ep, y, M, N, K, stream);
        }
        if (plan.form == 2 && M <= 4) {
            if (plan.nt == 2) return e8 ? launch_inst<1, 2, 8, 2, 2, 4, 3>(x, w, scales, ep, y, M, N, K, stream)
                                        : launch_inst<1, 2, 16, 2, 2, 4, 3>(x, w, scales, ep, y, M, N, K, stream);
            return e8 ? launch_inst<1, 1, 8, 2, 2, 4, 3>(x, w, scales, ep, y, M, N, K, stream)
                      : launch_inst<1, 1, 16, 2, 2, 4, 3>(x, w, scales, ep, y, M, N, K, stream);
        }
        if (plan.form == 2) {
            if (plan.nt == 2) return e8 ? launch_inst<1, 2, 8, 2, 2, 4, 2>(x, w, scales, ep, y, M, N, K, stream)
                                        : launch_inst<1, 2, 16, 2, 2, 4, 2>(x, w, scales, ep, y, M, N, K, stream);
            return e8 ? launch_inst<1, 1, 8, 2, 2, 4, 2>(x, w, scales, ep, y, M, N, K, stream)
                      : launch_inst<1, 1, 16, 2, 2, 4, 2>(x, w, scales, ep, y, M, N, K, stream);
        }
        if (plan.form == 1) {
            if (plan.nt == 2) return e8 ? launch_inst<1, 2, 8, 2, 2, 4, 1>(x, w, scales, ep, y, M, N, K, stream)
                                        : launch_inst<1, 2, 16, 2, 2, 4, 1>(x, w, scales, ep, y, M, N, K, stream);
            return e8 ? launch_inst<1, 1, 8, 2, 2, 4, 1>(x, w, scales, ep, y, M, N, K, stream)
                      : launch_inst<1, 1, 16, 2, 2, 4, 1>(x, w, scales, ep, y, M, N, K, stream);
        }
        if (plan.nt == 2) return e8 ? launch_inst<1, 2, 8, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream)
                                    : launch_inst<1, 2, 16, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
        return e8 ? launch_inst<1, 1, 8, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream)
                  : launch_inst<1, 1, 16, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
    }
    if (KT >= 16) return launch_inst<1, 1, 8, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
    if (KT >= 4) return launch_inst<1, 1, 4, 1, 1, 4>(x, w, scales, ep, y, M, N, K, stream);
    return launch_inst<1, 1, 1, 1, 1, 4>(x, w, scales, ep, y, M, N, K, stream);
}

}  // namespace

// Which plan a small-batch launch takes (eetq_diag_stream_plan): pure host arithmetic, the same functions the launchers call.
// Returns 0 and fills form (0 registers, 1 block copy, 2 ring), tile rows per workgroup, waves per workgroup, or -1 when the shape is
// outside the kernel (M > 16 rows of one tile, K not a multiple of the tile depth).  Environment overrides are NOT applied.
int stream_plan_query(int bits, int M, int N, int K, int ncu, int* form, int* nt, int* waves)
{
    const int tile_k = bits == 4 ? 128 : 64;
    if ((bits != 4 && bits != 8) || M < 1 || M > 16 || N < kTileN || N % kTileN || K < tile_k || K % tile_k || ncu < 1) return -1;
    const int  KT = K / tile_k;
    StreamPlan p{0, 1, 16};
    if (KT >= 32) {
        if (bits == 8) {
            const StreamPlan r = regs_plan_i8(M, N, ncu);
            p                  = pick_plan(M, N, K, ncu, r.nt, r.waves == 8);
            if (p.nt == 2 && N % (2 * kTileN) != 0) p.nt = 1;
            if (p.form == 1 && (K % 128 != 0 || (long)M * K * 2 > 128 * 1024 || !block_copy_fits(M, K, p.nt, p.waves))) p.form = 0;
        } else {
            const StreamPlan r = regs_plan_i4(N, ncu);
            p                  = pick_plan_i4(M, N, K, ncu, r.nt, r.waves == 8);
            if (p.form == 2 && M > 8) p.waves = 8;
            if (p.nt == 2 && N % (2 * kTileN) != 0) p.nt = 1;
            if (p.form == 1 && ((long)M * K * 2 > 144 * 1024 || !block_copy_fits(M, K, p.nt, p.waves))) p.form = 0;
        }
        if (p.nt == 2 && N % (2 * kTileN) != 0) p.nt = 1;
    } else {
        p.waves = KT >= 16 ? 8 : KT >= 4 ? 4 : 1;
    }
    *form  = p.form;
    *nt    = p.nt;
    *waves = p.waves;
    return 0;
}

int launch_streamk_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                      hipStream_t stream)
{
    if (M < 1 || M > 16 || K % 128)
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] W4A16 stream-MFMA path supports 1 <= M <= 16, K % 128 == 0");
    return launch_mt_i4(x, w, scales, ep, y, M, N, K, stream);
}

int launch_streamk(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                hipStream_t stream)
{
    if (M < 1 || M > kStreamMaxM)
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] stream-MFMA path supports 1 <= M <= 64");
    switch ((M + 15) / 16) {
        case 1: return launch_mt<1>(x, w, scales, ep, y, M, N, K, stream);
        case 2: return launch_mt<2>(x, w, scales, ep, y, M, N, K, stream);
        case 3: return launch_mt<3>(x, w, scales, ep, y, M, N, K, stream);
        default: return launch_mt<4>(x, w, scales, ep, y, M, N, K, stream);
    }
}

}  // namespace eetq
