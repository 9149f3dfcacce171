// Kernel template of the split-K medium-batch MFMA dequant-GEMM (AUTO: 9 <= M <= 128 on some shapes, 64-row groups up to
// M = 1024; the forced path any M <= 1024; included by gemm_splitk.hip and tools/kbench.hip).
//
// Why split K.  At these M the time of gemm_mid_kernel is not the weight stream but what ONE compute unit can pull in
// through its vector memory path (~57 B/clk = ~120 GB/s): a workgroup that owns BN columns re-reads all of x (M x K fp16)
// next to its BN x K weight bytes, and only N / BN workgroups exist.  N = K = 4096, M = 64, BN = 32: 128 workgroups pull
// 512 + 128 KiB each (5.3 us at 120 GB/s) while half the chip idles.  The bytes every CU must ingest are
//     ((N / BN) * M * K * 2  +  N * K) / CUs in use,
// so the lever is wider column blocks (fewer re-reads of x) TOGETHER with enough workgroups to use every CU: BN = 64 and
// the K range cut into S slices gives (N / 64) * S workgroups that ingest (M * 2 + 64) * K / S bytes each -- 192 KiB at
// S = 4 for the shape above (1.6 us), below the time the 16 MiB weight stream takes.
//
// Structure (per workgroup, 4 waves): the medium-batch tile of gemm_mid_kernel.hpp with NB column blocks of 32 --
// (32*MT rows) x (32*NB columns) x 256-deep K step, wave j owns k tile j of the step for all rows and columns
// (v_mfma_f32_32x32x16_f16, weights as the A operand: 4*MT*NB MFMAs per wave per step, every dequantised fragment feeds
// MT MFMAs, every activation fragment NB); x and the weight tiles go L2/HBM -> LDS by LDS-DMA through a 2- or 3-deep ring,
// x XOR-swizzled through the source address; the four k quarters are added through LDS at the end.
//
// The cross-workgroup reduction is in-launch and deterministic (no float atomics; replicas stay bit-identical):
//   * every slice writes its fp32 partial tile (a "slab", 4*MT*NB KiB) with WRITE-THROUGH 16-byte stores
//     (buffer_store_dwordx4 ... sc1), every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier,
//     one lane takes a ticket with a relaxed agent-scope atomic add on the tile's counter;
//   * the slice that draws the last ticket reads ALL S slabs of the tile back (sc1 loads: served below the per-CU L1,
//     and the XCD's L2 cannot hold an older copy -- slab lines are only ever written in a launch before they are read, and
//     kernel boundaries invalidate) and adds them in slice order 0..S-1, whichever slice it is itself and whatever the
//     arrival order was: the sum is a function of the data alone.  Then the usual epilogue.
//   * counters are monotonic: a tile's counter grows by exactly S per launch, "last" is (old & (S-1)) == S-1; they are
//     zeroed once when the scratch buffer is created and never reset.
// A barrier-free variant (wave-private x ring filled by each wave's own LDS-DMA, weights straight to registers, hand-counted
// vmcnt, no s_barrier in the main loop) was built and measured in round 2: within +-5 % of this kernel at N = K = 4096 and
// slower at K = 11008 (19.1 vs 16.3 us) -- with four waves per CU the time goes to a chain of ~1 us phases (first data,
// per-step issue, cross-wave reduction, publish + ticket, slab read), not to the barriers.  profiles/r02_kbench_splitk.txt.
// This is the hand-off of cdna_hip_programming.md (section 5, "in-launch split-K reduction", write-through form): no
// placement or dispatch-order assumption; the block-id -> (tile, slice) map below only makes a tile's slices neighbours
// on one XCD when the dispatcher places block b on XCD b % 8 (a speed matter).
#pragma once
#include "common.hpp"
#include "gemm_kernel.hpp"
#include "gemv_kernel.hpp"  // gemv::dequant_dword_i4

namespace eetq {
namespace gemm_splitk {

constexpr int kBK      = 256;
constexpr int kThreads = 256;
constexpr int kMaxSlices = 4;

// SA / SB: depth of the activation ring and of the weight ring.  They are separate because the two streams are not alike:
// the weight tiles come from HBM (~2 us under load) and every byte is read once, so what bounds their rate is the bytes a
// CU has IN FLIGHT -- with one shared 2- or 3-deep ring that was 16-32 KiB per CU, i.e. ~2-3.5 TB/s over the chip, and the
// K loop ran at the DMA latency per step (round 3: N = K = 4096, M = 64: 9.3 us where stream + launch cost 4.9) -- while the
// activations are L2-resident (short latency) but four to eight times as many bytes per step.  SB > SA gives the weights a
// GEMV-like 48-96 KiB in flight per CU in what the activation ring leaves of the 160 KiB.  SA == SB is the round-2 kernel.
// W = waves per workgroup: 4 (wave j owns k tile j of a step) or 8 (two waves share a k tile, one 32-deep half each).  A step
// is instruction-issue bound per wave -- 8-12 DMA pieces, 8-12 fragment reads, 96 dequant VALU ops and 8-16 MFMAs in a row,
// one wave per SIMD -- so eight waves halve each wave's share and put two waves on every SIMD to cover each other.
// BITS = 4 (W4A16, round 4): the weight stage holds int4 tiles -- 1 KiB = 16 columns x 128 k, a lane's 16 bytes = 32 consecutive k
// of one column (DESIGN.md "int4 layout") -- so a 256-deep step is TWO k tiles per 16 columns and half the weight bytes; wave
// j still multiplies k 64j..64j+63 of the step: lane (fn, fh) reads k group 2*(j&1) + fh of tile j>>1 (ONE 16-byte read per
// column block instead of two) and dword d of it is the weight operand of MFMA k chunk d, against x at k = 64j + 32fh + 8d.
// Everything after the MFMAs (cross-wave reduction, slabs, tickets, epilogue) is the int8 kernel's.
template <int MT, int NB, int SA, int SB = SA, int W = 4, int BITS = 8>
struct Cfg {
    static_assert(W == 4 || W == 8, "4 or 8 waves");
    static_assert(BITS == 8 || (BITS == 4 && W == 4), "int4 tiles: the 4-wave form");
    static_assert(SB >= SA && SA >= 2 && (SA <= 3 || (SA == 4 && SB == 4)), "the wait accounting covers SA <= 3 (SB >= SA) and 4 x 4");
    static constexpr int kRows   = 32 * MT;
    static constexpr int kBN     = 32 * NB;
    static constexpr int kABytes = kRows * kBK * 2;
    static constexpr int kBBytes = kBN * kBK * BITS / 8;
    static constexpr int kTilesPerStep = BITS == 8 ? 4 : 2;              // 1 KiB weight tiles per 16 columns and K step
    static constexpr int kARing  = SA * kABytes;                         // activation stages first, then the weight stages
    static constexpr int kThreads = W * 64;
    static constexpr int kRed    = W * MT * NB * 16 * 64 * 4;           // end-of-kernel cross-wave reduction area
    static constexpr int kSmem   = SA * kABytes + SB * kBBytes;
    static_assert(kSmem >= kRed + 16, "the reduction area and the ticket word must fit the rings");
    static_assert(kSmem <= 160 * 1024, "rings larger than the CU's LDS");
    static constexpr int kAPW    = kABytes / 1024 / W;                   // A pieces (2 rows of 512 B) per wave and stage
    static constexpr int kBPW    = kBBytes / 1024 / W;                   // B pieces (native 1 KiB tiles) per wave and stage
    static_assert(kAPW * 1024 * W == kABytes && kBPW * 1024 * W == kBBytes, "pieces must divide evenly among the waves");
    static constexpr int kPieces = kAPW + kBPW;
    static_assert((SA - 1) * kAPW + (SB - 1) * kBPW <= 63, "vmcnt is a 6-bit counter");
    static constexpr int kSlabFloats = kRows * kBN;                      // fp32 partial tile of one slice
};

// grid = (tiles_n * S, R) workgroups: R row groups of 32*MT rows each (R = 1: all of M in one row tile, M <= 32*MT -- rounds 2-4;
// R > 1, round 5: the batch is cut along M instead of -- or on top of -- K, so a launch fills the chip WITHOUT a cross-workgroup
// reduction: M = 64 at N = 4096 is 2 row groups x 128 column tiles = 256 workgroups of 32 rows x all of K; the price is that
// each weight tile is pulled out of L2 by R workgroups, which the block-id map below makes neighbours on one XCD).
// slabs: [R * tiles_n][S][kSlabFloats] floats, counters: [R * tiles_n] unsigned (both unused when S == 1).
// INTER: the DMA pieces of the stage being refilled are issued BETWEEN the MFMA groups of the step (a few right after the
// fragment reads, to cover their latency) instead of all of them before the first dequant.
template <int MT, int NB, int SA, int SB, bool KFULL, int W = 4, bool INTER = false, int BITS = 8>
__global__ __launch_bounds__(W * 64, (W == 4 && Cfg<MT, NB, SA, SB, W, BITS>::kSmem <= 80 * 1024 && MT * NB <= 4) ? 2 : 1) void gemm_splitk_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales, f16* __restrict__ y, int M,
    int N, int K, int S, float* __restrict__ slabs, unsigned* __restrict__ counters, Epilogue ep)
{
    using C = Cfg<MT, NB, SA, SB, W, BITS>;
    constexpr int TPS = C::kTilesPerStep;  // weight tiles per 16 columns and step
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lds0 = (int)(uint32_t)(uintptr_t)(gemm::lds_void*)smem;
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int ktw  = wave & 3;   // k tile of a step this wave works on
    const int half = wave >> 2;  // W == 8: which 32-deep half of that k tile (W == 4: both)
    const int KT   = K >> 6;                      // 64-deep k tiles (what a wave multiplies per step)
    const int KTW  = BITS == 8 ? KT : (K >> 7);  // 1 KiB weight tiles per 16 columns along K
    const int steps_total = (KT + 3) >> 2;

    // ---- block id -> (column tile, K slice): a tile's slices are consecutive ids on one XCD when tiles_n % 8 == 0 ----
    const int tiles_n = (N + C::kBN - 1) / C::kBN;
    // (shifts and masks only: S is 1, 2 or 4 and the row group is blockIdx.y.  The divisions by run-time S and R that stood here --
    // two of them 64-bit -- were ~330 dependent scalar instructions ahead of the first DMA request, 0.7-1 us of every launch:
    // gemm_mid_kernel, the same tile without them, ran 8.9 us where this kernel's unsplit plan ran 10.1, 4096 x 6144, M = 24)
    const int sh = S == 4 ? 2 : (S == 2 ? 1 : 0);
    const int rg = blockIdx.y;  // row group (gridDim.y = R); the groups of a column tile share block id % 8, i.e. an XCD
    int       tile, slice;
    if ((tiles_n & 7) == 0) {
        // ... and every XCD takes ONE CONTIGUOUS eighth of the column tiles, i.e. of the weight (round 5; rounds 2-4 dealt the tiles
        // round-robin, tile % 8 = XCD): same box, same binary otherwise (tools/experiments/ab_lib.py), 8192^2 M = 17 / 32 / 64
        // 19.5 / 18.1 / 22.7 -> 16.1 / 17.0 / 21.4 us, 8192 x 10240 M = 64 30.5 -> 28.6, 4096 x 12288 -3 %, within +-1 % elsewhere.
        // (The GEMV kernel is the other way round: round-robin tile rows are 3 % ahead of contiguous ones there.)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slice = j & (S - 1);
        tile  = xcd * (tiles_n >> 3) + (j >> sh);
    } else {
        // the same for any tile count (the tiled kernel's formula): tiles_n * S virtual tiles, XCD x takes q or q + 1 consecutive
        // ones (N = 11008 is 172 column tiles; round-robin until the end of round 5: 4096 x 11008 M = 64 17.2 -> 16.8 us,
        // 8192 x 11008 M = 64 30.2 -> 28.6, profiles/r05_ab_xcd_contiguous_any.jsonl)
        // gridDim.x is padded to a multiple of 8 (the launcher): the hardware deals workgroups by LINEAR id x + gridDim.x * y, so
        // only then do the row groups y > 0 of a tile land on the XCD its group 0 is on (round-5 ADVICE: N = 11008 with R > 1 had
        // them rotated across XCDs and lost the shared-L2 weight re-reads the planner prices); the <= 7 surplus ids leave here
        const int TT = tiles_n << sh, q = TT >> 3, r = TT & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        if (idx >= (xcd < r ? q + 1 : q)) return;
        const int vt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        slice = vt & (S - 1);
        tile  = vt >> sh;
    }
    const int n0 = tile * C::kBN;
    // row group rg: rows [rg * kRows, min(M, (rg + 1) * kRows)) of the batch -- from here on the kernel sees its own rows only
    {
        const int m_first = rg * C::kRows;
        x += (size_t)m_first * K;
        y += (size_t)m_first * N;
        if (ep.residual) ep.residual += (size_t)m_first * N;
        M = M - m_first < C::kRows ? M - m_first : C::kRows;
    }
    const int vtile = rg * tiles_n + tile;  // slab / ticket index
    // K steps of this slice: [s0, s1); slices differ by at most one step
    const int s0 = (steps_total * slice) >> sh, s1 = (steps_total * (slice + 1)) >> sh;
    const int n_tiles_total = N >> 4;

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(w), 0, (int)((size_t)N * K * BITS / 8), 0x00020000);

    // ---- DMA pieces of this wave (roles fixed per index: no run-time branch per piece) ----
    //   i < kAPW : A piece p = wave*kAPW + i  -> rows 2p, 2p+1 (512 B each)
    //   else     : B piece b = wave*kBPW + (i - kAPW) -> 16-column tile b / TPS, weight tile b % TPS of the step
    int dma_voff[C::kPieces];
#pragma unroll
    for (int i = 0; i < C::kPieces; ++i) {
        if (i < C::kAPW) {
            const int p    = wave * C::kAPW + i;
            const int row  = 2 * p + (lane >> 5);
            const int slot = (lane & 31) ^ (row & 15);  // source slot for this LDS slot
            int       gm   = row;
            gm             = gm < M ? gm : M - 1;
            dma_voff[i]    = (gm * K + slot * 8) * 2;
        } else {
            const int b  = wave * C::kBPW + (i - C::kAPW);
            int       nt = (n0 >> 4) + b / TPS;
            nt           = nt < n_tiles_total ? nt : n_tiles_total - 1;
            dma_voff[i]  = (nt * KTW + b % TPS) * kTileBytes + lane * 16;  // + step*TPS tiles
        }
    }
    auto issue_a = [&](int buf, int step) {
        uint8_t* sa = smem + buf * C::kABytes;
#pragma unroll
        for (int i = 0; i < C::kAPW; ++i)
            // the last step of a K that is not a multiple of 256 reads past the row end into the next row (or is
            // zero-filled by the descriptor bounds check at the very end): those k tiles are never multiplied
            gemm::dma16(x_rsrc, dma_voff[i], step * kBK * 2, sa + (wave * C::kAPW + i) * 1024);
    };
    auto issue_b = [&](int buf, int step) {
        uint8_t* sb = smem + C::kARing + buf * C::kBBytes;
#pragma unroll
        for (int i = C::kAPW; i < C::kPieces; ++i) {
            const int b    = wave * C::kBPW + (i - C::kAPW);
            const int kt   = step * TPS + b % TPS;
            // clamp to the last valid weight tile (KFULL: every step is whole, nothing to clamp -- and nothing to compute per step)
            const int back = (KFULL || kt < KTW) ? 0 : (kt - (KTW - 1)) * kTileBytes;
            gemm::dma16(w_rsrc, dma_voff[i] - back, step * TPS * kTileBytes, sb + b * 1024);
        }
    };
    // one piece (index i of this wave's kPieces) of the stages refilled during `step`: A(step + SA - 1) / B(step + SB - 1)
    auto issue_piece = [&](int i, int bufa_next, int bufb_next, int step, bool do_a, bool do_b) {
        if (i < C::kAPW) {
            if (do_a)
                gemm::dma16(x_rsrc, dma_voff[i], (step + SA - 1) * kBK * 2, smem + bufa_next * C::kABytes + (wave * C::kAPW + i) * 1024);
        } else if (do_b) {
            const int b    = wave * C::kBPW + (i - C::kAPW);
            const int kt   = (step + SB - 1) * TPS + b % TPS;
            const int back = (KFULL || kt < KTW) ? 0 : (kt - (KTW - 1)) * kTileBytes;
            gemm::dma16(w_rsrc, dma_voff[i] - back, (step + SB - 1) * TPS * kTileBytes, smem + C::kARing + bufb_next * C::kBBytes + b * 1024);
        }
    };
    // s_waitcnt vmcnt(ya * kAPW + yb * kBPW): the immediate must be a constant, the pair is wave-uniform run-time data
    auto wait_younger = [&](int ya, int yb) {
#define EETQ_SPLITK_WAIT(A, B)                                                                     \
    case (A) * 4 + (B):                                                                            \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A) * C::kAPW + (B) * C::kBPW) : "memory");      \
        break;
        switch (ya * 4 + yb) {
            EETQ_SPLITK_WAIT(0, 0)
            EETQ_SPLITK_WAIT(0, 1)
            EETQ_SPLITK_WAIT(0, 2)
            EETQ_SPLITK_WAIT(1, 0)
            EETQ_SPLITK_WAIT(1, 1)
            EETQ_SPLITK_WAIT(1, 2)
            EETQ_SPLITK_WAIT(2, 2)
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
#undef EETQ_SPLITK_WAIT
    };

    // ---- fragment addressing: lane (fn, fh); this wave owns k tile `wave` of every step ----
    const int fn = lane & 31, fh = lane >> 5;
    // weight fragment of column block nb: 16-column tile (2*nb + (fn >> 4)), k tile `wave`
    constexpr int SN = W == 8 ? 1 : 2;  // 32-deep halves of the k tile this wave multiplies
    // int8: tile (fn >> 4) of the block's two, k tile ktw, k group fh + 2s.  int4: k tile ktw >> 1, k group 2*(ktw & 1) + fh
    const int b_off = BITS == 8 ? ((fn >> 4) * 4 + ktw) * 1024 + (fn & 15) * 16 + fh * 256 + (W == 8 ? half * 512 : 0)  // + nb*8192 + s*512
                                : ((fn >> 4) * 2 + (ktw >> 1)) * 1024 + (fn & 15) * 16 + (2 * (ktw & 1) + fh) * 256;    // + nb*4096
    const int a_key = fn & 15;
    int       a_slot[SN][2];
#pragma unroll
    for (int s = 0; s < SN; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e)  // int8: k = 64 ktw + 32 s + 16 fh + 8 e; int4 (chunk d = 2s + e): k = 64 ktw + 32 fh + 8 d
            a_slot[s][e] = ((BITS == 8 ? 8 * ktw + 4 * (W == 8 ? half : s) + 2 * fh + e : 8 * ktw + 4 * fh + 2 * s + e) ^ a_key) << 4;
    const int a_row_off = fn * 512;

    f16x2 scale2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int ncol = n0 + 32 * nb + fn;
        const f16 sc   = scales[ncol < N ? ncol : N - 1];
        scale2[nb]     = f16x2{sc, sc};
    }

    f32x16 acc[MT][NB];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nb][i] = 0.f;

    // Issue order, kept by the prologue and by every step i (relative to s0): A(i + SA - 1) then B(i + SB - 1).  The prologue
    // plays the "virtual" steps -(SB-1) .. -1.
#pragma unroll
    for (int v = -(SB - 1); v < 0; ++v) {
        // (the first virtual step asks for step s0 itself: s0 < s1 always, so no test and no branch -- the compiler can then COUNT
        // the requests between the scale loads and the wait below)
        if (v + SA - 1 >= 0 && (v == -(SB - 1) || s0 + v + SA - 1 < s1)) issue_a((v + SA - 1) % SA, s0 + v + SA - 1);
        if (v == -(SB - 1) || s0 + v + SB - 1 < s1) issue_b((v + SB - 1) % SB, s0 + v + SB - 1);
        if (v == -(SB - 1)) {
            // The scales must be in their registers before the K loop (a wait inside it would be one the compiler inserts, and it
            // cannot count the ring), but not before the FIRST DMA requests: the scale loads were issued ahead of them and return
            // ahead of them, so their latency runs under the first stage's instead of in front of it.  Round 5; before, every launch
            // waited for its scales -- a cold HBM read in a real model, where every layer has its own -- and only then asked for
            // its first tile: tools/experiments/ab_lib.py with one scale vector per weight set, -0.1 .. -0.35 us per launch
            // (4096^2 M = 32 7.45 -> 7.20, M = 128 11.51 -> 11.16); unchanged when the scales are L2-resident.
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(scale2[nb]));
        }
    }
    int bufa = 0, bufb = 0;
    int step = s0;
    // One K step.  STEADY (every step that still refills both rings): the wait is a CONSTANT and the DMA
    // issue unconditional -- the general form below walks a compare-and-branch chain to pick its s_waitcnt and tests do_a / do_b
    // every step, ~22 scalar instructions and up to seven branches in a step of ~600 cycles: against gemm_mid_kernel (the same
    // tile, one branch per step) that was +13 % wave cycles on the same decomposition (profiles/r05_pmc_medium.txt: 4.26 M vs
    // 3.77 M SQ_WAVE_CYCLES, 806 k vs 536 k scalar instructions; 10.9-11.4 vs 10.3-10.6 us per dispatch, 4096 x 6144, M = 24).
    auto k_step = [&](auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        // this wave's pieces of A(step) and B(step) have landed; everything issued after the later of the two may stay in
        // flight.  R = steps after this one.  SB > SA: the later one is A(step), issued one virtual step before B(step+SB-SA)
        // ...: younger = A(step+1 .. step+SA-2) and B(step+SB-SA .. step+SB-2), as far as they exist.  SB == SA: the later one
        // is B(step); younger = A and B of steps step+1 .. step+SA-2.
        if constexpr (STEADY) {  // R >= SB - 1: ya = SA - 2, yb = SA - 2 (shared rings) / SA - 1 (deeper weight ring)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SA - 2) * C::kAPW + (SB > SA ? SA - 1 : SA - 2) * C::kBPW) : "memory");
        } else {
            const int R  = s1 - 1 - step;
            const int ya = R < SA - 2 ? R : SA - 2;
            int       yb;
            if constexpr (SB > SA) {
                yb = R - (SB - SA) + 1;
                yb = yb < 0 ? 0 : (yb > SA - 1 ? SA - 1 : yb);
            } else {
                yb = ya;
            }
            wait_younger(ya, yb);
        }
        __builtin_amdgcn_s_barrier();  // ... everyone's have; everyone is done with the buffers refilled below
        const bool     active = KFULL || step * 4 + ktw < KT;  // wave-uniform: k tile beyond K on the last step
        const int sa = lds0 + bufa * C::kABytes;               // integer LDS addresses: see gemm::lds_read16
        const int sb = lds0 + C::kARing + bufb * C::kBBytes;
        // order inside a step as in gemm_mid_kernel: all fragment reads, then this wave's DMA pieces of the stage
        // kStages-1 steps ahead (they run under the LDS read latency), then dequant + MFMA
        constexpr int WR = BITS == 8 ? SN : 1;  // 16-byte weight reads per column block
        u32x4 wq[NB][WR];
        f16x8 xa[SN][2][MT];
        if (active) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int s = 0; s < WR; ++s) wq[nb][s] = gemm::lds_read16(sb + b_off + nb * (BITS == 8 ? 8192 : 4096) + s * 512);
#pragma unroll
            for (int s = 0; s < SN; ++s)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        xa[s][e][mt] = __builtin_bit_cast(f16x8, gemm::lds_read16(sa + mt * 32 * 512 + a_row_off + a_slot[s][e]));
        }
        __builtin_amdgcn_sched_barrier(0);
        const int  bufa_next = bufa == 0 ? SA - 1 : bufa - 1, bufb_next = bufb == 0 ? SB - 1 : bufb - 1;  // buffers of step - 1
        const bool do_a = STEADY || step + SA - 1 < s1, do_b = STEADY || step + SB - 1 < s1;
        constexpr int NC  = SN * NB * 2;                          // MFMA groups of a step (MT MFMAs each)
        constexpr int PPS = (C::kPieces + NC) / (NC + 1);         // DMA pieces per slot: slot 0 before the first group
        auto dma_slot = [&](int slot) {
#pragma unroll
            for (int j = 0; j < PPS; ++j)
                if (slot * PPS + j < C::kPieces) issue_piece(slot * PPS + j, bufa_next, bufb_next, step, do_a, do_b);
        };
        if constexpr (!INTER) {
            if (do_a) issue_a(bufa_next, step + SA - 1);
            if (do_b) issue_b(bufb_next, step + SB - 1);
        } else {
            dma_slot(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int s = 0; s < WR; ++s) asm volatile("" : "+v"(wq[nb][s]));  // reads stay above the dequant
#pragma unroll
            for (int s = 0; s < SN; ++s) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f16x2 wd[8];
                    if constexpr (BITS == 8) {
                        dequant_16(wq[nb][s], scale2[nb], wd);
                    } else {  // dwords 2s, 2s + 1 of the lane's 32 k: MFMA k chunks d = 2s + e
                        f16x2 lo[4], hi[4];
                        gemv::dequant_dword_i4(s == 0 ? wq[nb][0].x : wq[nb][0].z, scale2[nb], lo);
                        gemv::dequant_dword_i4(s == 0 ? wq[nb][0].y : wq[nb][0].w, scale2[nb], hi);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            wd[i]     = lo[i];
                            wd[4 + i] = hi[i];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f16x8 wf = gemm::make_frag(wd[4 * e], wd[4 * e + 1], wd[4 * e + 2], wd[4 * e + 3]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xa[s][e][mt], acc[mt][nb], 0, 0, 0);
                        if constexpr (INTER) {
                            __builtin_amdgcn_sched_barrier(0);
                            dma_slot(1 + (s * NB + nb) * 2 + e);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        } else if constexpr (INTER) {
#pragma unroll
            for (int c = 1; c <= NC; ++c) dma_slot(c);
        }
        bufa = bufa + 1 == SA ? 0 : bufa + 1;
        bufb = bufb + 1 == SB ? 0 : bufb + 1;
    };
    for (; step + SB - 1 < s1; ++step) k_step(std::true_type{});  // (SB >= SA: both rings are still being refilled)
    for (; step < s1; ++step) k_step(std::false_type{});

    // ---- add the W partial tiles through LDS; wave q < 4 then owns accumulator registers 4q..4q+3 of every block ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [wave][mt][nb][reg][lane]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wave * MT + mt) * NB + nb) * 16 + r) * 64 + lane] = acc[mt][nb][r];
    __syncthreads();
    if constexpr (W == 8) {
        if (wave >= 4) return;  // waves 4..7 are done: an ended wave no longer takes part in the workgroup's barriers
    }
    // s4[mt][nb][i] = partial y[32*mt + fn][n0 + 32*nb + 8*wave + 4*fh + i]
    float s4[MT][NB][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < W; ++q) s += red[(((q * MT + mt) * NB + nb) * 16 + 4 * wave + i) * 64 + lane];
                s4[mt][nb][i] = s;
            }

    if (S > 1) {
        // ---- publish this slice's partial tile (write-through), take a ticket ----
        const size_t tile_floats = (size_t)S * C::kSlabFloats;
        const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            slabs + (size_t)vtile * tile_floats, 0, (int)(tile_floats * 4), 0x00020000);
        // float4 index inside a slab: ((mt*NB + nb)*4 + wave)*64 + lane
        const int lane_off = (wave * 64 + lane) * 16;
        u32x4     pub[MT][NB];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                pub[mt][nb] = u32x4{__builtin_bit_cast(u32, s4[mt][nb][0]), __builtin_bit_cast(u32, s4[mt][nb][1]),
                                    __builtin_bit_cast(u32, s4[mt][nb][2]), __builtin_bit_cast(u32, s4[mt][nb][3])};
                __builtin_amdgcn_raw_buffer_store_b128(pub[mt][nb], s_rsrc, (mt * NB + nb) * 4096 + lane_off,
                                                       slice * C::kSlabFloats * 4, /*sc1*/ 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores
        // The stores' DATA registers stay live until the stores have completed.  hipcc does not guard a 16-byte buffer store
        // whose soffset is an SGPR against the next instruction overwriting its data registers (GCNHazardRecognizer: "this
        // hazard only exists if the instruction is not using a register in the soffset field") and reused the first data
        // register of one store for the address of the next: `buffer_store_dwordx4 v[6:9], v10, ..., s6 offen sc1` /
        // `v_add_u32 v6, 0x1000, v10`.  On gfx950 the store then sometimes published 0x1000 + lane * 16 instead of the partial
        // sum in the lanes it reads last (12..15 of every 16) -- whenever the memory pipeline was slow to take the store, i.e.
        // with two workgroups per CU and more workgroups than fit (found with tools/experiments/sk_debug.py: the published slab
        // holds the integers 4288, 4304, 4320, 4336).  Nothing may write these registers before the vmcnt(0) above.
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(pub[mt][nb]));
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem + C::kSmem - 16);  // inside the one dynamic LDS array
        if (tid == 0)
            *flag = __hip_atomic_fetch_add(counters + vtile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *flag;
        if ((ticket & (unsigned)(S - 1)) != (unsigned)(S - 1)) return;  // not the last slice of this tile
        // ---- last arriver: all S slabs, summed in slice order (its own one read back like the others) ----
        u32x4 part[kMaxSlices][MT][NB];
#pragma unroll
        for (int s = 0; s < kMaxSlices; ++s) {
            const int ss = s < S ? s : S - 1;  // clamped, predicated use: no load behind a branch
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    part[s][mt][nb] = __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, (mt * NB + nb) * 4096 + lane_off,
                                                                            ss * C::kSlabFloats * 4, /*sc1*/ 16);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = __builtin_bit_cast(float, (u32)part[0][mt][nb][i]);
#pragma unroll
                    for (int s = 1; s < kMaxSlices; ++s) {
                        const float v = __builtin_bit_cast(float, (u32)part[s][mt][nb][i]);
                        t             = s < S ? t + v : t;
                    }
                    s4[mt][nb][i] = t;
                }
    }

    // ---- epilogue ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 32 * mt + fn;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int ncol = n0 + 32 * nb + 8 * wave + 4 * fh;
            if (m < M && ncol < N) {
                f16x2 lo, hi;
                finish_quad(s4[mt][nb], ep, ncol, lo, hi);
                if (ep.residual) {
                    const u32x2 r = *reinterpret_cast<const u32x2*>(ep.residual + (size_t)m * N + ncol);
                    lo            = lo + as_f16x2(r.x);
                    hi            = hi + as_f16x2(r.y);
                }
                *reinterpret_cast<u32x2*>(y + (size_t)m * N + ncol) = u32x2{as_u32(lo), as_u32(hi)};
            }
        }
    }
}

}  // namespace gemm_splitk
}  // namespace eetq
