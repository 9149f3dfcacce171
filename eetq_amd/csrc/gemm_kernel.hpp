// Kernel template of the LDS-tiled MFMA dequant-GEMM (included by gemm.hip and tools/kbench.hip).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace eetq {
namespace gemm {

// kbench-only instrumentation (tools/kbench_stamps gemmstamps): wave 0 of every workgroup records the 100 MHz device clock
// at kernel entry (0), first stage landed (1), end of the steady loop (2), end of the drain steps (3), accumulators of the
// second K half parked in LDS (4), output stored (5).
#ifdef EETQ_KBENCH_STAMPS
__device__ unsigned long long* g_gemm_stamps = nullptr;
#define EETQ_GEMM_STAMP(i)                                                                                  \
    do {                                                                                                    \
        unsigned long long t_;                                                                              \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
        if (g_gemm_stamps && threadIdx.x == 0) g_gemm_stamps[(size_t)blockIdx.x * 8 + (i)] = t_;            \
    } while (0)
#else
#define EETQ_GEMM_STAMP(i) do { } while (0)
#endif

constexpr int BM = 128, BK = 64;
constexpr int A_STAGE_BYTES = BM * BK * 2;  // 16 KiB of fp16 activations per K step
constexpr int STAGES        = 6;
constexpr int kMinKSteps    = STAGES - 1;   // the statically unrolled drain needs K/64 >= 5
// J = 32-column blocks per wave: J = 2 -> 128 x 128 tile (the MFMA-bound shape), J = 1 -> 128 x 64 tile (twice the
// workgroups: fills the chip at 129 <= M <= 512 and trims the last partial round of tiles on other shapes)
// CW = waves across the tile's columns (each owns 32*J of them); 2*CW waves per workgroup.  CW = 2: the round-1 geometry,
// one wave per SIMD.  (J, CW) = (1, 4): the 128 x 128 tile on EIGHT waves, two per SIMD -- half the accumulators and half the
// DMA pieces per wave, and a second wave on every SIMD to issue MFMAs while the first sits in an LDS-DMA issue or a barrier.
// RH = row halves: RH = 2 is the TALL tile, 256 x 128 on eight waves (round 6) -- waves 0..3 own rows 0..127 exactly as the four
// waves of the 128 x 128 tile do, waves 4..7 rows 128..255, both halves read the SAME weight stage (one weight DMA and 8 KiB of LDS
// per K step and 256 rows instead of two), 40 KiB stages in a 4-slot ring (all 160 KiB of LDS), the epilogue in two phases.
// RB = 32-row-block multiplier per wave: RB = 2 is the DEEP tile, 256 x 128 on FOUR waves -- a wave owns 256 rows x 64 columns of
// one K half (256 accumulators per lane, one wave per SIMD): every dequantised weight fragment feeds 8 MFMAs instead of 4, so the
// dequant work and the weight reads per MFMA halve as well as the weight DMA.  Same stages, ring and epilogue phases as the tall tile.
template <int J, int CW = 2, int RH = 1, int RB = 1>
struct TileCfg {
    static constexpr int BN            = 32 * J * CW;
    static constexpr int WAVES         = 2 * CW * RH;
    static constexpr int ROWS          = BM * RH * RB;
    static constexpr int RING          = RH * RB == 2 ? 4 : STAGES;
    static constexpr int A_BYTES       = A_STAGE_BYTES * RH * RB;
    static constexpr int B_STAGE_BYTES = BN * BK;  // BN/16 native 1 KiB tiles per K step
    static constexpr int STAGE_BYTES   = A_BYTES + B_STAGE_BYTES;
    static constexpr int SMEM_BYTES    = RING * STAGE_BYTES;  // 144 / 120 / 160 KiB (also covers the end-of-kernel reduction)
};

typedef __attribute__((address_space(3))) void lds_void;

// LDS-DMA: 16 B per lane from a buffer (base in the descriptor, per-lane byte offset in voff, wave-uniform byte
// offset in soff) straight into LDS at wave-uniform base + lane*16, no VGPR round trip.  The MUBUF form
// (buffer_load_dwordx4 ... lds) is used rather than global_load_lds: hipcc treats the latter as a FLAT access
// that may touch LDS and from then on degrades every counted "s_waitcnt lgkmcnt(N)" to lgkmcnt(0) while any DMA
// is in flight (SIInsertWaitcnts "pending flat"), which serialises the LDS fragment reads behind the MFMAs.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, uint8_t* lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// Same with an instruction offset IMM (<= 4095): the hardware adds it to the LDS address AND to the buffer offset, so
// pieces of one wave that are 1 KiB apart in LDS share one M0 value (their voff is pre-compensated by -IMM).
template <int IMM>
__device__ __forceinline__ void dma16_imm(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, uint8_t* lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, voff, soff, IMM, 0);
}

// 16-byte LDS read at an integer LDS address (the dynamic-LDS base is folded into the per-lane constants once)
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
__device__ __forceinline__ u32x4 lds_read16(int addr) { return *(lds_cu32x4*)(uintptr_t)(uint32_t)addr; }

__device__ __forceinline__ f16x8 make_frag(f16x2 a, f16x2 b, f16x2 c, f16x2 d)
{
    return f16x8{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}

// Workgroup tile 128 x (64*J) x 64, 4 waves = 2 K halves x 2 column halves: at J = 2 a wave owns 128 rows x 64 columns of
// one 32-deep half of every K step (16 x v_mfma_f32_32x32x16_f16 per step, 128 fp32 accumulators per lane, one wave per
// SIMD); at J = 1 it owns 32 columns (8 MFMAs per step; the slot tables below have a second, 8-gap form).  Every dequantised weight fragment feeds 4 MFMAs (row repeat), every activation fragment 2 (column repeat).
// The two K halves' partial sums are added once at the end through LDS.  6-stage LDS-DMA ring, one barrier per K step.
//
// Where each instruction sits (one wave per SIMD hides roughly five single-issue instructions under one 32-cycle MFMA,
// so the ~85 non-MFMA instructions of a K step are spread over its 16 MFMA gaps and pinned there with sched_barrier):
//   * gap 0: the two weight reads; gaps 1..9: the eight activation fragment reads of the NEXT step;
//   * gaps 4..15: the 48 int8->fp16 VALU ops as micro-ops, one kind (perm / -1152 / *scale) on four half-dwords per gap:
//     four independent ops per gap, dependent ops a gap apart;
//   * gaps 0,1,3,5,8,11: this wave's six LDS-DMA pieces of stage kt+5, spread over the step rather than bursting after
//     the barrier (the CU's one address/data path needs ~430 cycles for a stage's 24 pieces);
//   * every wave moves 4 activation + 2 weight pieces, so descriptors and source offsets are static per piece, and the
//     pieces of one kind share one M0 (the instruction offset advances the LDS address and the buffer offset together);
//   * gaps 12..15: ring offsets and LDS read addresses of the next step.
// An earlier schedule (all reads in gaps 0..3, DMA in gaps 0..5, dequant 6 ops per gap in gaps 4..11, bookkeeping after
// the last MFMA) ran 3 % slower; see profiles/r01_kbench_ablation_gemm.txt.
// ABLATE (kbench only): 1 = no DMA, 2 = no dequant, 4 = no LDS fragment reads, 8 = no MFMA, 16 = no barrier
// ACT: the activation epilogues (ep.act != 0) are a separate instantiation.  With them inlined behind a run-time test the
// kernel grew from 4.4 k to 21 k instructions (exp / tanh expanded for 128 accumulators) and the SAME main loop ran 11 %
// slower (M = 1024, N = K = 4096: 40.8 vs 36.7 us, tools/kbench gemm, same box) -- instruction fetch, not registers (both
// builds use 410).  The identity instantiation keeps the round-1 epilogue.
// SPLIT (gemm_tile_splitk_kernel below): S workgroups share a tile, each runs a contiguous part of the K steps; the partial
// tiles meet in `slabs` ([tile][slice][BM * BN] floats, accumulator order) and the workgroup that draws the last of the tile's
// S tickets adds them IN SLICE ORDER (replicas stay bit-identical) and runs the ordinary epilogue.  Same hand-over as
// gemm_splitk_kernel.hpp: write-through stores, every wave drains them, barrier, one relaxed agent-scope ticket.
// GLU (its own instantiation, identity rounding): the weight is in "glu8" column order (groups of 16 = 8 gate + the 8 matching up
// columns); the fp16 image of the tile is written out as silu_mul(gate, up) -- y is [M][N / 2] with row stride ldc, what
// eetq_silu_mul_glu8_f16 makes of the plain projection's output, without the [M][N] round trip through HBM.
template <int ABLATE, int J, bool ACT, int CW, bool SPLIT, bool GLU = false, int RH = 1, int RB = 1>
__device__ __forceinline__ void gemm_tile_body(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int M, int N, int K, int ldc, Epilogue ep, int S, float* __restrict__ slabs,
    unsigned* __restrict__ counters)
{
    // N = columns of THIS launch (w, scales, y, ep.* already point at its first column); ldc = row stride of y / residual
    EETQ_GEMM_STAMP(0);
    using Cfg = TileCfg<J, CW, RH, RB>;
    constexpr int BN = Cfg::BN, STAGE_BYTES = Cfg::STAGE_BYTES, SMEM_BYTES = Cfg::SMEM_BYTES, NW = Cfg::WAVES;
    constexpr int ST = Cfg::RING, BMT = Cfg::ROWS, A_BYTES = Cfg::A_BYTES;
    constexpr int APW = 16 * RH * RB / NW;    // activation DMA pieces (8 rows each) per wave and stage: 4 or 2; 8 in the deep tile
    constexpr int BPW = (BN / 16) / NW;       // weight tiles per wave and stage: J, or 1 in the tall tile
    constexpr int WN_COLS = 32 * J, PIECES = APW + BPW, NMFMA = 8 * J * RB;
    constexpr int MT = 4 * RB;                // 32-row blocks per wave
    static_assert(J == 1 || J == 2, "slot tables exist for J = 1 and J = 2");
    static_assert(CW == 2 || (CW == 4 && J == 1), "geometries with slot tables: 4 waves (J = 1, 2) and 8 waves (J = 1)");
    static_assert(RH == 1 || (RH == 2 && J == 2 && CW == 2 && !SPLIT), "the tall tile is 256 x 128, unsplit");
    static_assert(RB == 1 || (RB == 2 && RH == 1 && J == 2 && CW == 2 && !SPLIT && ABLATE == 0), "the deep tile is 256 x 128 on four waves, unsplit");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int rh   = wave / (2 * CW);         // which 128-row half of the tile (0 unless RH = 2)
    const int grp  = (wave % (2 * CW)) / CW;  // which 32-deep half of each K step
    const int wn   = wave % CW;               // which 32*J-column part of the tile
    const int KT   = K >> 6;

    const int tiles_m = (M + BMT - 1) / BMT;
    const int tiles_n = (N + BN - 1) / BN;
    const int T       = tiles_m * tiles_n;
    int       tile, slice = 0, k0 = 0, ksteps = KT;  // this workgroup's K steps: [k0, k0 + ksteps)
    {
        // SPLIT: T * S virtual tiles in the same XCD-grouped order, a tile's slices next to each other (one XCD, one L2)
        const int TT = SPLIT ? T * S : T;
        const int b = blockIdx.x, q = TT >> 3, r = TT & 7, xcd = b & 7, idx = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if constexpr (SPLIT) {  // S is 2 or 4: shifts, no run-time division (two 64-bit ones stood here until round 5: ~300
            const int sh = S == 4 ? 2 : 1;  // dependent scalar instructions ahead of the first DMA request)
            const int vt = tile;
            tile   = vt >> sh;
            slice  = vt & (S - 1);
            k0     = (KT * slice) >> sh;
            ksteps = ((KT * (slice + 1)) >> sh) - k0;
        }
    }
    // tile order: row tiles in chunks of kGroupM, inside a chunk column-major.  The 32 workgroups resident on an
    // XCD at a time are consecutive tiles = 4 row tiles x 8 column tiles: the footprint its L2 fetches over the fabric
    // per K step (4 x 16 KiB activations + 8 x 8 KiB weights) is the smallest any 32-tile set has -- 64 MB per launch
    // at M = 1024, N = K = 4096 instead of 84 MB for whole columns of row tiles.
    constexpr int kGroupM = 4;
    const int chunk   = tile / (kGroupM * tiles_n);
    const int in_ch   = tile - chunk * (kGroupM * tiles_n);
    const int ch_rows = tiles_m - chunk * kGroupM < kGroupM ? tiles_m - chunk * kGroupM : kGroupM;
    const int m0 = (chunk * kGroupM + in_ch % ch_rows) * BMT;
    const int n0 = (in_ch / ch_rows) * BN;

    // descriptors start kShift bytes below the operands: a piece's voff is pre-compensated by -(its instruction
    // offset), and must stay non-negative for the hardware's range check
    constexpr int kShift = 4096;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<uint8_t*>(const_cast<f16*>(x)) - kShift, 0, (int)((size_t)M * K * 2) + kShift, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(w) - kShift, 0, (int)((size_t)N * K) + kShift, 0x00020000);
    // DMA pieces of this wave: i < APW: activation piece APW*wave + i (rows 8p..8p+7, 128 B each); i >= APW: weight tile
    // J*wave + i - APW of the stage's BN/16
    int       dma_voff[PIECES];
    const int n_tiles_total = N >> 4;
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int p    = wave * APW + i;
        const int row  = p * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        int       gm   = m0 + row;
        gm             = gm < M ? gm : M - 1;
        dma_voff[i]    = (gm * K + slot * 8) * 2 + kShift - (i & 3) * 1024;  // the instruction offset: at most 3 KiB
    }
#pragma unroll
    for (int i = APW; i < PIECES; ++i) {
        int nt      = (n0 >> 4) + wave * BPW + (i - APW);
        nt          = nt < n_tiles_total ? nt : n_tiles_total - 1;
        dma_voff[i] = nt * KT * kTileBytes + lane * 16 + kShift - (i - APW) * 1024;
    }
    const int dma_lds_a = wave * APW * 1024;                  // + i * 1024
    const int dma_lds_b = A_BYTES + wave * BPW * 1024;        // + (i - APW) * 1024

    const int fn = lane & 31, fh = lane >> 5;
    const int a_key = (fn >> 1) & 7;
    // per-lane constants of the LDS fragment reads (added to the stage offset)
    const int lds0 = (int)(uint32_t)(uintptr_t)(lds_void*)smem;
    const int c_a0 = lds0 + rh * A_STAGE_BYTES + fn * 128 + (((4 * grp + 2 * fh + 0) ^ a_key) << 4);
    const int c_a1 = lds0 + rh * A_STAGE_BYTES + fn * 128 + (((4 * grp + 2 * fh + 1) ^ a_key) << 4);
    const int c_b0 = lds0 + A_BYTES + ((wn * WN_COLS + fn) >> 4) * 1024 + (fn & 15) * 16 + fh * 256 + grp * 512;
    const int c_b1 = c_b0 + 2048;

    f16x2 scale2[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int ncol = n0 + wn * WN_COLS + 32 * j + fn;
        const f16 sc   = scales[ncol < N ? ncol : N - 1];
        scale2[j]      = f16x2{sc, sc};
    }

    f32x16 acc[MT][J];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][j][i] = 0.f;

    struct Frags {
        u32x4 wq[J];
        f16x8 xa[2][4];
    };
    struct WFrag {
        f16x8 f[J][2];
    };

    // ring state of the step about to run (wave-uniform): rd = LDS offset of the stage whose fragments it reads
    // (stage kt+1), wr = LDS offset its DMA fills (stage kt+5), ka / kb = source offsets of that stage
    int rd = STAGE_BYTES, wr = (ST - 1) * STAGE_BYTES, ka = (k0 + ST - 1) * BK * 2, kb = (k0 + ST - 1) * kTileBytes;
    int ra0 = rd + c_a0, ra1 = rd + c_a1, rb0 = rd + c_b0, rb1 = rd + c_b1;

    auto dma_piece = [&](auto itag) {
        constexpr int i = decltype(itag)::value;
        if constexpr (i < APW)
            dma16_imm<(i & 3) * 1024>(x_rsrc, dma_voff[i], ka, smem + wr + dma_lds_a + (i >> 2) * 4096);
        else
            dma16_imm<(i - APW) * 1024>(w_rsrc, dma_voff[i], kb, smem + wr + dma_lds_b);
    };

    auto step = [&](const WFrag& wcur, const Frags& fcur, auto read_tag, Frags& fnext, WFrag& wnext, auto dma_tag) {
        constexpr bool READ = decltype(read_tag)::value;
        constexpr bool DMA  = decltype(dma_tag)::value;
        u32            wd[J][8];
        const f16x2    bias1152 = {(f16)1152.0f, (f16)1152.0f};
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) {
            const int e = i / (4 * J), j = (i / 4) % J, mt = i & 3;
            if constexpr (!(ABLATE & 8))
                acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur.f[j][e], fcur.xa[e][mt], acc[mt][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // the MFMA opens its gap: dependent VALU ops of adjacent gaps never abut
            if constexpr (READ && !(ABLATE & 4)) {
                if (i == 0) {
                    fnext.wq[0] = lds_read16(rb0);
                    if constexpr (J == 2) fnext.wq[1] = lds_read16(rb1);
                }
                // activation fragments q = 4e + mt.  J = 2: gap 1: q0; 2: q1,q2; 3: q3; 4: q4; 6: q5; 7: q6; 9: q7.
                // J = 1: two per gap in gaps 1..4.
                auto xa_read = [&](int q) {
                    fnext.xa[q >> 2][q & 3] =
                        __builtin_bit_cast(f16x8, lds_read16(((q >> 2) ? ra1 : ra0) + (q & 3) * 32 * 128));
                };
                if constexpr (J == 2) {
                    if (i == 1) xa_read(0);
                    if (i == 2) { xa_read(1); xa_read(2); }
                    if (i == 3) xa_read(3);
                    if (i == 4) xa_read(4);
                    if (i == 6) xa_read(5);
                    if (i == 7) xa_read(6);
                    if (i == 9) xa_read(7);
                } else {
                    if (i >= 1 && i <= 4) { xa_read(2 * i - 2); xa_read(2 * i - 1); }
                }
            }
            if constexpr (READ && !(ABLATE & 2)) {
                constexpr int kDq0 = J == 2 ? 4 : 2;  // first gap that carries dequant ops (6*J gaps of them follow)
                if (i >= kDq0) {
                    // dequant micro-ops: gap i works on column block jj = (i-kDq0)/6, dword pair dp = ((i-kDq0)/3)&1 and
                    // applies ONE kind of op (perm / -1152 / *scale) to its four half-dwords: four independent VALU
                    // ops per gap, dependent ops a gap apart (no hazard nops)
                    const int jj = (i - kDq0) / 6, dp = ((i - kDq0) / 3) & 1, kind = (i - kDq0) % 3;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int h = 4 * dp + u, d = h >> 1;
                        if (kind == 0) {
                            const u32 wdw = d == 0   ? fnext.wq[jj].x
                                            : d == 1 ? fnext.wq[jj].y
                                            : d == 2 ? fnext.wq[jj].z
                                                     : fnext.wq[jj].w;
                            wd[jj][h] = __builtin_amdgcn_perm(wdw, 0x64646464u, (h & 1) ? 0x00070005u : 0x00060004u);
                        } else if (kind == 1) {
                            wd[jj][h] = as_u32(as_f16x2(wd[jj][h]) - bias1152);
                        } else {
                            wd[jj][h] = as_u32(as_f16x2(wd[jj][h]) * scale2[jj]);
                        }
                    }
                    // pin the pure VALU ops to this gap (instruction selection would sink them to their uses); input-only
                    // operands: an asm with VGPR outputs costs a hazard pad (s_nop) before the next instruction
                    asm volatile("" ::"v"(wd[jj][4 * dp]), "v"(wd[jj][4 * dp + 1]), "v"(wd[jj][4 * dp + 2]),
                                 "v"(wd[jj][4 * dp + 3]));
                }
            }
            if constexpr (DMA && !(ABLATE & 1)) {
                if constexpr (J == 2) {
                    if (i == 0) dma_piece(std::integral_constant<int, 0>{});
                    if (i == 1) dma_piece(std::integral_constant<int, 1>{});
                    if (i == 3) dma_piece(std::integral_constant<int, 2>{});
                    if (i == 5) dma_piece(std::integral_constant<int, 3>{});
                    if (i == 8) dma_piece(std::integral_constant<int, 4>{});
                    if constexpr (PIECES > 5)
                        if (i == 11) dma_piece(std::integral_constant<int, 5>{});
                } else if constexpr (CW == 2) {
                    if (i == 0) dma_piece(std::integral_constant<int, 0>{});
                    if (i == 1) dma_piece(std::integral_constant<int, 1>{});
                    if (i == 2) dma_piece(std::integral_constant<int, 2>{});
                    if (i == 3) dma_piece(std::integral_constant<int, 3>{});
                    if (i == 5) dma_piece(std::integral_constant<int, 4>{});
                } else {  // eight waves: two activation pieces and one weight tile per wave
                    if (i == 0) dma_piece(std::integral_constant<int, 0>{});
                    if (i == 2) dma_piece(std::integral_constant<int, 1>{});
                    if (i == 5) dma_piece(std::integral_constant<int, 2>{});
                }
            }
            if constexpr (READ) {
                // ring state and read addresses of the next step, in the gaps that carry little else
                // (after the step's last fragment read / last DMA piece: J = 2 gaps 12..15, J = 1 gaps 5..7)
                if (i == (J == 2 ? 12 : 5)) {
                    rd = rd + STAGE_BYTES == SMEM_BYTES ? 0 : rd + STAGE_BYTES;
                    asm volatile("" : "+s"(rd));
                }
                if (i == (J == 2 ? 13 : 6)) {
                    wr = wr + STAGE_BYTES == SMEM_BYTES ? 0 : wr + STAGE_BYTES;
                    ka += BK * 2;
                    kb += kTileBytes;
                    asm volatile("" : "+s"(wr), "+s"(ka), "+s"(kb));
                }
                if (i == (J == 2 ? 14 : 6)) {
                    ra0 = rd + c_a0;
                    ra1 = rd + c_a1;
                    asm volatile("" : "+v"(ra0), "+v"(ra1));
                }
                if (i == (J == 2 ? 15 : 7)) {
                    rb0 = rd + c_b0;
                    rb1 = rd + c_b1;
                    asm volatile("" : "+v"(rb0), "+v"(rb1));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (READ && (ABLATE & 4)) fnext = fcur;
        if constexpr (READ && (ABLATE & 2)) wnext = wcur;
        if constexpr (READ && !(ABLATE & 2)) {
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                wnext.f[jj][0] = make_frag(as_f16x2(wd[jj][0]), as_f16x2(wd[jj][1]), as_f16x2(wd[jj][2]), as_f16x2(wd[jj][3]));
                wnext.f[jj][1] = make_frag(as_f16x2(wd[jj][4]), as_f16x2(wd[jj][5]), as_f16x2(wd[jj][6]), as_f16x2(wd[jj][7]));
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): free here, keeps hipcc's counted LDS waits exact
        }
    };

    // ---- the tall tile's K step (RH = 2).  Two waves per SIMD leave 256 registers each: the schedule above keeps two full fragment
    // sets next to 128 accumulators (1 373 spilled registers when compiled for eight waves: 1.2 - 1.3 x the 128 x 128 tile's time,
    // tools/experiments/tall_tile_ab.py).  Here only the WEIGHTS are double-buffered (raw + dequantised, as above); an activation
    // fragment is requested two MFMA pairs ahead of its use into a four-deep window -- pair p = (K half e = p / 4, row block mt = p % 4)
    // feeds the two column blocks back to back -- and the window carries the next step's first two pairs across the barrier.  Every
    // accumulator still adds its K halves in the order e = 0, 1: the same bits as the 128 x 128 tile.
    // The deep tile (RB = 2) runs the same K step on ONE wave per SIMD with 16 pairs, an eight-deep window and requests four pairs
    // ahead (128+ cycles of MFMA between request and use), 32 MFMA gaps: fragment requests in the even gaps, the dequant micro-ops in
    // the odd gaps 5..27, the wave's ten DMA pieces in gaps 1, 3, 4, 8, ..., 28, ring bookkeeping in gaps 29..31.
    constexpr int NP = 8 * RB, WIN = 4 * RB, DIST = 2 * RB;  // pairs per step, window depth, request distance
    f16x8 xw[WIN];
    int   ca0 = c_a0, ca1 = c_a1;  // fragment addresses of the CURRENT stage (ra0 / ra1: the next one's)
    auto  step_tall = [&](const WFrag& wcur, auto read_tag, Frags& fnext, WFrag& wnext, auto dma_tag) {
        constexpr bool READ = decltype(read_tag)::value;
        constexpr bool DMA  = decltype(dma_tag)::value;
        u32            wd[J][8];
        const f16x2    bias1152 = {(f16)1152.0f, (f16)1152.0f};
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) {
            const int p = i >> 1, j = i & 1, e = p / (NP / 2), mt = p % (NP / 2);
            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur.f[j][e], xw[p % WIN], acc[mt][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((i & 1) == 0) {  // the pair's first MFMA is issued: request pair p + DIST (its slot of the window was pair p + DIST - WIN's)
                const int q = p + DIST;
                if (q < NP) {
                    xw[q % WIN] = __builtin_bit_cast(f16x8, lds_read16((q >= NP / 2 ? ca1 : ca0) + (q % (NP / 2)) * 32 * 128));
                } else if constexpr (READ) {
                    xw[q % WIN] = __builtin_bit_cast(f16x8, lds_read16(ra0 + (q - NP) * 32 * 128));
                }
            }
            if constexpr (READ) {
                if (i == 0) {
                    fnext.wq[0] = lds_read16(rb0);
                    fnext.wq[1] = lds_read16(rb1);
                }
                // dequant micro-ops, as in the 128 x 128 schedule: one kind of op on four half-dwords per gap, dependent ops a gap
                // (deep tile: two gaps) apart; g = which of the 12 groups this gap carries, -1 = none
                const int g = RB == 2 ? ((i >= 5 && i <= 27 && (i & 1)) ? (i - 5) / 2 : -1) : (i >= 4 ? i - 4 : -1);
                if (g >= 0) {
                    const int jj = g / 6, dp = (g / 3) & 1, kind = g % 3;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int h = 4 * dp + u, d = h >> 1;
                        if (kind == 0) {
                            const u32 wdw = d == 0 ? fnext.wq[jj].x : d == 1 ? fnext.wq[jj].y : d == 2 ? fnext.wq[jj].z : fnext.wq[jj].w;
                            wd[jj][h] = __builtin_amdgcn_perm(wdw, 0x64646464u, (h & 1) ? 0x00070005u : 0x00060004u);
                        } else if (kind == 1) {
                            wd[jj][h] = as_u32(as_f16x2(wd[jj][h]) - bias1152);
                        } else {
                            wd[jj][h] = as_u32(as_f16x2(wd[jj][h]) * scale2[jj]);
                        }
                    }
                    asm volatile("" ::"v"(wd[jj][4 * dp]), "v"(wd[jj][4 * dp + 1]), "v"(wd[jj][4 * dp + 2]), "v"(wd[jj][4 * dp + 3]));
                }
            }
            if constexpr (DMA) {
                if constexpr (RB == 2) {
                    if (i == 1) dma_piece(std::integral_constant<int, 0>{});
                    if (i == 3) dma_piece(std::integral_constant<int, 1>{});
                    if (i == 4) dma_piece(std::integral_constant<int, 2>{});
                    if (i == 8) dma_piece(std::integral_constant<int, 3>{});
                    if (i == 12) dma_piece(std::integral_constant<int, 4>{});
                    if (i == 16) dma_piece(std::integral_constant<int, 5>{});
                    if (i == 20) dma_piece(std::integral_constant<int, 6>{});
                    if (i == 24) dma_piece(std::integral_constant<int, 7>{});
                    if (i == 26) dma_piece(std::integral_constant<int, 8>{});
                    if (i == 28) dma_piece(std::integral_constant<int, 9>{});
                } else {
                    if (i == 0) dma_piece(std::integral_constant<int, 0>{});
                    if (i == 1) dma_piece(std::integral_constant<int, 1>{});
                    if (i == 3) dma_piece(std::integral_constant<int, 2>{});
                    if (i == 5) dma_piece(std::integral_constant<int, 3>{});
                    if (i == 8) dma_piece(std::integral_constant<int, 4>{});
                }
            }
            if constexpr (READ) {
                if (i == NMFMA - 4 + (RB == 2 ? 1 : 0)) {  // gap 12 / 29
                    rd = rd + STAGE_BYTES == SMEM_BYTES ? 0 : rd + STAGE_BYTES;
                    asm volatile("" : "+s"(rd));
                }
                if (i == NMFMA - 3 + (RB == 2 ? 1 : 0)) {  // gap 13 / 30: behind the step's last DMA piece
                    wr = wr + STAGE_BYTES == SMEM_BYTES ? 0 : wr + STAGE_BYTES;
                    ka += BK * 2;
                    kb += kTileBytes;
                    asm volatile("" : "+s"(wr), "+s"(ka), "+s"(kb));
                }
                if (i == NMFMA - 1) {  // behind the step's last fragment request
                    ca0 = ra0;
                    ca1 = ra1;
                    ra0 = rd + c_a0;
                    ra1 = rd + c_a1;
                    rb0 = rd + c_b0;
                    rb1 = rd + c_b1;
                    asm volatile("" : "+v"(ca0), "+v"(ca1), "+v"(ra0), "+v"(ra1), "+v"(rb0), "+v"(rb1));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (READ) {
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                wnext.f[jj][0] = make_frag(as_f16x2(wd[jj][0]), as_f16x2(wd[jj][1]), as_f16x2(wd[jj][2]), as_f16x2(wd[jj][3]));
                wnext.f[jj][1] = make_frag(as_f16x2(wd[jj][4]), as_f16x2(wd[jj][5]), as_f16x2(wd[jj][6]), as_f16x2(wd[jj][7]));
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
    };

    // ---- prologue: STAGES-1 stages in flight; stage 0 -> fragments ----
    asm volatile("" ::"v"(scale2[0]));
    {
        int pwr = 0, pka = k0 * BK * 2, pkb = k0 * kTileBytes;
#pragma unroll
        for (int s = 0; s < ST - 1; ++s) {  // KT >= STAGES - 1 by launch contract
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {  // plain form: voff carries -IMM, so the LDS address gets it back here
                if (i < APW)
                    dma16(x_rsrc, dma_voff[i] + (i & 3) * 1024, pka, smem + pwr + dma_lds_a + i * 1024);
                else
                    dma16(w_rsrc, dma_voff[i] + (i - APW) * 1024, pkb, smem + pwr + dma_lds_b + (i - APW) * 1024);
            }
            pwr += STAGE_BYTES;
            pka += BK * 2;
            pkb += kTileBytes;
        }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PIECES) : "memory");  // stage 0 landed
    __builtin_amdgcn_s_barrier();
    EETQ_GEMM_STAMP(1);
    Frags f0, f1;
    WFrag w0, w1;
    {
        f0.wq[0] = lds_read16(c_b0);
        if constexpr (J == 2) f0.wq[1] = lds_read16(c_b1);
        if constexpr (RH * RB == 2) {  // the window's first pairs
#pragma unroll
            for (int q = 0; q < DIST; ++q) xw[q] = __builtin_bit_cast(f16x8, lds_read16(c_a0 + q * 32 * 128));
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f0.xa[0][mt] = __builtin_bit_cast(f16x8, lds_read16(c_a0 + mt * 32 * 128));
                f0.xa[1][mt] = __builtin_bit_cast(f16x8, lds_read16(c_a1 + mt * 32 * 128));
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f16x2 wdq[8];
            dequant_16(f0.wq[j], scale2[j], wdq);
            w0.f[j][0] = make_frag(wdq[0], wdq[1], wdq[2], wdq[3]);
            w0.f[j][1] = make_frag(wdq[4], wdq[5], wdq[6], wdq[7]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }

    // REM = K steps after this one, clamped to STAGES-1.  REM >= STAGES-1: steady state (DMA for stage kt+STAGES-1).
    auto k_step = [&](auto rem_tag, const WFrag& wcur, const Frags& fcur, WFrag& wnext, Frags& fnext) {
        constexpr int REM = decltype(rem_tag)::value;
        if constexpr (REM >= 1) {
            constexpr int younger = (REM - 1) < (ST - 3) ? (REM - 1) : (ST - 3);
            if constexpr (!(ABLATE & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger * PIECES) : "memory");
            if constexpr (!(ABLATE & 16)) __builtin_amdgcn_s_barrier();
            if constexpr (RH * RB == 2) step_tall(wcur, std::true_type{}, fnext, wnext, std::integral_constant<bool, (REM >= ST - 1)>{});
            else step(wcur, fcur, std::true_type{}, fnext, wnext, std::integral_constant<bool, (REM >= ST - 1)>{});
        } else {
            if constexpr (RH * RB == 2) step_tall(wcur, std::false_type{}, fnext, wnext, std::false_type{});
            else step(wcur, fcur, std::false_type{}, fnext, wnext, std::false_type{});
        }
    };
    using Steady = std::integral_constant<int, ST - 1>;
    const int tail = (ksteps & 1) ? 5 : 6;
    int       kt   = 0;
    for (; kt < ksteps - tail; kt += 2) {
        k_step(Steady{}, w0, f0, w1, f1);
        k_step(Steady{}, w1, f1, w0, f0);
    }
    EETQ_GEMM_STAMP(2);
    if (tail == 6) {
        k_step(std::integral_constant<int, 5>{}, w0, f0, w1, f1);
        k_step(std::integral_constant<int, 4>{}, w1, f1, w0, f0);
        k_step(std::integral_constant<int, 3>{}, w0, f0, w1, f1);
        k_step(std::integral_constant<int, 2>{}, w1, f1, w0, f0);
        k_step(std::integral_constant<int, 1>{}, w0, f0, w1, f1);
        k_step(std::integral_constant<int, 0>{}, w1, f1, w0, f0);
    } else {
        k_step(std::integral_constant<int, 4>{}, w0, f0, w1, f1);
        k_step(std::integral_constant<int, 3>{}, w1, f1, w0, f0);
        k_step(std::integral_constant<int, 2>{}, w0, f0, w1, f1);
        k_step(std::integral_constant<int, 1>{}, w1, f1, w0, f0);
        k_step(std::integral_constant<int, 0>{}, w0, f0, w1, f1);
    }

    // ---- epilogue.  The two K halves are combined through LDS (group 1 parks its accumulators, 16-byte LDS accesses; group 0
    // adds).  An MFMA accumulator lane holds ONE output row and 4 x 4 columns of it, so storing from registers means 8-byte
    // writes scattered over 32 rows per instruction: 3.1 us of a 36.5 us kernel (tools/kbench_stamps gemmstamps).  Instead
    // group 0 rounds to fp16 (+ bias / activation) into a row-major LDS image of the tile, and all four waves write it out 16
    // bytes per lane, whole 256-byte rows (4 rows per wave instruction), adding the residual on the way. ----
    EETQ_GEMM_STAMP(3);
    // (tall tile: one 128-row half after the other -- the parked K half (64 KiB) and the fp16 image (34 KiB) of both do not fit)
#pragma unroll
    for (int ph = 0; ph < RH * RB; ++ph) {
    const bool mine = RH == 1 || rh == ph;  // this wave's rows are the phase's
    const int  ab   = RB == 2 ? 4 * ph : 0; // deep tile: the phase's four row blocks of this wave's eight
    __builtin_amdgcn_s_barrier();
    f32x4* red4 = reinterpret_cast<f32x4*>(smem) + (size_t)wn * (16 * J) * 64;  // [block][quad][lane]
    if (grp == 1 && mine) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red4[((mt * J + j) * 4 + q) * 64 + lane] =
                        f32x4{acc[ab + mt][j][4 * q], acc[ab + mt][j][4 * q + 1], acc[ab + mt][j][4 * q + 2], acc[ab + mt][j][4 * q + 3]};
    }
    __syncthreads();
    EETQ_GEMM_STAMP(4);
    if constexpr (SPLIT) {
        // ---- this slice's partial tile -> slab; the last of the tile's S slices adds all of them in slice order ----
        static_assert(!SPLIT || (CW == 2 && !ACT), "the split form exists for the 4-wave tile with the identity epilogue");
        constexpr int kSlabFloats = BM * BN;
        const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            slabs + (size_t)tile * S * kSlabFloats, 0, S * kSlabFloats * 4, 0x00020000);
        // float4 index inside a slab: (((wn*4 + mt)*J + j)*4 + q)*64 + lane  (only the K-half-0 waves hold sums)
        const int lane_off = (wn * 16 * J * 64 + lane) * 16;
        if (grp == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 o = red4[((mt * J + j) * 4 + q) * 64 + lane];
                        acc[mt][j][4 * q + 0] += o.x;
                        acc[mt][j][4 * q + 1] += o.y;
                        acc[mt][j][4 * q + 2] += o.z;
                        acc[mt][j][4 * q + 3] += o.w;
                    }
            if (S > 1) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            // (through named floats: __builtin_bit_cast applied directly to an element of the 16-wide
                            // accumulator vector read element 0 for every index with this hipcc)
                            const float f0 = acc[mt][j][4 * q + 0], f1 = acc[mt][j][4 * q + 1], f2 = acc[mt][j][4 * q + 2],
                                        f3 = acc[mt][j][4 * q + 3];
                            const u32x4 v = {__builtin_bit_cast(u32, f0), __builtin_bit_cast(u32, f1), __builtin_bit_cast(u32, f2),
                                             __builtin_bit_cast(u32, f3)};
                            // everything in the per-lane offset, soffset 0: hipcc then guards the store's data registers
                            // itself (gemm_splitk_kernel.hpp has the story of the form it does not guard); the build
                            // disassembles this object and fails if a store's data register is rewritten too early or
                            // the uniform part ever moves into an SGPR soffset (tools/check_store_hazard.py, Makefile)
                            __builtin_amdgcn_raw_buffer_store_b128(v, s_rsrc, slice * kSlabFloats * 4 + ((mt * J + j) * 4 + q) * 1024 + lane_off,
                                                                   0, /*sc1*/ 16);
                        }
            }
        }
        if (S > 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores
            __syncthreads();
            unsigned* flag = reinterpret_cast<unsigned*>(smem + SMEM_BYTES - 16);
            if (tid == 0) *flag = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned ticket = *flag;
            if ((ticket & (unsigned)(S - 1)) != (unsigned)(S - 1)) return;  // not the last slice of this tile
            // All four waves read back: wave (grp, wn) takes row blocks 2*grp, 2*grp + 1 of its column half, every slice's
            // float4s requested before the first is used (S * 8 * J loads in flight per lane), summed in slice order -- the
            // last arriver's own slab like the others, so the result does not depend on who arrived last.
            constexpr int kMaxS = J == 1 ? 4 : 2;  // slices a launch may use: the read-back keeps kMaxS * 8 * J float4 per lane
            u32x4 part[kMaxS][2][J][4];
#pragma unroll
            for (int sl = 0; sl < kMaxS; ++sl)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)  // slices beyond S: clamped address, value unused (no load behind a branch)
                            part[sl][mm][j][q] = __builtin_amdgcn_raw_buffer_load_b128(
                                s_rsrc, (sl < S ? sl : S - 1) * kSlabFloats * 4 + (((2 * grp + mm) * J + j) * 4 + q) * 1024 + lane_off, 0, /*sc1*/ 16);
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float t = __builtin_bit_cast(float, (u32)part[0][mm][j][q][i]);
#pragma unroll
                            for (int sl = 1; sl < kMaxS; ++sl) {
                                const float v = __builtin_bit_cast(float, (u32)part[sl][mm][j][q][i]);
                                t             = sl < S ? t + v : t;
                            }
                            // both row blocks of this wave end up in acc[mm] .. the epilogue below maps them back to 2*grp + mm
                            if (mm == 0) acc[0][j][4 * q + i] = t; else acc[1][j][4 * q + i] = t;
                        }
        }
    }
    constexpr int kRowHalfs = BN + 8;                       // row stride of the image: 272 / 144 bytes (bank shift per row)
    f16* image = reinterpret_cast<f16*>(smem + CW * (16 * J) * 64 * 16);  // behind every column part's parked accumulators
    static_assert(CW * (16 * J) * 64 * 16 + BM * kRowHalfs * 2 <= SMEM_BYTES, "the output image must fit behind the parked halves");
    // who rounds which row blocks into the image: the K-half-0 waves all four -- or, after a split read-back, every wave the two
    // it summed (held in acc[0], acc[1])
    const bool summed = SPLIT && S > 1;
    if ((grp == 0 && mine) || summed) {
#pragma unroll
        for (int mt_ = 0; mt_ < 4; ++mt_) {
            if (summed && mt_ >= 2) break;
            const int mt = summed ? 2 * grp + mt_ : mt_;  // row block of the tile; its sums sit in acc[mt_]
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int ncol = wn * WN_COLS + 32 * j + 4 * fh;  // tile-local column of quad 0
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (SPLIT: the other K half -- and the other slices -- were added above)
                    const f32x4 o  = SPLIT ? f32x4{0.f, 0.f, 0.f, 0.f} : red4[((mt * J + j) * 4 + q) * 64 + lane];
                    const float a4[4] = {acc[ab + mt_][j][4 * q + 0] + o.x, acc[ab + mt_][j][4 * q + 1] + o.y, acc[ab + mt_][j][4 * q + 2] + o.z,
                                         acc[ab + mt_][j][4 * q + 3] + o.w};
                    f16x2      lo = {}, hi = {};
                    const bool in_n = n0 + ncol + 8 * q < N;  // columns beyond a ragged launch edge: nothing to read or keep
                    if constexpr (ACT) {
                        if (in_n) finish_quad(a4, ep, n0 + ncol + 8 * q, lo, hi);
                    } else {  // identity: round to fp16, then the fp16 bias add (the reference's `output + bias`)
                        lo = f16x2{(f16)a4[0], (f16)a4[1]};
                        hi = f16x2{(f16)a4[2], (f16)a4[3]};
                        if (ep.bias && in_n) {
                            const u32x2 b = *reinterpret_cast<const u32x2*>(ep.bias + n0 + ncol + 8 * q);
                            lo            = lo + as_f16x2(b.x);
                            hi            = hi + as_f16x2(b.y);
                        }
                    }
                    *reinterpret_cast<u32x2*>(image + (mt * 32 + fn) * kRowHalfs + ncol + 8 * q) = u32x2{as_u32(lo), as_u32(hi)};
                }
            }
        }
    }
    __syncthreads();
    if constexpr (GLU) {
        static_assert(!GLU || (!ACT && !SPLIT), "the gated write-out exists for the unsplit identity tile");
        constexpr int kLanesPerRow = BN / 16, kRowsPerWave = 64 / kLanesPerRow, kRowsPerRound = NW * kRowsPerWave;
        const int     c            = (lane % kLanesPerRow) * 16;  // one group per lane: 8 gate + 8 up halfs -> 8 outputs
#pragma unroll
        for (int r0 = 0; r0 < BM; r0 += kRowsPerRound) {
            const int r = r0 + wave * kRowsPerWave + lane / kLanesPerRow;
            const int m = m0 + ph * BM + r;
            if (m < M && n0 + c < N) {
                const f16x8 g = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(image + r * kRowHalfs + c));
                const f16x8 u = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(image + r * kRowHalfs + c + 8));
                f16x8       o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = silu_mul_f16(g[j], u[j]);
                *reinterpret_cast<f16x8*>(y + (size_t)m * ldc + ((n0 + c) >> 1)) = o;
            }
        }
    } else {
        constexpr int kLanesPerRow = BN / 8, kRowsPerWave = 64 / kLanesPerRow, kRowsPerRound = NW * kRowsPerWave;
        const int     c            = (lane % kLanesPerRow) * 8;
#pragma unroll
        for (int r0 = 0; r0 < BM; r0 += kRowsPerRound) {
            const int r = r0 + wave * kRowsPerWave + lane / kLanesPerRow;
            const int m = m0 + ph * BM + r;
            if (m < M && n0 + c < N) {
                u32x4 v = *reinterpret_cast<const u32x4*>(image + r * kRowHalfs + c);
                if (ep.residual) {
                    const u32x4 rr = *reinterpret_cast<const u32x4*>(ep.residual + (size_t)m * ldc + n0 + c);
                    v.x = as_u32(as_f16x2(v.x) + as_f16x2(rr.x));
                    v.y = as_u32(as_f16x2(v.y) + as_f16x2(rr.y));
                    v.z = as_u32(as_f16x2(v.z) + as_f16x2(rr.z));
                    v.w = as_u32(as_f16x2(v.w) + as_f16x2(rr.w));
                }
                *reinterpret_cast<u32x4*>(y + (size_t)m * ldc + n0 + c) = v;
            }
        }
    }
    if constexpr (RH * RB > 1) __syncthreads();  // the next phase reuses the parked area and the image
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    EETQ_GEMM_STAMP(5);
}

template <int ABLATE, int J, bool ACT = false, int CW = 2, bool GLU = false, int RH = 1, int RB = 1>
__global__ __launch_bounds__(128 * CW * RH, 1) void gemm_tile_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int M, int N, int K, int ldc, Epilogue ep)
{
    gemm_tile_body<ABLATE, J, ACT, CW, false, GLU, RH, RB>(x, w, scales, y, M, N, K, ldc, ep, 1, nullptr, nullptr);
}

// Round 5, measured and shelved with their patch (tools/experiments/tile_ring_depth_and_persistent.patch): the
// ring depth as a template parameter (3 / 4 slots with one or two workgroups per CU for short K slices: 3 slots 30-60 % slower,
// 4 slots within +-2 % of 6) and a persistent form of the wide tile (one workgroup per CU walks its tiles, the next tile's stages
// requested under a two-pass epilogue: bit-identical, 57.9 k cycles per tile against 58.0 k, 0.98-1.00 x the time).
// grid = tiles * S workgroups; every slice must own >= kMinKSteps K steps; counters: one per tile, shared only by launches
// with the same S (they grow by S per launch; "last" is (old & (S-1)) == S-1), S in {2, 4} for J = 1, S = 2 for J = 2
template <int J>
__global__ __launch_bounds__(256, 1) void gemm_tile_splitk_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int M, int N, int K, int ldc, Epilogue ep, int S, float* __restrict__ slabs,
    unsigned* __restrict__ counters)
{
    gemm_tile_body<0, J, false, 2, true>(x, w, scales, y, M, N, K, ldc, ep, S, slabs, counters);
}

}  // namespace gemm
}  // namespace eetq
