// Fused int8-dequant + fp16 MFMA GEMM for prefill (M >= 5), gfx950.
//
// Replaces the reference's CUTLASS path: ft::gemm_fp16_int (csrc/cutlass_kernels/fpA_intB_gemm.cu:21-33) ->
// GemmFpAIntB / DqMmaMultistage (csrc/cutlass_extensions/.../gemm/kernel/fpA_intB_gemm.h:296-462,
// gemm/threadblock/dq_mma_multistage.h:310-590).  Numerics contract kept from that code: int8 -> fp16 exact,
// multiplied by the per-column fp16 scale in registers *before* the matrix instruction
// (mma_tensorop_dequantizer.h:259-274), fp32 accumulation (default_fpA_intB_traits.h:110), fp16 store with
// alpha = 1, beta = 0 (epilogue_helpers.h:73-80).  Nothing else is shared with it.
//
// Structure (details in gemm_kernel.hpp):
//   * workgroup tile 128(M) x 128(N), K step 64; 4 waves = 2 K halves x 2 column halves; a wave owns 128 rows x
//     64 columns of one 32-deep K half, so each dequantised weight fragment is reused by 4 MFMAs (M-repeat) and each
//     activation fragment by 2 (N-repeat) -- the int8->fp16 VALU work and the LDS reads are the scarce resources;
//   * v_mfma_f32_32x32x16_f16 with the weights as the A operand (4 consecutive accumulators = 4 consecutive n of one
//     token: 8-byte fp16 stores, no transpose), 128 fp32 accumulators per lane; the two K halves are added once in LDS;
//   * A (fp16 activations) and B (uint8 native tiles) are staged HBM/L2 -> LDS with buffer_load_dwordx4 ... lds
//     (LDS-DMA, no VGPR round trip), 6-stage ring, one s_barrier per K step, counted vmcnt;
//   * A's LDS image is XOR-swizzled through the *source* address (key (row>>1)&7 on the 16-byte slot): ds_read_b128 of
//     32 rows at one k offset is conflict-free; B's LDS image is the lane-linear global tile, conflict-free as is;
//   * LDS -> register fragment reads, the dequant of the next K step and the DMA of stage kt+5 are hand-placed in the
//     shadow of the current step's MFMAs (sched_barrier-pinned issue order, see gemm_kernel.hpp);
//   * blockIdx -> tile mapping gives each XCD (own L2) a contiguous run of tiles ordered in groups of 4 row tiles: the
//     32 workgroups resident on an XCD cover 4 row tiles x 8 column tiles, the smallest fabric footprint per K step.
#include <cstdio>
#include <cstdlib>

#include "gemm_kernel.hpp"

namespace eetq {

using namespace gemm;

namespace {
// S (2 or 4) K slices of the 128 x 64 tile over the columns [c0, c0 + cols) of an M x N problem whose row stride is ldc.
// EETQ_ERR_UNSUPPORTED (no message): the caller runs those columns unsplit.
int launch_tile_splitk_cols(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int rows, int c0, int cols,
                            int K, int ldc, int S, hipStream_t stream)
{
    static const bool allowed = [] {  // EETQ_AMD_SPLITK=0: no library-owned scratch anywhere (gemm_splitk.hip)
        const char* e = getenv("EETQ_AMD_SPLITK");
        return !(e && e[0] == '0');
    }();
    constexpr int BN = TileCfg<1>::BN;
    const int     KT = K / BK;
    if (!allowed || (S != 2 && S != 4) || ep.act != 0 || K % BK != 0 || KT / S < kMinKSteps) return EETQ_ERR_UNSUPPORTED;
    const int tiles = ((rows + BM - 1) / BM) * ((cols + BN - 1) / BN);
    float*    slabs = nullptr;
    unsigned *t2 = nullptr, *t4 = nullptr;
    size_t    slab_bytes = 0, max_tiles = 0;
    int st = splitk_region(stream, &slabs, &slab_bytes, &t2, &t4, &max_tiles);
    if (st == EETQ_ERR_UNSUPPORTED || (st == EETQ_OK && ((size_t)tiles > max_tiles || (size_t)tiles * S * BM * BN * 4 > slab_bytes)))
        return EETQ_ERR_UNSUPPORTED;  // no scratch of its own for this stream right now
    if (st != EETQ_OK) return st;
    static std::atomic<unsigned long long> opted{0};
    st = opt_in_large_lds(gemm_tile_splitk_kernel<1>, opted);
    if (st != EETQ_OK) return st;
    Epilogue e = ep;
    if (e.bias) e.bias += c0;
    if (e.residual) e.residual += c0;
    const uint8_t* wc = w + (size_t)(c0 / kTileN) * (K / kTileK) * kTileBytes;
    launch_kernel(gemm_tile_splitk_kernel<1>, dim3(tiles * S), dim3(256), TileCfg<1>::SMEM_BYTES, stream, x, wc, scales + c0, y + c0, rows,
                  cols, K, ldc, e, S, slabs, S == 2 ? t2 : t4);
    return check_hip(hipGetLastError(), "gemm_tile_splitk_kernel launch");
}
}  // namespace

int launch_gemm_mfma(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                     hipStream_t stream)
{
    // kActGlu8: the weight's columns are gate / up groups of 8 + 8 and y is [M][N / 2] (gemm_kernel.hpp, GLU); no residual
    const bool glu = ep.act == kActGlu8;
    if (glu && (K / BK < kMinKSteps || N % 16 != 0 || ep.residual)) return EETQ_ERR_UNSUPPORTED;  // quiet: the caller runs two launches
    if (K / BK < kMinKSteps) {
        // K < 320: a few KiB of weights per column tile; run the stream kernel over 64-row chunks instead of carrying a
        // second tiled kernel for it (the weights are re-read from L2, the activations are read once)
        for (int m = 0; m < M; m += kStreamMaxM) {
            const int rows = M - m < kStreamMaxM ? M - m : kStreamMaxM;
            Epilogue  e    = ep;
            if (e.residual) e.residual += (size_t)m * N;
            int       st   = launch_streamk(x + (size_t)m * K, w, scales, e, y + (size_t)m * N, rows, N, K, stream);
            if (st != EETQ_OK) return st;
        }
        return EETQ_OK;
    }
    {   // > 64 KiB of dynamic LDS: one opt-in per kernel and device (common.hpp)
        static std::atomic<unsigned long long> opted2{0}, opted1{0}, opted2a{0}, opted1a{0}, opted2g{0}, opted1g{0}, opted_tall{0};
        static std::atomic<unsigned long long> opted_deep{0};
        int st = opt_in_large_lds(gemm_tile_kernel<0, 2, false, 2, false, 2>, opted_tall);
        if (st == EETQ_OK) st = opt_in_large_lds(gemm_tile_kernel<0, 2, false, 2, false, 1, 2>, opted_deep);
        if (st != EETQ_OK) return st;
        if (glu) {
            st = opt_in_large_lds(gemm_tile_kernel<0, 2, false, 2, true>, opted2g);
            if (st == EETQ_OK) st = opt_in_large_lds(gemm_tile_kernel<0, 1, false, 2, true>, opted1g);
        } else if (ep.act == 0) {
            st = opt_in_large_lds(gemm_tile_kernel<0, 2>, opted2);
            if (st == EETQ_OK) st = opt_in_large_lds(gemm_tile_kernel<0, 1>, opted1);
        } else {  // the activation epilogues are their own instantiation (gemm_kernel.hpp)
            st = opt_in_large_lds(gemm_tile_kernel<0, 2, true>, opted2a);
            if (st == EETQ_OK) st = opt_in_large_lds(gemm_tile_kernel<0, 1, true>, opted1a);
        }
        if (st != EETQ_OK) return st;
    }
    // the LDS-DMA path addresses its operands with 32-bit buffer offsets
    EETQ_REQUIRE((size_t)N * K < (1ull << 31), "weight larger than 2 GiB is not supported by the buffer-addressed DMA path");
    // activations: split M into row chunks below 2 GiB (a multiple of the 128-row tile), one launch per chunk
    const size_t row_bytes  = (size_t)K * 2;
    const int    max_rows   = (int)((((1ull << 31) - 1) / row_bytes) / BM * BM);
    EETQ_REQUIRE(max_rows >= BM, "K too large for the buffer-addressed DMA path");
    const int n_cu = device_cu_count();
    // one launch over columns [c0, c0 + cols) of the problem with the cheaper of the two tile shapes
    auto launch_cols = [&](int m, int rows, int c0, int cols, int force_j) -> int {
        const int tiles_m = (rows + BM - 1) / BM;
        const int tiles2  = tiles_m * ((cols + TileCfg<2>::BN - 1) / TileCfg<2>::BN);
        const int tiles1  = tiles_m * ((cols + TileCfg<1>::BN - 1) / TileCfg<1>::BN);
        // 128 x 128 tiles are the efficient shape when they fill the chip; 128 x 64 tiles double the workgroup count:
        // they win when the wide tiles leave CUs idle (tiles < CUs) or end in a mostly empty round.  Cost in units of one
        // wide-tile pass; a narrow tile costs kNarrow of it (measured, profiles/r01_kbench_tile_shapes.txt).
        constexpr double kNarrow = 0.70;
        const double cost2 = (double)((tiles2 + n_cu - 1) / n_cu);
        const double cost1 = kNarrow * (double)((tiles1 + n_cu - 1) / n_cu);
        Epilogue     e     = ep;
        if (e.bias) e.bias += c0;
        if (e.residual) e.residual += (size_t)m * N + c0;
        const uint8_t* wc = w + (size_t)(c0 / kTileN) * (K / kTileK) * kTileBytes;
        const bool     narrow = force_j == 1 || (force_j == 0 && cost1 < cost2);
        auto go = [&](auto kern, int tiles, size_t smem, int threads = 256) {
            const int ldc = glu ? N / 2 : N;
            launch_kernel(kern, dim3(tiles), dim3(threads), smem, stream, x + (size_t)m * K, wc, scales + c0,
                          y + (size_t)m * ldc + (glu ? c0 / 2 : c0), rows, cols, K, ldc, e);
        };
        // the tall tile (256 x 128 on eight waves, gemm_kernel.hpp): EETQ_AMD_TILE_TALL=1 (behind EETQ_AMD_TUNING) for A/B runs
        static const int tall_env = [] {
            const char* t = tuning_env("EETQ_AMD_TILE_TALL");
            return t ? atoi(t) : 0;
        }();
        if (tall_env == 2 && !glu && e.act == 0 && rows >= 256) {  // the deep tile: 256 x 128 on four waves
            using Deep = TileCfg<2, 2, 1, 2>;
            const int tiles_d = ((rows + Deep::ROWS - 1) / Deep::ROWS) * ((cols + Deep::BN - 1) / Deep::BN);
            go(gemm_tile_kernel<0, 2, false, 2, false, 1, 2>, tiles_d, Deep::SMEM_BYTES, 256);
            return check_hip(hipGetLastError(), "gemm_tile_kernel (deep) launch");
        }
        if (tall_env == 1 && !glu && e.act == 0 && rows >= 256) {
            using Tall = TileCfg<2, 2, 2>;
            const int tiles_t = ((rows + Tall::ROWS - 1) / Tall::ROWS) * ((cols + Tall::BN - 1) / Tall::BN);
            go(gemm_tile_kernel<0, 2, false, 2, false, 2>, tiles_t, Tall::SMEM_BYTES, 512);
            return check_hip(hipGetLastError(), "gemm_tile_kernel (tall) launch");
        }
        if (glu && narrow) go(gemm_tile_kernel<0, 1, false, 2, true>, tiles1, TileCfg<1>::SMEM_BYTES);
        else if (glu) go(gemm_tile_kernel<0, 2, false, 2, true>, tiles2, TileCfg<2>::SMEM_BYTES);
        else if (narrow && e.act == 0) go(gemm_tile_kernel<0, 1>, tiles1, TileCfg<1>::SMEM_BYTES);
        else if (narrow) go(gemm_tile_kernel<0, 1, true>, tiles1, TileCfg<1>::SMEM_BYTES);
        else if (e.act == 0) go(gemm_tile_kernel<0, 2>, tiles2, TileCfg<2>::SMEM_BYTES);
        else go(gemm_tile_kernel<0, 2, true>, tiles2, TileCfg<2>::SMEM_BYTES);
        return check_hip(hipGetLastError(), "gemm_tile_kernel launch");
    };
    for (int m = 0; m < M; m += max_rows) {
        const int rows    = M - m < max_rows ? M - m : max_rows;
        const int tiles_m = (rows + BM - 1) / BM;
        const int tn2     = (N + TileCfg<2>::BN - 1) / TileCfg<2>::BN;
        const int T2      = tiles_m * tn2;
        // Whole rounds of wide tiles, then the ragged last round: when that round would be less than half full its
        // columns go to narrow tiles in a second launch (M = 1024, N = 5120: 320 wide tiles = 256 + 64 -> 256 wide +
        // 128 narrow: 73.8 -> ~59 us).  Tile rows of the weight layout are 16 columns, so any multiple of 128 splits.
        const int rem = T2 % n_cu;
        if (T2 > n_cu && rem != 0 && rem * 2 < n_cu && tiles_m <= n_cu) {
            const int cols1 = ((T2 - rem) / tiles_m) * TileCfg<2>::BN;  // columns covered by complete rounds (rounded down)
            if (cols1 > 0 && cols1 < N) {
                int st = launch_cols(m, rows, 0, cols1, 2);
                if (st != EETQ_OK) return st;
                // the ragged round: narrow tiles in TWO K slices when those fill the chip once -- half the loop for ~3.5 us of
                // hand-over (M = 1024, N = 5120: 128 narrow tiles -> 256 workgroups of K / 2; K = 13824 164 -> ~135 us)
                const int rem_tiles = tiles_m * ((N - cols1 + TileCfg<1>::BN - 1) / TileCfg<1>::BN);
                st                  = EETQ_ERR_UNSUPPORTED;
                if (rem_tiles * 2 <= n_cu && (K / BK) / 2 >= 40) {
                    Epilogue e = ep;
                    if (e.residual) e.residual += (size_t)m * N;
                    st = launch_tile_splitk_cols(x + (size_t)m * K, w, scales, e, y + (size_t)m * N, rows, cols1, N - cols1, K, N, 2, stream);
                }
                if (st == EETQ_ERR_UNSUPPORTED) st = launch_cols(m, rows, cols1, N - cols1, 0);
                if (st != EETQ_OK) return st;
                continue;
            }
        }
        int st = launch_cols(m, rows, 0, N, 0);
        if (st != EETQ_OK) return st;
    }
    return EETQ_OK;
}

// ---- K slices of the 128 x 64 tile ------------------------------------------------------------------------------------
// The tiled kernel runs ONE workgroup per tile: M <= 128 at N = 4096 is 64 tiles on 256 CUs (20.6 us at M = 128), M = 256 is
// 128.  With S workgroups per tile, each on a contiguous S-th of the K steps, the chip fills and the per-workgroup loop
// shortens S-fold; the price is the hand-over of S partial tiles (32 KiB each) to the workgroup that finishes last.
int tile_splitk_slices(int M, int N, int K)
{
    const int ncu   = device_cu_count();
    const int tiles = ((M + BM - 1) / BM) * ((N + TileCfg<1>::BN - 1) / TileCfg<1>::BN);
    const int KT    = K / BK;
    // Every workgroup must get a CU of its own (a second round costs more than the slices save: M = 256 at 5120^2, 160 tiles,
    // 35.8 us with two slices vs 27.6 unsplit), and the hand-over (~3.5 us: publish, ticket, read-back of S slabs) must be
    // small against the loop it shortens -- measured (tools/experiments/tilesplit_check.py, graph-replayed chains, us,
    // split vs best other path):
    //   four slices of 43 steps:  M = 128 at 11008 x 4096 23.9 vs 27.1, M = 100 23.3 vs 25.6
    //   two slices of >= 40:      M = 128 at 5120^2 20.1 vs 22.6, 13824 x 5120 40.9 vs 46.8, M = 256 at 11008 x 4096 37.1 vs 54.4
    //   four slices of 16:        M = 97..128 at 4096^2 13.5 vs 15.5 on one box, 15.3-15.8 vs 15.0-15.5 on two others: not taken
    //   two slices of 32:         M = 160 at 4096^2 17.7 vs 16.7, M = 256 19.8 vs 20.9: not taken
    if (tiles * 4 <= ncu && KT / 4 >= 24) return 4;
    if (tiles * 2 <= ncu && KT / 2 >= 40) return 2;
    return 1;
}

// the same question for the 128 x 128 tile (two slices at most: its partial tile is 64 KiB): M = 257..512 on N <= 4096, where
// whole wide tiles cover half the chip and the narrow tiles that fill it cost 0.70 of a wide pass each.  The 64 KiB hand-over
// is dearer (~6 us), so only very deep K pays: M = 512 at 11008 x 4096 (86 steps per slice) 51.7 vs 58.3 us; at 4096^2 (32)
// 27.8 vs 23.2, 8192 x 4096 (64) 40.6 vs 40.0: not taken.
int wide_tile_splitk_slices(int M, int N, int K)
{
    const int ncu   = device_cu_count();
    const int tiles = ((M + BM - 1) / BM) * ((N + TileCfg<2>::BN - 1) / TileCfg<2>::BN);
    const int KT    = K / BK;
    return (tiles * 2 <= ncu && tiles * 4 > ncu && KT / 2 >= 80) ? 2 : 1;
}

int launch_gemm_tile_splitk(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                            hipStream_t stream, int force_s, int* used_s, bool env_plan)
{
    if (used_s) *used_s = 1;
    static const bool allowed = [] {  // EETQ_AMD_SPLITK=0: no library-owned scratch anywhere (gemm_splitk.hip)
        const char* e = getenv("EETQ_AMD_SPLITK");
        return !(e && e[0] == '0');
    }();
    int S = !allowed ? 1 : (force_s ? force_s : tile_splitk_slices(M, N, K));
    bool wide = false;
    if (allowed && !force_s && S == 1 && wide_tile_splitk_slices(M, N, K) == 2) {
        S    = 2;
        wide = true;
    }
    const int KT = K / BK;
    // EETQ_AMD_TILESPLIT_PLAN="S" overrides the slice count of the 128 x 64 tile on the explicitly FORCED path only
    // (EETQ_PATH_TILESPLIT: tuning and tests, read per call); AUTO launches never read it
    if (const char* e = (env_plan && allowed) ? getenv("EETQ_AMD_TILESPLIT_PLAN") : nullptr) {
        int a = 0;
        if (sscanf(e, "%d", &a) == 1) {
            S    = a;
            wide = false;
        }
    }
    const bool fits = (size_t)M * K * 2 < (1ull << 31) && (size_t)N * K < (1ull << 31);
    if ((S != 2 && S != 4) || ep.act != 0 || !fits || K % BK != 0 || KT / S < kMinKSteps)
        return launch_gemm_mfma(x, w, scales, ep, y, M, N, K, stream);
    const int BN    = wide ? TileCfg<2>::BN : TileCfg<1>::BN;
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    float*        slabs = nullptr;
    unsigned *    t2 = nullptr, *t4 = nullptr;
    size_t        slab_bytes = 0, max_tiles = 0;
    int st = splitk_region(stream, &slabs, &slab_bytes, &t2, &t4, &max_tiles);
    if (st == EETQ_ERR_UNSUPPORTED || (st == EETQ_OK && ((size_t)tiles > max_tiles || (size_t)tiles * S * BM * BN * 4 > slab_bytes)))
        return launch_gemm_mfma(x, w, scales, ep, y, M, N, K, stream);  // no scratch of its own for this stream: unsplit
    if (st != EETQ_OK) return st;
    static std::atomic<unsigned long long> opted{0}, opted_wide{0};
    st = wide ? opt_in_large_lds(gemm_tile_splitk_kernel<2>, opted_wide) : opt_in_large_lds(gemm_tile_splitk_kernel<1>, opted);
    if (st != EETQ_OK) return st;
    if (wide)
        launch_kernel(gemm_tile_splitk_kernel<2>, dim3(tiles * S), dim3(256), TileCfg<2>::SMEM_BYTES, stream, x, w, scales, y, M, N, K, N,
                      ep, S, slabs, t2);
    else
        launch_kernel(gemm_tile_splitk_kernel<1>, dim3(tiles * S), dim3(256), TileCfg<1>::SMEM_BYTES, stream, x, w, scales, y, M, N, K, N,
                      ep, S, slabs, S == 2 ? t2 : t4);
    st = check_hip(hipGetLastError(), "gemm_tile_splitk_kernel launch");
    if (st == EETQ_OK && used_s) *used_s = S;
    return st;
}

}  // namespace eetq
