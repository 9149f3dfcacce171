// Kernel template of the small-batch W8A16 / W4A16 "stream" GEMM (AUTO: 2 <= M <= 16; the template itself runs up to 64 rows).
//
// Same HBM-bound weight stream as the GEMV kernel: the waves of a workgroup split K, 16 B/lane tile loads go straight to registers
// and are MFMA B operands as they are (after the exact dequantisation); one workgroup owns NT adjacent 16-column tile rows so each
// activation fragment is reused by NT weight tiles; MT = ceil(M/16) row tiles of v_mfma_f32_16x16x32_f16; the waves' partial sums
// meet in LDS at the end.  Where the activation (A) fragments come from is the template parameter XM:
//   XM = 0  registers: straight from global memory (L2), 16 clamped rows x 128 B per weight tile and wave (any MT)
//   XM = 1  block copy: the M rows once per workgroup into LDS (MT = 1, K % 128 == 0, M*K*2 bytes must fit)
//   XM = 2  per-wave ring of 8 rows (M <= 8), XM = 3 of 4 rows (int4, M <= 4), XM = 4 of 16 rows (M <= 16): LDS-DMA per k tile
//   XM = 5  (round 6, int8) per-wave ring of 32 rows, MT = 2: four DMAs per k tile, two MFMA row tiles per weight register tile
// All forms feed the same fragments to the same MFMAs in the same order: bit-identical results at equal WAVES.
// Included by streamk.hip and tools/kbench.hip.
#pragma once
#include "common.hpp"
#include "gemv_kernel.hpp"

namespace eetq {
namespace streamk {

typedef __attribute__((address_space(3))) void       lds_void;
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;

// grid.x = N / (16*NT); block = WAVES*64; wave w takes k tiles w, w+WAVES, ... (each >= D tiles by launch contract).
// Dynamic LDS: WAVES * MT*NT*256 floats (cross-wave reduction only).
// XM = 1 (MT = 1, K % 128 == 0, M*K*2 bytes fit): the M activation rows are copied ONCE per workgroup into LDS (LDS-DMA,
// 16 B/lane, no VGPR round trip; the 16-byte chunk index within a row is xor-ed with the row number so that the 16 lanes of an MFMA
// A fragment -- 16 rows, same k -- hit 16 different bank groups) and the A fragments are LDS reads: the vector-memory path then
// carries the weight stream only, instead of 16 (clamped) rows x 128 B of activations from L2 per 1 KiB weight tile.
// Dynamic LDS then: [roundup(M*K*2, 1024) activations] + the reduction floats.
// BITS = 4 (W4A16): the 1 KiB tile holds 16 columns x 128 k, a lane's 16 bytes are 32 k values of its column (the int4
// layout of int4.hip / gemv_kernel.hpp): four MFMAs and four activation vectors per tile instead of two.
// XM = 2 (MT = 1, M <= 8; any K): per-wave ring instead of a whole-x copy -- ONE LDS-DMA instruction per int8 k tile and wave (two
// per int4 tile: 8 rows x 256 B) brings the activations the tile needs (lane = row * chunks + chunk, chunk xor row) into a slot of the
// wave's own ring (2*D - 1 slots), issued right before the tile's weight load so the wait for the weights covers it; no barrier, no
// up-front copy, half the activation load instructions of the register form.  XM = 3: int4 with M <= 4 -- 4 rows, one DMA per tile;
// XM = 4: 9 <= M <= 16 -- 16 rows, two DMAs per int8 tile (four per int4 tile: 4 KiB slots, 8-wave workgroups only).  Dynamic LDS then: [WAVES * (2*D - 1) slots] + the reduction floats.
template <int MT, int NT, int WAVES, int D, int MIN_WAVES_PER_SIMD, int BITS = 8, int XM = 0>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void streamk_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int M, int N, int K, Epilogue ep)
{
    constexpr bool XLDS = XM == 1, XRING = XM >= 2;
    constexpr int  kRing     = 2 * D - 1;
    constexpr int  kRowBytes = BITS == 8 ? 128 : 256;          // activation bytes per row and k tile
    constexpr int  kRingRows = XM == 3 ? 4 : XM == 4 ? 16 : XM == 5 ? 32 : 8;  // XM = 4: 9 <= M <= 16, two DMAs per int8 tile, four per int4 tile; XM = 5: 17 <= M <= 32
    constexpr int  kSlot     = kRingRows * kRowBytes;          // 1 KiB; int4 with 8 rows: 2 KiB
    constexpr int  kDma      = kSlot / 1024;
    static_assert(XM == 0 || MT == 1 || (XM == 5 && MT == 2 && BITS == 8), "LDS-staged activations: one row tile (two in the 32-row ring)");
    constexpr int  kRT       = XM == 5 ? 2 : 1;                 // row tiles whose fragments come from the ring
    static_assert(XM != 3 || BITS == 4, "the 4-row ring is the int4 form for M <= 4");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int xs_bytes = XLDS ? ((M * K * 2 + 1023) & ~1023) : XRING ? WAVES * kRing * kSlot : 0;
    float*    red      = reinterpret_cast<float*>(smem + xs_bytes);

    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int g = lane >> 4, c = lane & 15;
    using CD = gemv::Codec<BITS>;
    constexpr int XQ = CD::kXQ;          // 16-byte activation vectors per lane and tile
    const int KT     = K / CD::kTileK;
    const int ntile0 = blockIdx.x * NT;

    u32 sraw[NT];
    if constexpr (!XLDS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) sraw[t] = reinterpret_cast<const uint16_t*>(scales)[(ntile0 + t) * 16 + c];
    }

    // per-lane activation row pointers: row 16*mt + c (clamped: rows >= M compute garbage that is never stored)
    const u32x4* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int r    = 16 * mt + c;
        r        = r < M ? r : M - 1;
        xrow[mt] = reinterpret_cast<const u32x4*>(x + (size_t)r * K + CD::kLaneK * g);  // + kTileK / 8 u32x4 per k tile
    }
    const u32x4* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        wp[t] = reinterpret_cast<const u32x4*>(w + (size_t)(ntile0 + t) * KT * kTileBytes) + lane;  // + 64 per k tile

    // XLDS: the copy goes first in the memory queue (returns are in order), the first D weight stages right behind it
    // a row's 16-byte chunks are xor-ed with the row number inside 256-byte windows: one int4 k tile, two int8 k tiles (bit 3 of
    // the row number then flips the parity of the k tile)
    int xl_base[XQ], xl_hi = 0;
#pragma unroll
    for (int q = 0; q < XQ; ++q) xl_base[q] = 0;
    if constexpr (XLDS) {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, M * K * 2, 0x00020000);
        const int row_bytes = K * 2;
        for (int p = wave; p < (xs_bytes >> 10); p += WAVES) {
            const int flat = p * 1024 + lane * 16;  // LDS byte this lane fills: (row r, chunk) -> source chunk ^ (r & 15)
            const int r    = flat / row_bytes;      // rows are multiples of 256 B: uniform per 16 lanes; r >= M reads 0 (bounds)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(smem + p * 1024), 16,
                                                     r * row_bytes + ((flat - r * row_bytes) ^ ((r & 15) << 4)), 0, 0, 0);
        }
        const int rc   = c < M ? c : M - 1;         // rows >= M: garbage in, never stored
        const int lds0 = (int)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
#pragma unroll
        for (int q = 0; q < XQ; ++q)
            xl_base[q] = lds0 + rc * row_bytes + (((XQ * g + q) ^ (BITS == 8 ? (rc & 7) : rc)) << 4);
        xl_hi = BITS == 8 ? rc >> 3 : 0;
        // the scales come after the copy: a load pending at the head of the copy loop would be waited for there
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < NT; ++t) sraw[t] = reinterpret_cast<const uint16_t*>(scales)[(ntile0 + t) * 16 + c];
    }

    __amdgpu_buffer_rsrc_t ring_rs;
    int                    ring_voff[4] = {0, 0, 0, 0}, ring_rd[kRT][XQ];
    uint8_t*               ring_wr      = smem;
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
        for (int q = 0; q < XQ; ++q) ring_rd[rt][q] = 0;
    if constexpr (XRING) {
        constexpr int kChunks = kRowBytes / 16;
        ring_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, M * K * 2, 0x00020000);
#pragma unroll
        for (int j = 0; j < kDma; ++j) {  // DMA lane = (row, chunk): slot byte j*1024 + lane*16 <- row rr, chunk ^ rr
            const int row = (j * 1024 + lane * 16) / kRowBytes;
            const int rr  = row < M ? row : M - 1;
            ring_voff[j]  = rr * K * 2 + (((lane & (kChunks - 1)) ^ (rr & (kChunks - 1))) << 4);
        }
        ring_wr        = smem + wave * (kRing * kSlot);
        const int lds0 = (int)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem + wave * (kRing * kSlot);
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) {
            const int rc = 16 * rt + c < M ? 16 * rt + c : M - 1;  // fragment lane (g, c): row rc, chunks XQ*g .. XQ*g + XQ - 1
#pragma unroll
            for (int q = 0; q < XQ; ++q) ring_rd[rt][q] = lds0 + rc * kRowBytes + (((XQ * g + q) ^ (rc & (kChunks - 1))) << 4);
        }
    }

    struct Stage {
        u32x4 wq[NT];
        u32x4 xa[XM ? 1 : MT][XM ? 1 : XQ];
        int   kt;
    };
    auto load_stage = [&](int kt, Stage& s, int slot) {
        if constexpr (XRING)  // before the weights: whoever waits for this stage's weights has the activations too
        {
#pragma unroll
            for (int j = 0; j < kDma; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ring_rs, (lds_void*)(ring_wr + slot * kSlot + j * 1024), 16, ring_voff[j],
                                                         kt * kRowBytes, 0, 0);
            // the DMA must stay ahead of this stage's weight loads at every level of the compiler (the reads of the slot are ordered
            // by the wait for those weights): a compiler barrier for the IR passes, a scheduling barrier for the machine scheduler
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) s.wq[t] = gemv::load_w<true>(wp[t] + (size_t)kt * 64);
        if constexpr (XRING) {
            s.kt = slot;
        } else if constexpr (XLDS) {
            s.kt = kt;
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < XQ; ++q) s.xa[mt][q] = xrow[mt][(size_t)kt * (CD::kTileK / 8) + q];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    f16x2 scale2[NT];
    auto  consume = [&](const Stage& s) {
        u32x4 xa[XQ], xb[XQ];  // xb: the second row tile of the 32-row ring
        if constexpr (XLDS) {
            const int ko = BITS == 8 ? (s.kt ^ xl_hi) << 7 : s.kt << 8;
#pragma unroll
            for (int q = 0; q < XQ; ++q) xa[q] = *(lds_cu32x4*)(uintptr_t)(uint32_t)(xl_base[q] + ko);
        }
        if constexpr (XRING) {
            // Hand-written LDS reads: hipcc tracks LDS-DMA against the LDS loads it can see and would wait for EVERY DMA in flight
            // (the refill of the other stage included) before each fragment read, i.e. halve the prefetch depth.  The weight register
            // among the inputs makes the compiler wait for this stage's weights first -- the DMA was issued before them and returns
            // are in order, so the slot is complete.
            const int ko = s.kt * kSlot;
            if constexpr (XM == 5) {
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(xa[0]), "=&v"(xa[1]), "=&v"(xb[0]), "=&v"(xb[1])
                             : "v"(ring_rd[0][0] + ko), "v"(ring_rd[0][1] + ko), "v"(ring_rd[1][0] + ko), "v"(ring_rd[1][1] + ko),
                               "v"(s.wq[NT - 1].x)
                             : "memory");
            } else if constexpr (BITS == 8) {
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(xa[0]), "=&v"(xa[1])
                             : "v"(ring_rd[0][0] + ko), "v"(ring_rd[0][1] + ko), "v"(s.wq[NT - 1].x)
                             : "memory");
            } else {
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(xa[0]), "=&v"(xa[1]), "=&v"(xa[2]), "=&v"(xa[3])
                             : "v"(ring_rd[0][0] + ko), "v"(ring_rd[0][1] + ko), "v"(ring_rd[0][2] + ko), "v"(ring_rd[0][3] + ko),
                               "v"(s.wq[NT - 1].x)
                             : "memory");
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (BITS == 8 && XM != 0) {
                f16x2 wq[8];
                dequant_16(s.wq[t], scale2[t], wq);
                const f16x8 b0 = {wq[0].x, wq[0].y, wq[1].x, wq[1].y, wq[2].x, wq[2].y, wq[3].x, wq[3].y};
                const f16x8 b1 = {wq[4].x, wq[4].y, wq[5].x, wq[5].y, wq[6].x, wq[6].y, wq[7].x, wq[7].y};
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xa[0]), b0, acc[0][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xa[1]), b1, acc[0][t], 0, 0, 0);
                if constexpr (XM == 5) {
                    acc[MT - 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xb[0]), b0, acc[MT - 1][t], 0, 0, 0);
                    acc[MT - 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xb[1]), b1, acc[MT - 1][t], 0, 0, 0);
                }
            } else if constexpr (BITS == 8) {
                f16x2 wq[8];
                dequant_16(s.wq[t], scale2[t], wq);
                const f16x8 b0 = {wq[0].x, wq[0].y, wq[1].x, wq[1].y, wq[2].x, wq[2].y, wq[3].x, wq[3].y};
                const f16x8 b1 = {wq[4].x, wq[4].y, wq[5].x, wq[5].y, wq[6].x, wq[6].y, wq[7].x, wq[7].y};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, s.xa[mt][0]), b0,
                                                                        acc[mt][t], 0, 0, 0);
                    acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, s.xa[mt][1]), b1,
                                                                        acc[mt][t], 0, 0, 0);
                }
            } else {
                const u32 wd[4] = {s.wq[t].x, s.wq[t].y, s.wq[t].z, s.wq[t].w};  // dword d = k values 8d .. 8d+7 of the lane
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    f16x2 wq[4];
                    gemv::dequant_dword_i4(wd[d], scale2[t], wq);
                    const f16x8 b = {wq[0].x, wq[0].y, wq[1].x, wq[1].y, wq[2].x, wq[2].y, wq[3].x, wq[3].y};
                    if constexpr (XM != 0) {
                        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xa[d]), b, acc[0][t], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, s.xa[mt][d]), b,
                                                                                acc[mt][t], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- software-pipelined K loop over this wave's tiles: no load behind a branch (see gemv_kernel.hpp) ----
    const int n = (KT - wave + WAVES - 1) / WAVES;  // >= D
    Stage     st[D];
#pragma unroll
    for (int d = 0; d < D; ++d) load_stage(wave + d * WAVES, st[d], d);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        asm volatile("" : "+v"(sraw[t]));
        scale2[t] = as_f16x2(sraw[t] | (sraw[t] << 16));
    }
    if constexpr (XLDS) {
        // this wave's pieces of the copy have landed once only the D * NT weight loads issued after them are outstanding
        constexpr int kOut = D * NT;  // (the scale loads sit between the copy and these: already waited for above)
        __builtin_amdgcn_s_waitcnt(((kOut >> 4) << 14) | 0x0F70 | (kOut & 15));
        __syncthreads();
    }
    int i = 0;
    for (; i + 2 * D <= n; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            consume(st[d]);
            load_stage(wave + (i + d + D) * WAVES, st[d], d);
        }
    }
    const int r = n - (i + D);  // 0 <= r < D tiles remain beyond the D already loaded
    Stage     tail[D > 1 ? D - 1 : 1];
#pragma unroll
    for (int d = 0; d < D - 1; ++d) {
        const int t = i + D + d;
        load_stage(wave + (t < n ? t : n - 1) * WAVES, tail[d], D + d);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) consume(st[d]);
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < r) consume(tail[d]);

    // ---- cross-wave reduction: acc[mt][t][j] = partial y[16*mt + 4g + j][16*(ntile0+t) + c] ----
    constexpr int kPerWave = MT * NT * 256;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                red[wave * kPerWave + ((mt * NT + t) * 16 + 4 * g + j) * 16 + c] = acc[mt][t][j];
    __syncthreads();
    if (ep.act == kActGlu8) {
        // "glu8" column order (8 gate + the 8 matching up columns per 16-column tile): N/2 outputs per row,
        // y[m][8 tile + c] = silu_mul(gate, up) with both operands rounded (+ bias) like the plain epilogue
        Epilogue lin = ep;
        lin.act      = 0;
        for (int o = tid; o < kPerWave; o += WAVES * 64) {
            const int cc = o & 15, rr = (o >> 4) & 15, tt = (o >> 8) % NT, mt = (o >> 8) / NT;
            const int m = 16 * mt + rr;
            if (m < M && cc < 8) {
                float sg = 0.f, su = 0.f;
#pragma unroll
                for (int wv = 0; wv < WAVES; ++wv) {
                    sg += red[wv * kPerWave + o];
                    su += red[wv * kPerWave + o + 8];
                }
                const f16 gv = finish_element(sg, lin, (ntile0 + tt) * 16 + cc);
                const f16 uv = finish_element(su, lin, (ntile0 + tt) * 16 + cc + 8);
                y[(size_t)m * (N >> 1) + (ntile0 + tt) * 8 + cc] = silu_mul_f16(gv, uv);
            }
        }
        return;
    }
    for (int o = tid; o < kPerWave; o += WAVES * 64) {
        const int cc = o & 15, rr = (o >> 4) & 15, tt = (o >> 8) % NT, mt = (o >> 8) / NT;
        const int m = 16 * mt + rr;
        if (m < M) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) s += red[wv * kPerWave + o];
            f16 v = finish_element(s, ep, (ntile0 + tt) * 16 + cc);
            if (ep.residual) v = v + ep.residual[(size_t)m * N + (ntile0 + tt) * 16 + cc];
            y[(size_t)m * N + (ntile0 + tt) * 16 + cc] = v;
        }
    }
}

inline size_t streamk_smem_bytes(int mt, int nt, int waves) { return (size_t)waves * mt * nt * 256 * 4; }

}  // namespace streamk
}  // namespace eetq
