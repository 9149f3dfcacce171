// Medium-batch (2 <= M <= 64) W8A16 stream GEMM launcher; kernel in streamk_kernel.hpp.
// Covers the reference's batched-GEMV range (m <= 4, weightOnlyBatchedGemv/kernelLauncher.cu:165-192) and the
// small-M end of its CUTLASS range, where the weight stream -- not the matrix cores -- bounds the time.
#include <cstdlib>

#include "streamk_kernel.hpp"

namespace eetq {

namespace {

template <int MT, int NT, int WAVES, int D, int OCC, int BITS = 8, bool XLDS = false>
int launch_inst(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                hipStream_t stream)
{
    auto         kern = streamk::streamk_kernel<MT, NT, WAVES, D, OCC, BITS, XLDS>;
    const size_t smem = streamk::streamk_smem_bytes(MT, NT, WAVES) + (XLDS ? (((size_t)M * K * 2 + 1023) & ~(size_t)1023) : 0);
    if (smem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    launch_kernel(kern, dim3(N / (kTileN * NT)), dim3(WAVES * 64), smem, stream, x, w, scales, y, M, N, K, ep);
    return check_hip(hipGetLastError(), "streamk_kernel launch");
}

template <int MT>
int launch_mt(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                hipStream_t stream)
{
    const int KT = K / kTileK;  // every wave must own >= D k tiles
    if (KT >= 32) {
        // every workgroup re-reads the activations (M x K fp16 from L2) for its columns; with two tile rows per
        // workgroup each activation fragment feeds two weight tiles.  Cost per k tile in wave-load units: NT weight loads
        // + 2 activation loads, times the workgroups the busiest CU gets.  N = 5120 (320 tile rows on 256 CUs): M = 8
        // 10.0 -> 7.5 us, K = 13824 23.6 -> 17.2 us; N = 22016: 24.5 -> 17.0 us; N = 4096 stays at NT = 1 (5.0 vs 5.8 us)
        // -- profiles/r01_kbench_streamk_nt.txt
        static const int forced_waves = [] {  // EETQ_AMD_I8_STREAM_WAVES=8 / 16: force the workgroup size (A/B runs)
            const char* e = getenv("EETQ_AMD_I8_STREAM_WAVES");
            return e ? atoi(e) : 0;
        }();
        if constexpr (MT == 1) {
            const int ncu = device_cu_count();
            int       nt  = 1;
            if (N % (2 * kTileN) == 0) {
                const long cost1 = (long)((N / kTileN + ncu - 1) / ncu) * 3;
                const long cost2 = (long)((N / (2 * kTileN) + ncu - 1) / ncu) * 4;
                if (cost2 < cost1) nt = 2;
            }
            // Workgroup size (round 4, profiles/r04_int8_stream_waves_ab.txt, us with 16 / 8 waves): 8-wave workgroups win
            // wherever there are two tile rows per workgroup or more workgroups than CUs, and at M > 4 -- 5120^2 M = 4 7.59 /
            // 7.08, 13824 x 5120 M = 4 17.03 / 15.26, 5120 x 13824 M = 8 15.69 / 14.67, 4096^2 M = 8 5.66 / 5.32; one tile row
            // per workgroup on <= one workgroup per CU at M <= 4 keeps 16 waves (11008 x 4096 M = 2 9.66 / 10.27).  Four tiles
            // in flight per wave instead of two lose (r04_int8_stream_waves_ab2.txt).
            const int  wgs   = N / (kTileN * nt);
            const bool eight = forced_waves ? forced_waves == 8 : (M > 4 || nt == 2 || wgs > ncu);
            // Activations staged ONCE per workgroup in LDS instead of 16 clamped rows from L2 per weight tile (round 4,
            // profiles/r04_stream_xlds_ab.txt; same fragments into the same MFMAs: bit-identical to the register form).  The
            // per-tile cost is then the weight load alone, the copy (M*K*2 bytes) is per workgroup -- one tile row per
            // workgroup becomes affordable again where two leave the CUs unevenly loaded.  Adopted where it measured faster:
            //   2..4 tile rows per CU (7B / 13B up, gate, fused q|k|v): 4096 x 11008 M = 2 10.99 -> 9.05 us, M = 4 11.17 -> 9.47,
            //     M = 8 11.48 -> 10.97; 4096 x 12288 M = 4 11.40 -> 10.12; 5120 x 13824 M = 4 14.18 -> 13.29
            //   <= one tile row per CU, M <= 4: 4096^2 M = 4 5.23 -> 5.06, 11008 x 4096 M = 2 9.68 -> 9.18
            // and left alone elsewhere (4096 x 22016, 8192^2, 5120^2: within +-4 % either way; M*K*2 > 64 KiB: one workgroup per CU).
            // EETQ_AMD_I8_STREAM_XLDS = 0: never; = 1 (default): the rule; >= 1024: everywhere up to that many bytes (A/B runs),
            // EETQ_AMD_I8_STREAM_XLDS_NT = 1 / 2 then forces the tile rows per workgroup.
            static const long xlds_mode = [] {
                const char* e = getenv("EETQ_AMD_I8_STREAM_XLDS");
                return e ? atol(e) : 1L;
            }();
            static const int xlds_nt = [] {
                const char* e = getenv("EETQ_AMD_I8_STREAM_XLDS_NT");
                return e ? atoi(e) : 0;
            }();
            const long xbytes = (long)M * K * 2;
            const int  rows   = N / kTileN;
            int        xnt    = 0;  // 0: register form
            if (xlds_mode >= 1024) {
                if (K % 128 == 0 && xbytes <= xlds_mode) xnt = xlds_nt ? xlds_nt : nt;
            } else if (xlds_mode == 1 && K % 128 == 0 && xbytes <= 64 * 1024) {
                if (rows > 2 * ncu && rows <= 4 * ncu && M <= (K <= 4096 ? 8 : 5)) {
                    const long c1 = (long)((rows + ncu - 1) / ncu) * (8 + M);
                    const long c2 = (long)((rows / 2 + ncu - 1) / ncu) * (16 + M);
                    xnt           = (N % (2 * kTileN) == 0 && c2 <= c1) ? 2 : 1;
                } else if (rows <= ncu && M <= 4) {
                    xnt = 1;
                }
            }
            if (xnt == 2 && N % (2 * kTileN) != 0) xnt = 1;
            if (xnt) {
                const bool e8 = forced_waves ? forced_waves == 8 : (M > 4 || xnt == 2 || N / (kTileN * xnt) > ncu);
                if (xnt == 2)
                    return e8 ? launch_inst<MT, 2, 8, 2, 4, 8, true>(x, w, scales, ep, y, M, N, K, stream)
                              : launch_inst<MT, 2, 16, 2, 4, 8, true>(x, w, scales, ep, y, M, N, K, stream);
                return e8 ? launch_inst<MT, 1, 8, 2, 4, 8, true>(x, w, scales, ep, y, M, N, K, stream)
                          : launch_inst<MT, 1, 16, 2, 4, 8, true>(x, w, scales, ep, y, M, N, K, stream);
            }
            if (nt == 2)
                return eight ? launch_inst<MT, 2, 8, 2, 4>(x, w, scales, ep, y, M, N, K, stream)
                             : launch_inst<MT, 2, 16, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
            if (eight) return launch_inst<MT, 1, 8, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
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
        const char* e = getenv("EETQ_AMD_I4_STREAM_WAVES");
        return e ? atoi(e) : 0;
    }();
    if (KT >= 32) {
        const int ncu = device_cu_count();
        int       nt  = 1;
        if (N % (2 * kTileN) == 0) {
            const long cost1 = (long)((N / kTileN + ncu - 1) / ncu) * 5;
            const long cost2 = (long)((N / (2 * kTileN) + ncu - 1) / ncu) * 6;
            if (cost2 < cost1) nt = 2;
        }
        // 8-wave workgroups when there are enough of them to give every CU one (round 4, profiles/r04_int4_stream_waves_ab.txt:
        // 4096 x 11008 M = 4 8.99 -> 8.34 us, 5120 x 13824 10.93 -> 10.44, 11008 x 4096 8.46 -> 8.10; the int4 kernels are issue-
        // and latency-bound and smaller workgroups overlap better); 16 waves when two tile rows per workgroup leave fewer
        // workgroups than CUs (N = 5120: 160 workgroups, 5.84 vs 6.29 us with 8 waves)
        const int  wgs   = N / (kTileN * nt);
        const bool eight = forced_waves ? forced_waves == 8 : wgs >= ncu;
        // Activations staged once per workgroup in LDS (see launch_mt; four activation loads per weight load otherwise).
        // profiles/r04_stream_xlds_i4_ab.txt: wins at K = 4096 with 2..4 tile rows per CU and M <= 4, one tile row per workgroup
        // (4096 x 11008 M = 2 8.32 -> 6.79 us, M = 4 8.37 -> 7.39; 4096 x 12288 8.30 -> 6.96 / 8.34 -> 7.51), and with one
        // workgroup per CU at M >= 8 (4096^2 M = 16 6.08 -> 5.29, M = 12 5.36 -> 4.93, M = 8 4.73 -> 4.63); neutral or slower
        // elsewhere (K = 5120, 8192, 11008, 13824; N = 22016).  EETQ_AMD_I4_STREAM_XLDS = 0: never; = 1 (default): that rule;
        // >= 1024: everywhere up to that many bytes of M*K*2, EETQ_AMD_I4_STREAM_XLDS_NT = tile rows per workgroup (A/B runs).
        static const long xlds_mode = [] {
            const char* e = getenv("EETQ_AMD_I4_STREAM_XLDS");
            return e ? atol(e) : 1L;
        }();
        static const int xlds_nt = [] {
            const char* e = getenv("EETQ_AMD_I4_STREAM_XLDS_NT");
            return e ? atoi(e) : 0;
        }();
        const long xbytes = (long)M * K * 2;
        const int  rows   = N / kTileN;
        int        xnt    = 0;
        if (xlds_mode >= 1024) {
            if (xbytes <= xlds_mode) xnt = xlds_nt ? xlds_nt : nt;
        } else if (xlds_mode == 1 && K <= 4096) {
            if (rows > 2 * ncu && rows <= 4 * ncu && M <= 4) xnt = 1;
            if (rows <= ncu && M >= 8 && xbytes <= 144 * 1024) xnt = 1;
        }
        if (xnt == 2 && N % (2 * kTileN) != 0) xnt = 1;
        if (xnt) {
            const bool e8 = forced_waves ? forced_waves == 8 : N / (kTileN * xnt) >= ncu;
            if (xnt == 2) return e8 ? launch_inst<1, 2, 8, 2, 2, 4, true>(x, w, scales, ep, y, M, N, K, stream)
                                    : launch_inst<1, 2, 16, 2, 2, 4, true>(x, w, scales, ep, y, M, N, K, stream);
            return e8 ? launch_inst<1, 1, 8, 2, 2, 4, true>(x, w, scales, ep, y, M, N, K, stream)
                      : launch_inst<1, 1, 16, 2, 2, 4, true>(x, w, scales, ep, y, M, N, K, stream);
        }
        if (nt == 2) return eight ? launch_inst<1, 2, 8, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream)
                                  : launch_inst<1, 2, 16, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
        return eight ? launch_inst<1, 1, 8, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream)
                     : launch_inst<1, 1, 16, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
    }
    if (KT >= 16) return launch_inst<1, 1, 8, 2, 2, 4>(x, w, scales, ep, y, M, N, K, stream);
    if (KT >= 4) return launch_inst<1, 1, 4, 1, 1, 4>(x, w, scales, ep, y, M, N, K, stream);
    return launch_inst<1, 1, 1, 1, 1, 4>(x, w, scales, ep, y, M, N, K, stream);
}

}  // namespace

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
