// W8A16 decode GEMV launcher (M = 1 in AUTO dispatch; the kernel also serves M <= 4 on request), kernels in
// gemv_kernel.hpp: wave64 dot2 + lane-swap reduction, no MFMA.
//
// Replaces the reference's weight_only_batched_gemv (csrc/weightOnlyBatchedGemv/kernel.h:294-468, launched from
// kernelLauncher.cu:165-192 for m <= 4).  Same math up to summation order, except that partial sums are kept in
// fp32 throughout (the CUDA kernel accumulates 32 products per thread in fp16, kernel.h:325-329):
//   y[m][n] = fp16( sum_k fp32(x[m][k]) * fp32(fp16(q[k][n] * s[n])) )
// Design notes (HBM-bound, K*N weight bytes read exactly once): DESIGN.md section 4.1 and gemv_kernel.hpp.
#include "gemv_kernel.hpp"

namespace eetq {

namespace {

using gemv::gemv_kernel;

template <int M, int WAVES, int D, bool EXACT, bool XREG, int XV, int OCC, int NORM = 0>
int launch_inst(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K,
                hipStream_t stream, Prologue pro = Prologue{})
{
    if constexpr (!NORM && !XREG && M == 1) {
        if (pro.gamma) return launch_inst<M, WAVES, D, EXACT, XREG, XV, OCC, 1>(x, w, scales, ep, y, N, K, stream, pro);
        if (pro.up) return launch_inst<M, WAVES, D, EXACT, XREG, XV, OCC, 2>(x, w, scales, ep, y, N, K, stream, pro);
    }
    const size_t smem = gemv::gemv_smem_bytes(M, K, WAVES, XREG);
    auto go = [&](auto kern, std::atomic<unsigned long long>& opted) {
        if (smem > 64 * 1024) {
            int st = opt_in_large_lds(kern, opted);
            if (st != EETQ_OK) return st;
        }
        launch_kernel(kern, dim3(N / kTileN), dim3(WAVES * 64), smem, stream, x, w, scales, y, N, K, pro.gamma ? pro.gamma : pro.up, ep.bias,
                      ep.residual, ep.act, pro.eps);
        return check_hip(hipGetLastError(), "gemv_kernel launch");
    };
    if constexpr (M == 1 && !NORM) {
        // the plain projection (no bias, residual or activation) has its own instantiation of every M = 1 kernel form: same
        // arithmetic, same bits, no run-time epilogue (gemv_kernel.hpp: PLAIN)
        if (!ep.bias && !ep.residual && ep.act == 0) {
            static std::atomic<unsigned long long> opted_plain{0};
            return go(gemv_kernel<M, WAVES, D, EXACT, XREG, XV, OCC, NORM, 8, true>, opted_plain);
        }
    }
    static std::atomic<unsigned long long> opted{0};
    return go(gemv_kernel<M, WAVES, D, EXACT, XREG, XV, OCC, NORM>, opted);
}

// LDS-staged activations: pick the number of 16-byte x loads per thread at compile time (no conditional loads)
template <int M, int WAVES, int D, bool EXACT, int OCC>
int launch_lds(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K,
                hipStream_t stream, Prologue pro = Prologue{})
{
    const int xvecs = M * K / 8, threads = WAVES * 64;
    const int need  = (xvecs + threads - 1) / threads;
    if (gemv::gemv_smem_bytes(M, K, WAVES, false) > 160 * 1024 || need > 8)
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] GEMV: M*K too large for LDS staging");
    if (need <= 1) return launch_inst<M, WAVES, D, EXACT, false, 1, OCC>(x, w, scales, ep, y, N, K, stream, pro);
    if (need <= 2) return launch_inst<M, WAVES, D, EXACT, false, 2, OCC>(x, w, scales, ep, y, N, K, stream, pro);
    if (need <= 4) return launch_inst<M, WAVES, D, EXACT, false, 4, OCC>(x, w, scales, ep, y, N, K, stream, pro);
    return launch_inst<M, WAVES, D, EXACT, false, 8, OCC>(x, w, scales, ep, y, N, K, stream, pro);
}

// 8-column units (gemv_half_kernel), M = 1: when they put fewer bytes on the busiest CU than whole tile rows do
template <int XV, int NORM = 0>
int launch_half_xv(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K, hipStream_t stream,
                   Prologue pro)
{
    if constexpr (!NORM) {
        if (pro.gamma) return launch_half_xv<XV, 1>(x, w, scales, ep, y, N, K, stream, pro);
        if (pro.up) return launch_half_xv<XV, 2>(x, w, scales, ep, y, N, K, stream, pro);
    }
    auto         kern = gemv::gemv_half_kernel<8, 2, XV, 8, NORM>;
    const size_t smem = gemv::gemv_half_smem_bytes(K, 8);
    if (smem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    launch_kernel(kern, dim3(N / 8), dim3(8 * 64), smem, stream, x, w, scales, y, N, K, pro.gamma ? pro.gamma : pro.up, ep.bias, ep.residual,
                  ep.act, pro.eps);
    return check_hip(hipGetLastError(), "gemv_half_kernel launch");
}

// 8 + 8 + 4 (gemv_mixed_kernel): N splits into exactly N / CUs = 8a + 4 columns per CU -- N = 5120 on 256 CUs: two 8-column
// units and one 4-column unit each, 768 workgroups resident at once, instead of 640 eight-column units of which a quarter of
// the CUs get three.  EETQ_AMD_GEMV_MIXED=0 keeps the 8-column units (A/B runs).
bool mixed_units_pay(int N, int K)
{
    static const bool allowed = [] {
        const char* e = tuning_env("EETQ_AMD_GEMV_MIXED");
        return !(e && e[0] == '0');
    }();
    const int KT = K / kTileK, ncu = device_cu_count();
    if (!allowed || KT % 4 || KT / 4 < 16 || K > 32768 || N % (4 * ncu)) return false;  // groups of 4 k tiles, >= 2 per wave
    const int per = N / ncu;
    return per % 8 == 4 && per >= 12 && per <= 28;
}

template <int XV, int NORM = 0>
int launch_mixed_xv(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K, hipStream_t stream,
                    Prologue pro)
{
    if constexpr (!NORM) {
        if (pro.gamma) return launch_mixed_xv<XV, 1>(x, w, scales, ep, y, N, K, stream, pro);
        if (pro.up) return launch_mixed_xv<XV, 2>(x, w, scales, ep, y, N, K, stream, pro);
    }
    auto         kern = gemv::gemv_mixed_kernel<8, 2, XV, 8, NORM>;
    const size_t smem = gemv::gemv_half_smem_bytes(K, 8);
    if (smem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    const int ncu = device_cu_count(), per = N / ncu;
    const int n8 = ncu * (per / 8), n4 = (N - 8 * n8) / 4;
    launch_kernel(kern, dim3(n8 + n4), dim3(8 * 64), smem, stream, x, w, scales, y, N, K, pro.gamma ? pro.gamma : pro.up, ep.bias, ep.residual,
                  ep.act, pro.eps);  // (the kernel derives n8 from N and the grid)
    return check_hip(hipGetLastError(), "gemv_mixed_kernel launch");
}

bool half_units_pay(int N, int K)
{
    const int KT = K / kTileK;
    if (KT % 2 || KT < 32 || K > 32768) return false;  // pairs of k tiles, >= 2 pairs per wave in flight, x fits in LDS
    static const int forced = [] {  // EETQ_AMD_I8_UNITS=1: column units wherever they can run, =0: never (A/B runs)
        const char* e = tuning_env("EETQ_AMD_I8_UNITS");
        return e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1)) : -1;
    }();
    if (forced >= 0) return forced == 1;
    // Measured (profiles/r01_kbench_gemv.txt): worth it when whole tile rows leave CUs idle (rows <= CUs / 2) or put a
    // second workgroup on only a few CUs (N = 5120: 320 rows on 256 CUs, K = 13824 14.5 -> 13.2 us, K = 5120 6.24 -> 6.04);
    // a wash or a small loss elsewhere (N = 6144, 13824, 4096).
    const int ncu = device_cu_count(), rows = N / kTileN;
    return 2 * rows <= ncu || (rows > ncu && 10 * rows <= 13 * ncu);
}

int launch_half(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K, hipStream_t stream,
                Prologue pro)
{
    const int need = (K / 8 + 511) / 512;  // 16-byte activation loads per thread
    if (mixed_units_pay(N, K)) {
        if (need <= 2) return launch_mixed_xv<2>(x, w, scales, ep, y, N, K, stream, pro);
        if (need <= 4) return launch_mixed_xv<4>(x, w, scales, ep, y, N, K, stream, pro);
        return launch_mixed_xv<8>(x, w, scales, ep, y, N, K, stream, pro);
    }
    if (need <= 2) return launch_half_xv<2>(x, w, scales, ep, y, N, K, stream, pro);
    if (need <= 4) return launch_half_xv<4>(x, w, scales, ep, y, N, K, stream, pro);
    return launch_half_xv<8>(x, w, scales, ep, y, N, K, stream, pro);
}

template <int M>
int launch_m(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K,
                hipStream_t stream, Prologue pro)
{
    if constexpr (M != 1) {
        if (pro.gamma || pro.up)
            return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] the activation prologues are implemented for M = 1");
    }
    const int KT = K / kTileK;
    if constexpr (M == 1) {
        // N = 5120: 320 tile rows on 256 CUs -> 640 half rows, 3 instead of 4 eight-column units on the busiest CU
        if (ep.act != kActGlu8 && half_units_pay(N, K)) return launch_half(x, w, scales, ep, y, N, K, stream, pro);
    }
    // Tuned on MI355X with tools/kbench (profiles/r01_kbench_gemv.txt):
    if constexpr (M == 1) {
        // K = 4096 with more than two tile rows per CU (the 7B up / gate projections and their fused forms): the 8-wave generic form
        // beats the straight-line 16-wave one -- 4096 x 11008 9.57 -> 8.87 us, 4096 x 22016 16.79 -> 15.52, 4096 x 12288 9.98 -> 9.48
        // (profiles/r04_i8_gemv_k4096_ab.txt); at one tile row per CU (4096^2) it loses (5.10 vs 4.69).
        // EETQ_AMD_I8_GEMV_K4096 = 16 / 82 / 88: force the straight-line form / 8 waves generic / 8 waves x 8 tiles (A/B runs)
        static const int forced64 = [] {
            const char* e = tuning_env("EETQ_AMD_I8_GEMV_K4096");
            return e ? atoi(e) : 0;
        }();
        if (KT == 64 && forced64 == 88) return launch_lds<M, 8, 8, true, 4>(x, w, scales, ep, y, N, K, stream, pro);
        if (KT == 64 && (forced64 == 82 || (forced64 == 0 && N / kTileN > 2 * device_cu_count())))
            return launch_lds<M, 8, 2, false, 8>(x, w, scales, ep, y, N, K, stream, pro);
    }
    if (KT == 64) {  // K = 4096: 16 waves x 4 tiles, straight-line; <= 64 VGPRs so two workgroups fit a CU
        // (register-resident activations only for one row: at M = 2 they cost 32 more VGPRs and the second workgroup per CU --
        // 14.5 instead of 5.9 us at N = 4096, tools/path_compare.py)
        if constexpr (M == 1) {
            if (!pro.gamma && !pro.up) return launch_inst<M, 16, 4, true, true, 1, 8>(x, w, scales, ep, y, N, K, stream);
            return launch_lds<M, 16, 4, true, 4>(x, w, scales, ep, y, N, K, stream, pro);  // the prologue needs x in LDS
        } else {
            return launch_lds<M, 16, 4, true, 4>(x, w, scales, ep, y, N, K, stream);
        }
    }
    // generic K: every wave must own >= D tiles; D = 2 in flight per wave won at K = 11008
    if constexpr (M == 1) {
        // Workgroup size (round 4, profiles/r04_i8_gemv_waves_ab.txt, us with 16 / 8 waves): with more than two tile rows per CU
        // 8-wave workgroups overlap one another better -- 5120 x 13824 13.91 / 13.37, 5120 x 15360 14.64 / 13.89, 5120 x 27648
        // 23.16 / 22.50 -- with one tile row per CU they starve it (11008 x 4096 9.37 / 10.28); 8192^2 is a wash.
        // EETQ_AMD_I8_GEMV_WAVES = 16 / 82 / 84 forces 16 waves / 8 waves with 2 / 4 tiles in flight (A/B runs)
        static const int forced = [] {
            const char* e = tuning_env("EETQ_AMD_I8_GEMV_WAVES");
            return e ? atoi(e) : 0;
        }();
        if (KT >= 32 && forced == 84) return launch_lds<M, 8, 4, false, 8>(x, w, scales, ep, y, N, K, stream, pro);
        if (KT >= 32 && (forced == 82 || (forced == 0 && N / kTileN > 2 * device_cu_count())))
            return launch_lds<M, 8, 2, false, 8>(x, w, scales, ep, y, N, K, stream, pro);
    }
    if (KT >= 32) return launch_lds<M, 16, 2, false, (M == 1 ? 8 : 4)>(x, w, scales, ep, y, N, K, stream, pro);
    if (KT >= 16) return launch_lds<M, 8, 2, false, 2>(x, w, scales, ep, y, N, K, stream, pro);
    if (KT >= 4) return launch_lds<M, 4, 1, false, 1>(x, w, scales, ep, y, N, K, stream, pro);
    return launch_lds<M, 1, 1, false, 1>(x, w, scales, ep, y, N, K, stream, pro);
}

// ---- int4 tiles (W4A16): the same kernel template with BITS = 4 (128 k per tile, 32 k per lane) ----
template <int M, int WAVES, int D, bool EXACT, bool XREG, int XV, int OCC>
int launch_inst_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K, hipStream_t stream)
{
    auto         kern = gemv_kernel<M, WAVES, D, EXACT, XREG, XV, OCC, 0, 4>;
    const size_t smem = gemv::gemv_smem_bytes(M, K, WAVES, XREG);
    if (smem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    launch_kernel(kern, dim3(N / kTileN), dim3(WAVES * 64), smem, stream, x, w, scales, y, N, K, (const f16*)nullptr, ep.bias, ep.residual, ep.act, 0.f);
    return check_hip(hipGetLastError(), "gemv_kernel (int4) launch");
}

template <int M, int WAVES, int D, int OCC>
int launch_lds_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K, hipStream_t stream)
{
    const int xvecs = M * K / 8, threads = WAVES * 64;
    const int need  = (xvecs + threads - 1) / threads;
    if (gemv::gemv_smem_bytes(M, K, WAVES, false) > 160 * 1024 || need > 8)
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] W4A16 GEMV: M*K too large for LDS staging");
    if (need <= 1) return launch_inst_i4<M, WAVES, D, false, false, 1, OCC>(x, w, scales, ep, y, N, K, stream);
    if (need <= 2) return launch_inst_i4<M, WAVES, D, false, false, 2, OCC>(x, w, scales, ep, y, N, K, stream);
    if (need <= 4) return launch_inst_i4<M, WAVES, D, false, false, 4, OCC>(x, w, scales, ep, y, N, K, stream);
    return launch_inst_i4<M, WAVES, D, false, false, 8, OCC>(x, w, scales, ep, y, N, K, stream);
}

// 8-column units on int4 tiles (gemv_half_kernel<..., BITS = 4>): the same rule as for int8 tiles -- whole tile rows would leave CUs
// idle or put a second workgroup on only a few (N = 5120 on 256 CUs) -- with pairs of 128-deep k tiles
template <int XV>
int launch_half_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K, hipStream_t stream)
{
    auto         kern = gemv::gemv_half_kernel<8, 2, XV, 4, 0, 4>;
    const size_t smem = gemv::gemv_half_smem_bytes(K, 8);
    if (smem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    launch_kernel(kern, dim3(N / 8), dim3(8 * 64), smem, stream, x, w, scales, y, N, K, (const f16*)nullptr, ep.bias, ep.residual, ep.act, 0.f);
    return check_hip(hipGetLastError(), "gemv_half_kernel (int4) launch");
}

bool half_units_pay_i4(int N, int K)
{
    const int KT = K / 128;
    if (KT % 2 || KT < 32 || K > 32768) return false;  // pairs of k tiles, >= 2 pairs per wave in flight (8 waves), x fits in LDS
    static const int forced = [] {  // EETQ_AMD_I4_UNITS=1: 8-column units wherever they can run, =0: never (A/B runs)
        const char* e = tuning_env("EETQ_AMD_I4_UNITS");
        return e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1)) : -1;
    }();
    if (forced >= 0) return forced == 1;
    // Measured (profiles/r04_int4_units_ab.txt, us with / without): deep K wins everywhere -- 11008 x 4096 7.45 / 8.40, 5120 x 13824
    // 10.14 / 11.36, 13824 x 5120 10.82 / 13.3, 5120^2 5.66 / 6.10 -- the int4 GEMV is issue- and latency-bound and
    // 8-wave workgroups of 8 columns overlap better than 16-wave workgroups of 16; K = 4096 loses (4.53 / 4.06: the
    // straight-line register-resident form exists there, as at K = 8192)
    if (KT != 64 && KT >= 40) return true;
    const int ncu = device_cu_count(), rows = N / kTileN;
    return KT != 32 && KT != 64 && (2 * rows <= ncu || (rows > ncu && 10 * rows <= 13 * ncu));
}

template <int M>
int launch_m_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int N, int K, hipStream_t stream)
{
    const int KT = K / 128;
    if constexpr (M == 1) {
        if (ep.act == 0 && half_units_pay_i4(N, K)) {
            const int need = (K / 8 + 511) / 512;  // 16-byte activation loads per thread
            if (need <= 2) return launch_half_i4<2>(x, w, scales, ep, y, N, K, stream);
            if (need <= 4) return launch_half_i4<4>(x, w, scales, ep, y, N, K, stream);
            return launch_half_i4<8>(x, w, scales, ep, y, N, K, stream);
        }
    }
    if constexpr (M == 1) {
        // K = 4096 with more than two tile rows per CU: the 8-wave generic form, as for int8 tiles -- 4096 x 11008 8.0 -> 7.05 us,
        // 4096 x 22016 13.6 -> 11.8, 4096 x 12288 8.2 -> 7.2; 4096^2 keeps the straight-line form (4.04 vs 4.12)
        // (profiles/r04_i4_gemv_k4096_ab.txt).  EETQ_AMD_I4_GEMV_K4096 = 16 / 82 / 84 forces a form (A/B runs)
        static const int forced32 = [] {
            const char* e = tuning_env("EETQ_AMD_I4_GEMV_K4096");
            return e ? atoi(e) : 0;
        }();
        if (KT == 32 && forced32 == 84) return launch_lds_i4<M, 8, 4, 2>(x, w, scales, ep, y, N, K, stream);
        if (KT == 32 && (forced32 == 82 || (forced32 == 0 && N / kTileN > 2 * device_cu_count())))
            return launch_lds_i4<M, 8, 2, 2>(x, w, scales, ep, y, N, K, stream);
    }
    if constexpr (M <= 2) {
        // whole tile row in flight, activations straight to registers (the K = 4096 / 8192 decode shapes)
        if (KT == 32) return launch_inst_i4<M, 16, 2, true, true, 1, 4>(x, w, scales, ep, y, N, K, stream);
        if (KT == 64) return launch_inst_i4<M, 16, 4, true, true, 1, 2>(x, w, scales, ep, y, N, K, stream);
    }
    // (round 4: straight-line instantiations for the 13B widths -- K = 5120 as 8 waves x 5 tiles, K = 13824 as 12 x 9 -- were
    // measured and dropped: 5120 x 13824 11.9 vs 11.4 us with the generic loop.  The int4 GEMV is not bound by the shape of its
    // load stream but by ~2.1 VALU operations per weight that overlap poorly with it: profiles/r04_int4_m1.jsonl)
    if (KT >= 64) return launch_lds_i4<M, 16, 4, 4>(x, w, scales, ep, y, N, K, stream);  // every wave owns >= 4 tiles
    if (KT >= 32) return launch_lds_i4<M, 16, 2, 4>(x, w, scales, ep, y, N, K, stream);
    if (KT >= 16) return launch_lds_i4<M, 8, 2, 2>(x, w, scales, ep, y, N, K, stream);
    if (KT >= 4) return launch_lds_i4<M, 4, 1, 1>(x, w, scales, ep, y, N, K, stream);
    return launch_lds_i4<M, 1, 1, 1>(x, w, scales, ep, y, N, K, stream);
}

}  // namespace

int launch_gemv(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                hipStream_t stream, Prologue pro)
{
    switch (M) {
        case 1: return launch_m<1>(x, w, scales, ep, y, N, K, stream, pro);
        case 2: return launch_m<2>(x, w, scales, ep, y, N, K, stream, pro);
        case 3: return launch_m<3>(x, w, scales, ep, y, N, K, stream, pro);
        case 4: return launch_m<4>(x, w, scales, ep, y, N, K, stream, pro);
        default: return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] GEMV path only supports M <= 4");
    }
}

// ---- grouped launch (gemv_grouped_kernel): problems of equal K, at most kMaxGroup per dispatch -------------------------------
namespace {
template <int WAVES, int D, bool EXACT, bool XREG, int XV, int OCC>
int launch_grouped_inst(const gemv::GroupedArgs& g, int K, int rows, hipStream_t stream)
{
    auto         kern = gemv::gemv_grouped_kernel<WAVES, D, EXACT, XREG, XV, OCC>;
    const size_t smem = gemv::gemv_smem_bytes(1, K, WAVES, XREG);
    if (smem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    launch_kernel(kern, dim3(rows), dim3(WAVES * 64), smem, stream, g, K);
    return check_hip(hipGetLastError(), "gemv_grouped_kernel launch");
}
}  // namespace

bool gemv_grouped_supports(int K) { return K % kTileK == 0 && K / kTileK >= 32 && K <= 32768; }

// (body choice and what it means for bit identity with separate launches: see inside)
int launch_gemv_grouped(const gemv::GroupedArgs& g, int K, int rows, hipStream_t stream)
{
    if (!gemv_grouped_supports(K)) return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] grouped GEMV: K must be a multiple of 64 in [2048, 32768]");
    // Body: with more than two tile rows per CU in the dispatch the 8-wave generic form (two tiles in flight per wave, x staged in
    // LDS), as for single launches with many tile rows -- 8 x (4096 x 4096): 2.83 -> 2.64 us per problem (0.80 of 8 TB/s), 16
    // problems 2.50 (0.84) (profiles/r04_grouped_waves_ab.txt); smaller dispatches keep the 16-wave forms (K = 4096: the
    // straight-line register-resident one, bit-identical to a separate 4096 x 4096 launch).  A grouped result is always the
    // same bits from call to call; against separate launches it is tier A (another summation order) unless both take the same
    // body.  EETQ_AMD_GROUPED_WAVES = 16 / 8 forces a body (A/B runs).
    static const int forced = [] {
        const char* e = tuning_env("EETQ_AMD_GROUPED_WAVES");
        return e ? atoi(e) : 0;
    }();
    const bool eight = forced ? forced == 8 : rows > 2 * device_cu_count();
    const int  need8 = (K / 8 + 511) / 512;  // 16-byte activation loads per thread, 8 waves
    if (eight) {
        if (need8 <= 1) return launch_grouped_inst<8, 2, false, false, 1, 8>(g, K, rows, stream);
        if (need8 <= 2) return launch_grouped_inst<8, 2, false, false, 2, 8>(g, K, rows, stream);
        if (need8 <= 4) return launch_grouped_inst<8, 2, false, false, 4, 8>(g, K, rows, stream);
        return launch_grouped_inst<8, 2, false, false, 8, 8>(g, K, rows, stream);
    }
    if (K == 4096) return launch_grouped_inst<16, 4, true, true, 1, 8>(g, K, rows, stream);
    const int need = (K / 8 + 1023) / 1024;  // 16-byte activation loads per thread
    if (need <= 1) return launch_grouped_inst<16, 2, false, false, 1, 8>(g, K, rows, stream);
    if (need <= 2) return launch_grouped_inst<16, 2, false, false, 2, 8>(g, K, rows, stream);
    return launch_grouped_inst<16, 2, false, false, 4, 8>(g, K, rows, stream);
}

}  // namespace eetq

namespace eetq {
int launch_gemv_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                   hipStream_t stream)
{
    switch (M) {
        case 1: return launch_m_i4<1>(x, w, scales, ep, y, N, K, stream);
        case 2: return launch_m_i4<2>(x, w, scales, ep, y, N, K, stream);
        case 3: return launch_m_i4<3>(x, w, scales, ep, y, N, K, stream);
        case 4: return launch_m_i4<4>(x, w, scales, ep, y, N, K, stream);
        default: return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] W4A16 GEMV path only supports M <= 4");
    }
}
}  // namespace eetq
