// extern "C" boundary of libeetq_amd.so (declarations + reference citations: include/eetq_amd.h).
// Stateless apart from (a) the thread-local last-error string and (b) a per-device scratch buffer for the
// quantiser's column maxima (N floats), so the reference's "no workspace argument" signature is kept.
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "gemv_kernel.hpp"

namespace eetq {

// dropped-step counters of the decode kernels (attn_decode.hip, norm_rope.hip)
int attn_dropped_steps(unsigned* count, bool reset);
int rope_dropped_steps(unsigned* count, bool reset);

namespace {
thread_local std::string g_last_error;

struct Scratch {
    float* ptr   = nullptr;
    size_t elems = 0;
};
std::mutex g_scratch_mutex;
Scratch    g_scratch[64];
}  // namespace

// ---- optional per-dispatch timing (bench.py): every kernel launched on this thread between eetq_prof_begin and
// eetq_prof_end gets a start/stop event pair attached to its dispatch packet (begin/end timestamps of the kernel
// itself, the same clock rocprofv3 --kernel-trace reports).
namespace {
thread_local std::vector<hipEvent_t> g_prof_events;  // 2 per launch
thread_local size_t                  g_prof_used  = 0;
thread_local bool                    g_prof_armed = false;
}  // namespace

ProfEvents next_prof_events()
{
    ProfEvents ev;
    if (g_prof_armed && g_prof_used + 2 <= g_prof_events.size()) {
        ev.start = g_prof_events[g_prof_used];
        ev.stop  = g_prof_events[g_prof_used + 1];
        g_prof_used += 2;
    }
    return ev;
}

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const std::string& msg)
{
    set_error(msg);
    return code;
}

int check_hip(hipError_t e, const char* what)
{
    if (e == hipSuccess) return EETQ_OK;
    (void)hipGetLastError();  // clear the sticky launch error
    return fail(EETQ_ERR_HIP, std::string("[eetq_amd] HIP error: ") + hipGetErrorString(e) + " in " + what);
}

static int colmax_scratch(size_t n, float** out)
{
    int dev = 0;
    EETQ_TRY_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    Scratch&                    s = g_scratch[dev & 63];
    if (s.elems < n) {
        if (s.ptr) EETQ_TRY_HIP(hipFree(s.ptr));
        s.ptr   = nullptr;
        s.elems = 0;
        size_t want = n < 65536 ? 65536 : n;
        EETQ_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&s.ptr), want * sizeof(float)));
        s.elems = want;
    }
    *out = s.ptr;
    return EETQ_OK;
}

static int check_gemm_args(const void* x, const void* w, const void* s, const void* y, int M, int N, int K)
{
    EETQ_REQUIRE(x && w && s && y, "null pointer");
    EETQ_REQUIRE(M >= 1 && N >= 1 && K >= 1, "invalid GEMM shape");
    // the reference throws "Temp assertion: k must be multiple of threadblockK" (fpA_intB_gemm_template.h:139-142)
    EETQ_REQUIRE(K % 64 == 0, "k must be a multiple of 64");
    EETQ_REQUIRE(N % 16 == 0, "n must be a multiple of 16");
    EETQ_REQUIRE(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) % 16 == 0 && (uintptr_t)s % 2 == 0,
                 "x, weight and y must be 16-byte aligned");
    return EETQ_OK;
}

// RAII device buffer for the blocking host-pointer variants
struct DevBuf {
    void* p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) { return check_hip(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc"); }
};

}  // namespace eetq

using namespace eetq;

int eetq::device_cu_count()
{
    static int cached[64] = {0};
    int        dev        = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cached[dev & 63];
    if (!c) {
        int v = 0;
        c     = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return c;
}

namespace eetq {
int launch_greedy_handover(const f16* logits, long row_stride, int vocab, int batch, int64_t* out_buf, long out_stride, int out_cols,
                           int64_t* s_idx, int64_t* s_tok, int64_t* s_pos, hipStream_t stream);  // norm_rope.hip
}

extern "C" {

const char* eetq_last_error(void) { return g_last_error.c_str(); }

const char* eetq_version(void) { return "eetq_amd 0.1.0 gfx950"; }

int eetq_prof_begin(int max_launches)
{
    EETQ_REQUIRE(max_launches > 0 && max_launches <= (1 << 20), "invalid launch count");
    while (g_prof_events.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        EETQ_TRY_HIP(hipEventCreate(&e));
        g_prof_events.push_back(e);
    }
    g_prof_used  = 0;
    g_prof_armed = true;
    return EETQ_OK;
}

int eetq_prof_end(float* durations_us, int capacity, int* count)
{
    g_prof_armed = false;
    EETQ_TRY_HIP(hipDeviceSynchronize());
    const int n = (int)(g_prof_used / 2);
    if (count) *count = n;
    for (int i = 0; i < n && i < capacity && durations_us; ++i) {
        float ms = 0.f;
        EETQ_TRY_HIP(hipEventElapsedTime(&ms, g_prof_events[2 * i], g_prof_events[2 * i + 1]));
        durations_us[i] = ms * 1000.f;
    }
    g_prof_used = 0;
    return EETQ_OK;
}

int eetq_diag_stream_read(const void* p, size_t bytes, void* sink, void* stream)
{
    return launch_stream_read(p, bytes, static_cast<unsigned*>(sink), static_cast<hipStream_t>(stream));
}

int eetq_diag_empty(void* sink, int grid, int block, void* stream)
{
    return launch_empty(static_cast<unsigned*>(sink), grid, block, static_cast<hipStream_t>(stream));
}

int eetq_diag_clock_stamp(unsigned long long* out, int grid, void* stream)
{
    return launch_clock_stamp(out, grid, static_cast<hipStream_t>(stream));
}

int eetq_diag_stream_plan(int bits, int M, int N, int K, int cus, int* form, int* tile_rows, int* waves)
{
    EETQ_REQUIRE(form && tile_rows && waves, "eetq_diag_stream_plan: null output pointer");
    const int ncu = cus > 0 ? cus : device_cu_count();
    if (stream_plan_query(bits, M, N, K, ncu, form, tile_rows, waves) != 0)
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] eetq_diag_stream_plan: bits 4 / 8, 1 <= M <= 16, N % 16 == 0, K % 64 (128) == 0");
    return EETQ_OK;
}

int eetq_device_supported(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return std::string(prop.gcnArchName).rfind("gfx950", 0) == 0 ? 1 : 0;
}

int eetq_quantize_i8_ws(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed, int layout,
                        void* scales, float* workspace, size_t workspace_floats, void* stream)
{
    if (!workspace) {
        workspace_floats = quantize_workspace_floats(K, N);
        int st           = colmax_scratch(workspace_floats, &workspace);
        if (st != EETQ_OK) return st;
    }
    return launch_quantize(w, w_dtype, K, N, q_raw, q_packed, layout, scales, workspace, workspace_floats,
                           static_cast<hipStream_t>(stream));
}

// The size-less entry of ABI revision 1: a caller-provided workspace is only known to hold the N floats that revision asked
// for, so it takes the N-float route (zero fill + atomicMax maxima); NULL uses the library's buffer and the fast route.
int eetq_quantize_i8(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed, int layout,
                     void* scales, float* workspace, void* stream)
{
    return eetq_quantize_i8_ws(w, w_dtype, K, N, q_raw, q_packed, layout, scales, workspace, workspace ? N : 0, stream);
}

int eetq_abi_version(void) { return EETQ_AMD_ABI_VERSION; }

size_t eetq_quantize_workspace_floats(size_t K, size_t N) { return quantize_workspace_floats(K, N); }

int eetq_release_stream_workspace(void* stream) { return release_splitk_region(static_cast<hipStream_t>(stream)); }

int eetq_release_workspace(size_t* bytes_freed)
{
    size_t freed = 0;
    int    st    = release_splitk_workspace(&freed);
    if (st != EETQ_OK) return st;
    st = release_w4a16_workspace(&freed);
    if (st != EETQ_OK) return st;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        int keep = 0;
        (void)hipGetDevice(&keep);
        for (int d = 0; d < 64; ++d) {
            Scratch& s = g_scratch[d];
            if (s.ptr && hipSetDevice(d) == hipSuccess) {
                (void)hipDeviceSynchronize();
                (void)hipFree(s.ptr);
                freed += s.elems * sizeof(float);
            }
            s = Scratch{};
        }
        (void)hipSetDevice(keep);
    }
    if (bytes_freed) *bytes_freed = freed;
    return EETQ_OK;
}

int eetq_pack_i8(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout, void* stream)
{
    return launch_pack(q_raw, K, N, q_packed, layout, static_cast<hipStream_t>(stream));
}

int eetq_unpack_i8(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout, void* stream)
{
    return launch_unpack(q_packed, K, N, q_raw, layout, static_cast<hipStream_t>(stream));
}

int eetq_quantize_i8_host(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed,
                          int layout, void* scales)
{
    EETQ_REQUIRE(w && scales, "null pointer");
    EETQ_REQUIRE(w_dtype == EETQ_DTYPE_F16 || w_dtype == EETQ_DTYPE_F32,
                 "Invalid datatype. Weight must be FP16 or FP32");
    EETQ_REQUIRE(K > 0 && N > 0, "weight should not be empty tensor");
    const size_t esz = w_dtype == EETQ_DTYPE_F16 ? 2 : 4;
    DevBuf       dw, draw, dpk, dsc, dmax;
    int          st;
    if ((st = dw.alloc(K * N * esz)) || (st = dsc.alloc(N * esz)) || (st = dmax.alloc(quantize_workspace_floats(K, N) * sizeof(float)))) return st;
    if (q_raw && (st = draw.alloc(K * N))) return st;
    if (q_packed && (st = dpk.alloc(K * N))) return st;
    EETQ_TRY_HIP(hipMemcpy(dw.p, w, K * N * esz, hipMemcpyHostToDevice));
    st = launch_quantize(dw.p, w_dtype, K, N, static_cast<int8_t*>(draw.p), static_cast<int8_t*>(dpk.p), layout,
                         dsc.p, static_cast<float*>(dmax.p), quantize_workspace_floats(K, N), nullptr);
    if (st != EETQ_OK) return st;
    EETQ_TRY_HIP(hipStreamSynchronize(nullptr));
    if (q_raw) EETQ_TRY_HIP(hipMemcpy(q_raw, draw.p, K * N, hipMemcpyDeviceToHost));
    if (q_packed) EETQ_TRY_HIP(hipMemcpy(q_packed, dpk.p, K * N, hipMemcpyDeviceToHost));
    EETQ_TRY_HIP(hipMemcpy(scales, dsc.p, N * esz, hipMemcpyDeviceToHost));
    return EETQ_OK;
}

static int relayout_host(const int8_t* src, size_t K, size_t N, int8_t* dst, int layout, bool pack)
{
    EETQ_REQUIRE(src && dst, "null pointer");
    EETQ_REQUIRE(K > 0 && N > 0, "weight should not be empty tensor");
    DevBuf a, b;
    int    st;
    if ((st = a.alloc(K * N)) || (st = b.alloc(K * N))) return st;
    EETQ_TRY_HIP(hipMemcpy(a.p, src, K * N, hipMemcpyHostToDevice));
    st = pack ? launch_pack(static_cast<int8_t*>(a.p), K, N, static_cast<int8_t*>(b.p), layout, nullptr)
              : launch_unpack(static_cast<int8_t*>(a.p), K, N, static_cast<int8_t*>(b.p), layout, nullptr);
    if (st != EETQ_OK) return st;
    EETQ_TRY_HIP(hipStreamSynchronize(nullptr));
    EETQ_TRY_HIP(hipMemcpy(dst, b.p, K * N, hipMemcpyDeviceToHost));
    return EETQ_OK;
}

int eetq_pack_i8_host(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout)
{
    return relayout_host(q_raw, K, N, q_packed, layout, true);
}

int eetq_unpack_i8_host(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout)
{
    return relayout_host(q_packed, K, N, q_raw, layout, false);
}

// ---- what AUTO launches --------------------------------------------------------------------------------------------------
// The reference switches on m alone (m <= 4: batched GEMV, else CUTLASS with a tile picked by occupancy queries at run time:
// fpA_intB_gemm_wrapper.cu:149-162, cutlass_heuristic.cc:123-206).  Here one pure function of (M, N, K, epilogue) names the
// kernel path; crossovers measured as graph-replayed chains (profiles/r03_path_compare_mid.jsonl, r04_stream_splitk_seam2.jsonl,
// r05_auto_regret.jsonl).  No environment reads except the operational EETQ_AMD_SPLITK=0 (no library-owned scratch).
struct AutoChoice {
    int path;    // EETQ_PATH_*
    int detail;  // TILESPLIT: K slices the launch uses (1 = the unsplit tiled kernel); SPLITK: row groups of the row-group plan (0 = K-slice plan); otherwise 0
};

static AutoChoice auto_path_i8(int M, int N, int K, int act)
{
    // M = 1 -> wave-reduction GEMV (no MFMA).  (Round 5 tried the small-batch kernel for K = 4096 with 1.3 < tile rows per CU <= 2 --
    // Llama-3-8B's fused q|k|v, 4096 x 6144: 6.8 -> 6.4 us, the one M = 1 point above 5 % in tools/auto_regret.py -- and took it
    // back: the GEMV entry points with fused prologues / epilogues (norm, glu8, the compiled decode layer) promise the bits of the
    // plain M = 1 projection, which therefore has to stay on the GEMV kernel too: tests/test_gpu_glu.py.)
    if (M == 1) return {EETQ_PATH_GEMV, 0};
    // One row tile (2 <= M <= 16): the MFMA stream kernel (same weight stream as the GEMV, activation rows through a per-wave
    // LDS ring or a per-workgroup copy: streamk.hip::pick_plan) on every shape (us stream / split-K at M = 16: 4096 x 11008
    // 12.60 / 12.74, 11008 x 4096 12.48 / 13.15, 8192^2 13.9 / 18.5, 28672 x 8192 41.8 / 46.9; the other way only 5120 x 27648
    // from M = 13, 29.9 / 29.0)
    static const bool use_splitk = [] {  // EETQ_AMD_SPLITK=0 keeps the unsplit kernels (the split forms own per-stream scratch)
        const char* e = getenv("EETQ_AMD_SPLITK");
        return !(e && e[0] == '0');
    }();
    if (M <= 16) {
        // ... except a narrow, deep weight from M = 9 (a GQA k / v projection: Llama-3-70B 8192 x 1024): the stream kernel has one
        // workgroup per 16 columns -- N <= 2048 leaves at least half the CUs without one -- and walks all of K in each, while the
        // split-K tile cuts K and fills the chip; from M = 9 the stream kernel also needs its 16-row ring.  tools/auto_regret.py
        // (profiles/r05_auto_regret_small_n.jsonl), us stream / split-K: 8192 x 1024 M = 9 7.4 / 6.8, M = 16 8.3 / 6.8; 7168 x 1024
        // M = 12 7.2 / 6.5; 14336 x 1024 M = 9 11.3 / 9.0, M = 16 13.2 / 9.0; 8192 x 2048 M = 12 8.0 / 7.6, M = 16 8.7 / 7.7.  A shallow
        // K keeps the stream kernel (4096 x 1024 M = 16 5.3 / 6.3, 5120 x 1280 M = 12 5.9 / 5.9), as does M <= 8 (8192 x 1024 6.6 / 6.8).
        if (M >= 9 && use_splitk && K >= 7168 && 2 * (N / kTileN) <= device_cu_count() && (size_t)N * K < (1ull << 31))
            return {EETQ_PATH_SPLITK, 0};
        return {EETQ_PATH_STREAM, 0};
    }
    if (M <= kMidMaxM && (size_t)M * K * 2 < (1ull << 31) && (size_t)N * K < (1ull << 31)) {
        // wide N, M > 64: the 128 x 64 tiles of the tiled kernel already give most CUs a workgroup and read the activations
        // once per 64 columns (N = 11008: M = 96 21.0 vs 23.9 us split-K, M = 128 22.4 vs 28.2; up to M = 64 the split-K tile
        // is ahead: 17.6 vs 18.5)
        // ... and from M = 33 once the narrow tiles alone give EVERY CU a workgroup (round 5, held-out shapes of
        // tools/auto_regret.py, us split-K / tiled at M = 48 and 64: 3584 x 18944 34.2 / 24.2, 35.3 / 24.9; 4096 x 28672 40.5 / 33.9,
        // 43.4 / 35.1; 8192 x 57344 153 / 127, 169 / 128 -- the split-K tile doubles its row blocks at M = 33 and needs a second
        // round of workgroups there; below 256 narrow tiles it stays ahead: 5120 x 13824 M = 64 22.7 / 25.9, 8192 x 10240 32.0 / 33.0)
        const int tiles1 = (N + 63) / 64;
        // (... except 65 <= M <= 96 where three 32-row groups of the split-K tile's 64-column blocks still fit the chip two per
        // CU, i.e. N <= 10880 -- Llama-3-70B's fused q|k|v, 8192 x 10240: M = 80 35.6 tiled vs 33.9 us, M = 96 39.1 vs 33.7-34.7;
        // 2560 x 10240 M = 72 14.5 vs 12.7, M = 96 16.2 vs 13.0 (profiles/r05_splitk_plan_regret_fitted.jsonl, ..._share.jsonl);
        // 4096 x 11008 does not fit and stays: M = 96 21.4 vs 23.5)
        const bool rows3 = M > 64 && M <= 96 && tiles1 * 3 <= 2 * device_cu_count() && use_splitk;
        if (K >= 320 && !rows3 && ((M > 64 && tiles1 >= 160) || (M > 32 && tiles1 >= device_cu_count()))) return {EETQ_PATH_MFMA, 0};
        // few tiles, M > 96, K deeper than 8192: K slices of the tiled kernel's 128 x 64 tile -- M = 128: 13824 x 5120 40.7 vs
        // 41.9 us the best split-K plan, 28672 x 8192 89 vs 98, 11008 x 4096 23.3 vs 23.6 (tile_splitk_slices has the rule).  Up to
        // K = 8192 the split-K tile's plans with row groups are ahead (profiles/r05_splitk_plan_regret_after.jsonl, us K-sliced
        // tiled kernel / split-K plan: 5120^2 19.9 / 18.4, 8192 x 4096 20.1 / 18.6, 4096 x 2048 10.8 / 8.3, 8192 x 2048 16.1 / 12.3).
        if (M > 96 && act == 0 && K > 8192) {
            const int S = tile_splitk_slices(M, N, K);
            if (S > 1) return {EETQ_PATH_TILESPLIT, use_splitk ? S : 1};
        }
        // narrower N: the split-K tile (K slices + in-launch deterministic reduction, and / or the batch cut into row groups:
        // splitk_plan; N = K = 4096: M = 64 8.7 us vs 11.2 round-1 tile).  (Until the planner learnt row groups, shallow-K shapes
        // with 128 < N / 32 <= 256 column blocks went to the round-1 tile at M <= 64 -- the same decomposition without slicing
        // machinery, 5-10 % ahead of the K-slice plans at M = 48 / 64.  Two 32-row groups are ahead of it now -- us round-1 tile /
        // split-K plan, profiles/r05_splitk_plan_regret_fitted.jsonl: 4096 x 6144 M = 48 11.2 / 10.9, M = 64 12.0 / 11.5; 4096 x 8192
        // M = 48 12.1 / 11.5; 3584 x 4608 M = 64 11.4 / 10.6 -- and at M <= 32 the two were within 1 % of each other: the rule is gone,
        // the round-1 tile stays as the path without library-owned scratch, EETQ_AMD_SPLITK=0.)
        if (!use_splitk) return {EETQ_PATH_MID, 0};
        // (the split-K launcher's plan may cut the batch into row groups on top of -- or instead of -- K slices: splitk_plan)
        int nb = 1, s = 1, stages = 2, rp = 1;
        splitk_plan(M, N, K, &nb, &s, &stages, &rp);
        return {EETQ_PATH_SPLITK, rp > 1 ? rp : 0};
    }
    // 128 < M <= 1024 on few-tile shapes with a K too shallow to slice: the split-K tile with the batch cut into row groups of
    // <= 64 rows, one round of workgroups, no reduction (gemm_splitk.hip::splitk_rows_plan: 4096^2 M = 256 20.8 -> 17.0 us,
    // 5120^2 M = 192 22.6 -> 21.3); needs no scratch, so EETQ_AMD_SPLITK=0 does not switch it off
    {
        int r = 0;
        if (splitk_rows_plan(M, N, K, &r) && (size_t)M * K * 2 < (1ull << 31) && (size_t)N * K < (1ull << 31)) return {EETQ_PATH_SPLITK, r};
    }
    // M > 128: the tiled kernel; with K slices when its tiles would leave most CUs idle (M = 256 at 11008 x 4096: 37.4 vs
    // 54.7 us) -- launch_gemm_tile_splitk applies the same rule and runs the unsplit kernel otherwise
    int S = (use_splitk && act == 0) ? tile_splitk_slices(M, N, K) : 1;
    if (S == 1 && use_splitk && act == 0 && wide_tile_splitk_slices(M, N, K) == 2) S = 2;
    return {EETQ_PATH_TILESPLIT, S};
}

int eetq_diag_auto_path(int bits, int M, int N, int K, int* path, int* detail)
{
    EETQ_REQUIRE(path && M >= 1 && N >= kTileN && K >= kTileK, "eetq_diag_auto_path: bad argument");
    AutoChoice c{EETQ_PATH_AUTO, 0};
    if (bits == 8) {
        c = auto_path_i8(M, N, K, EETQ_ACT_IDENTITY);
    } else if (bits == 4) {
        c.path = w4a16_auto_path(M, N, K);
    } else {
        return fail(EETQ_ERR_INVALID, "[eetq_amd] eetq_diag_auto_path: bits must be 8 or 4");
    }
    *path = c.path;
    if (detail) *detail = c.detail;
    return EETQ_OK;
}

int eetq_diag_splitk_plan(int M, int N, int K, int* column_blocks, int* k_slices, int* ring, int* row_groups)
{
    EETQ_REQUIRE(column_blocks && k_slices && ring && row_groups, "eetq_diag_splitk_plan: null pointer");
    EETQ_REQUIRE(M >= 1 && M <= kSplitkMaxM && N >= kTileN && K >= kTileK && K % 64 == 0, "eetq_diag_splitk_plan: bad argument");
    int stages = 2;
    splitk_plan(M, N, K, column_blocks, k_slices, &stages, row_groups);
    *ring = 11 * stages;
    return EETQ_OK;
}

static int gemm_dispatch(const void* x, const int8_t* w_packed, const void* scales, const void* bias,
                         const void* residual, void* y, int M, int N, int K, int path, void* stream, int act = EETQ_ACT_IDENTITY)
{
    int st = check_gemm_args(x, w_packed, scales, y, M, N, K);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(!bias || (uintptr_t)bias % 8 == 0, "bias must be 8-byte aligned");
    EETQ_REQUIRE(!residual || (uintptr_t)residual % 16 == 0, "residual must be 16-byte aligned");
    const f16*     xp = static_cast<const f16*>(x);
    const uint8_t* wp = reinterpret_cast<const uint8_t*>(w_packed);
    const f16*     sp = static_cast<const f16*>(scales);
    Epilogue       bp;
    bp.bias     = static_cast<const f16*>(bias);
    bp.residual = static_cast<const f16*>(residual);
    EETQ_REQUIRE(act >= EETQ_ACT_IDENTITY && act <= EETQ_ACT_SILU, "Invalid activation type.");
    bp.act         = act;
    f16*           yp = static_cast<f16*>(y);
    hipStream_t    s  = static_cast<hipStream_t>(stream);
    switch (path) {
        case EETQ_PATH_AUTO: {
            // ONE heuristic (auto_path_i8 above), like the reference's (cutlass_heuristic.cc:123-206) -- and visible from outside
            // through eetq_diag_auto_path, which calls the same function
            const AutoChoice c = auto_path_i8(M, N, K, act);
            switch (c.path) {
                case EETQ_PATH_GEMV: return launch_gemv(xp, wp, sp, bp, yp, M, N, K, s);
                case EETQ_PATH_STREAM: return launch_streamk(xp, wp, sp, bp, yp, M, N, K, s);
                case EETQ_PATH_MFMA: return launch_gemm_mfma(xp, wp, sp, bp, yp, M, N, K, s);
                case EETQ_PATH_SPLITK: return launch_gemm_splitk(xp, wp, sp, bp, yp, M, N, K, s);
                case EETQ_PATH_MID: return launch_gemm_mid(xp, wp, sp, bp, yp, M, N, K, s);
                default: return launch_gemm_tile_splitk(xp, wp, sp, bp, yp, M, N, K, s);
            }
        }
        case EETQ_PATH_GEMV: return launch_gemv(xp, wp, sp, bp, yp, M, N, K, s);
        case EETQ_PATH_MFMA: return launch_gemm_mfma(xp, wp, sp, bp, yp, M, N, K, s);
        case EETQ_PATH_STREAM: return launch_streamk(xp, wp, sp, bp, yp, M, N, K, s);
        case EETQ_PATH_MID: return launch_gemm_mid(xp, wp, sp, bp, yp, M, N, K, s);
        case EETQ_PATH_SPLITK: return launch_gemm_splitk(xp, wp, sp, bp, yp, M, N, K, s, 0, 0, /*env_plan=*/true);
        case EETQ_PATH_TILESPLIT: return launch_gemm_tile_splitk(xp, wp, sp, bp, yp, M, N, K, s, 0, nullptr, /*env_plan=*/true);
        default: return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] unknown or unimplemented GEMM path");
    }
}

int eetq_w8a16_gemm_ex(const void* x, const int8_t* w_packed, const void* scales, void* y, int M, int N, int K,
                       int path, void* stream)
{
    return gemm_dispatch(x, w_packed, scales, nullptr, nullptr, y, M, N, K, path, stream);
}

int eetq_w8a16_gemm(const void* x, const int8_t* w_packed, const void* scales, void* y, int M, int N, int K,
                    void* stream)
{
    return gemm_dispatch(x, w_packed, scales, nullptr, nullptr, y, M, N, K, EETQ_PATH_AUTO, stream);
}

int eetq_w8a16_gemm_bias(const void* x, const int8_t* w_packed, const void* scales, const void* bias, void* y, int M,
                         int N, int K, int path, void* stream)
{
    return gemm_dispatch(x, w_packed, scales, bias, nullptr, y, M, N, K, path, stream);
}

int eetq_w8a16_gemm_fused(const void* x, const int8_t* w_packed, const void* scales, const void* bias,
                          const void* residual, void* y, int M, int N, int K, int path, void* stream)
{
    return gemm_dispatch(x, w_packed, scales, bias, residual, y, M, N, K, path, stream);
}

int eetq_w8a16_gemm_act(const void* x, const int8_t* w_packed, const void* scales, const void* bias, const void* residual,
                        void* y, int M, int N, int K, int path, int act, void* stream)
{
    return gemm_dispatch(x, w_packed, scales, bias, residual, y, M, N, K, path, stream, act);
}

int eetq_quantize_i4(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed, int layout,
                     void* scales, float* workspace, void* stream)
{
    if (!workspace) {
        int st = colmax_scratch(N, &workspace);
        if (st != EETQ_OK) return st;
    }
    return launch_quantize_i4(w, w_dtype, K, N, q_raw, q_packed, layout, scales, workspace, static_cast<hipStream_t>(stream));
}

int eetq_pack_i4(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout, void* stream)
{
    return launch_pack_i4(q_raw, K, N, q_packed, layout, static_cast<hipStream_t>(stream));
}

int eetq_unpack_i4(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout, void* stream)
{
    return launch_unpack_i4(q_packed, K, N, q_raw, layout, static_cast<hipStream_t>(stream));
}

int eetq_w4a16_gemm(const void* x, const int8_t* w_packed, const void* scales, const void* bias, const void* residual,
                    void* y, int M, int N, int K, void* stream)
{
    return eetq_w4a16_gemm_ex(x, w_packed, scales, bias, residual, y, M, N, K, EETQ_PATH_AUTO, stream);
}

int eetq_w4a16_gemm_ex(const void* x, const int8_t* w_packed, const void* scales, const void* bias, const void* residual,
                       void* y, int M, int N, int K, int path, void* stream)
{
    EETQ_REQUIRE(x && w_packed && scales && y, "null pointer");
    EETQ_REQUIRE(M >= 1 && N >= 1 && K >= 1, "invalid GEMM shape");
    EETQ_REQUIRE(K % 128 == 0, "int4: k must be a multiple of 128");
    EETQ_REQUIRE(N % 16 == 0, "n must be a multiple of 16");
    EETQ_REQUIRE(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y) % 16 == 0, "x, weight and y must be 16-byte aligned");
    Epilogue ep;
    ep.bias     = static_cast<const f16*>(bias);
    ep.residual = static_cast<const f16*>(residual);
    return launch_w4a16(static_cast<const f16*>(x), reinterpret_cast<const uint8_t*>(w_packed),
                        static_cast<const f16*>(scales), ep, static_cast<f16*>(y), M, N, K, static_cast<hipStream_t>(stream),
                        path);
}

int eetq_rmsnorm_f16(const void* x, const void* gamma, void* out, float eps, int rows, int cols, void* stream)
{
    return launch_rmsnorm(static_cast<const f16*>(x), static_cast<const f16*>(gamma), static_cast<f16*>(out), eps,
                          rows, cols, static_cast<hipStream_t>(stream));
}

int eetq_rotary_neox_f16(const int64_t* positions, void* query, void* key, const void* cos_sin_cache, int tokens,
                         int heads, int head_size, int rot_dim, void* stream)
{
    return launch_rotary(positions, static_cast<f16*>(query), static_cast<f16*>(key),
                         static_cast<const f16*>(cos_sin_cache), tokens, heads, heads, head_size, rot_dim,
                         heads * head_size, heads * head_size, static_cast<hipStream_t>(stream));
}

int eetq_w8a16_gemv_grouped(const eetq_gemv_problem* problems, int count, void* stream)
{
    EETQ_REQUIRE(count >= 0 && (problems || count == 0), "grouped GEMV: null problem array");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < count; ++i) {
        const eetq_gemv_problem& p = problems[i];
        int st = check_gemm_args(p.x, p.w_packed, p.scales, p.y, 1, p.N, p.K);
        if (st != EETQ_OK) return st;
        EETQ_REQUIRE(!p.bias || (uintptr_t)p.bias % 8 == 0, "bias must be 8-byte aligned");
        EETQ_REQUIRE(!p.residual || (uintptr_t)p.residual % 16 == 0, "residual must be 16-byte aligned");
    }
    std::vector<char> done(count, 0);
    for (int i = 0; i < count; ++i) {
        if (done[i]) continue;
        const int K = problems[i].K;
        if (!gemv_grouped_supports(K)) {  // one by one through the ordinary dispatcher
            done[i] = 1;
            int st = gemm_dispatch(problems[i].x, problems[i].w_packed, problems[i].scales, problems[i].bias,
                                   problems[i].residual, problems[i].y, 1, problems[i].N, K, EETQ_PATH_AUTO, stream);
            if (st != EETQ_OK) return st;
            continue;
        }
        gemv::GroupedArgs g;
        g.count  = 0;
        int rows = 0;
        for (int j = i; j < count; ++j) {  // every not yet launched problem of this K, kMaxGroup per dispatch
            if (done[j] || problems[j].K != K) continue;
            done[j] = 1;
            gemv::GroupedProblem& q = g.p[g.count++];
            q.x         = static_cast<const f16*>(problems[j].x);
            q.w         = reinterpret_cast<const uint8_t*>(problems[j].w_packed);
            q.scales    = static_cast<const f16*>(problems[j].scales);
            q.y         = static_cast<f16*>(problems[j].y);
            q.bias      = static_cast<const f16*>(problems[j].bias);
            q.residual  = static_cast<const f16*>(problems[j].residual);
            q.N         = problems[j].N;
            q.first_row = rows;
            rows += problems[j].N / kTileN;
            if (g.count == gemv::kMaxGroup) {
                int st = launch_gemv_grouped(g, K, rows, s);
                if (st != EETQ_OK) return st;
                g.count = 0;
                rows    = 0;
            }
        }
        if (g.count) {
            int st = launch_gemv_grouped(g, K, rows, s);
            if (st != EETQ_OK) return st;
        }
    }
    return EETQ_OK;
}

int eetq_rotary_neox(const int64_t* positions, void* query, void* key, const void* cos_sin_cache, int dtype, int tokens,
                     int heads, int head_size, int rot_dim, void* stream)
{
    return launch_rotary_any(positions, query, key, cos_sin_cache, dtype, tokens, heads, heads, head_size, rot_dim,
                             heads * head_size, heads * head_size, static_cast<hipStream_t>(stream));
}

int eetq_rotary_neox_strided_f16(const int64_t* positions, void* query, void* key, const void* cos_sin_cache,
                                 int tokens, int q_heads, int k_heads, int head_size, int rot_dim, int q_stride,
                                 int k_stride, void* stream)
{
    return launch_rotary(positions, static_cast<f16*>(query), static_cast<f16*>(key),
                         static_cast<const f16*>(cos_sin_cache), tokens, q_heads, k_heads, head_size, rot_dim, q_stride,
                         k_stride, static_cast<hipStream_t>(stream));
}

int eetq_w8a16_gemv_rmsnorm(const void* x, const void* gamma, float eps, const int8_t* w_packed, const void* scales,
                            const void* bias, const void* residual, void* y, int N, int K, void* stream)
{
    int st = check_gemm_args(x, w_packed, scales, y, 1, N, K);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(gamma, "null pointer");
    EETQ_REQUIRE((uintptr_t)gamma % 16 == 0, "gamma must be 16-byte aligned");
    Epilogue ep;
    ep.bias     = static_cast<const f16*>(bias);
    ep.residual = static_cast<const f16*>(residual);
    Prologue pro;
    pro.gamma = static_cast<const f16*>(gamma);
    pro.eps   = eps;
    return launch_gemv(static_cast<const f16*>(x), reinterpret_cast<const uint8_t*>(w_packed),
                       static_cast<const f16*>(scales), ep, static_cast<f16*>(y), 1, N, K, static_cast<hipStream_t>(stream),
                       pro);
}

int eetq_w8a16_gemv_silu_gated(const void* gate_up, const int8_t* w_packed, const void* scales, const void* bias,
                               const void* residual, void* y, int N, int K, void* stream)
{
    int st = check_gemm_args(gate_up, w_packed, scales, y, 1, N, K);
    if (st != EETQ_OK) return st;
    Epilogue ep;
    ep.bias     = static_cast<const f16*>(bias);
    ep.residual = static_cast<const f16*>(residual);
    Prologue pro;
    pro.up = static_cast<const f16*>(gate_up) + K;
    return launch_gemv(static_cast<const f16*>(gate_up), reinterpret_cast<const uint8_t*>(w_packed),
                       static_cast<const f16*>(scales), ep, static_cast<f16*>(y), 1, N, K, static_cast<hipStream_t>(stream),
                       pro);
}

int eetq_w8a16_gemv_glu8(const void* x, const void* gamma, float eps, const int8_t* w_packed, const void* scales,
                         const void* bias, void* y, int N, int K, void* stream)
{
    int st = check_gemm_args(x, w_packed, scales, y, 1, N, K);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(gamma == nullptr || (uintptr_t)gamma % 16 == 0, "gamma must be 16-byte aligned");
    Epilogue ep;
    ep.bias = static_cast<const f16*>(bias);
    ep.act  = kActGlu8;
    Prologue pro;
    pro.gamma = static_cast<const f16*>(gamma);
    pro.eps   = eps;
    return launch_gemv(static_cast<const f16*>(x), reinterpret_cast<const uint8_t*>(w_packed),
                       static_cast<const f16*>(scales), ep, static_cast<f16*>(y), 1, N, K, static_cast<hipStream_t>(stream),
                       pro);
}

int eetq_w8a16_gemm_glu8(const void* x, const int8_t* w_packed, const void* scales, const void* bias, void* y, int M, int N,
                         int K, void* stream)
{
    int st = check_gemm_args(x, w_packed, scales, y, M, N, K);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(!bias || (uintptr_t)bias % 8 == 0, "bias must be 8-byte aligned");
    Epilogue ep;
    ep.bias = static_cast<const f16*>(bias);
    ep.act  = kActGlu8;
    const auto xp = static_cast<const f16*>(x);
    const auto wp = reinterpret_cast<const uint8_t*>(w_packed);
    const auto sp = static_cast<const f16*>(scales);
    if (M == 1) return launch_gemv(xp, wp, sp, ep, static_cast<f16*>(y), 1, N, K, static_cast<hipStream_t>(stream));
    if (M <= 16) return launch_streamk(xp, wp, sp, ep, static_cast<f16*>(y), M, N, K, static_cast<hipStream_t>(stream));
    // larger batches (prompts): where AUTO runs the plain projection on the tiled MFMA kernel, that kernel writes the activation
    // out of its fp16 tile image; the other kernels have no gated write-out
    st = EETQ_ERR_UNSUPPORTED;
    const AutoChoice c = auto_path_i8(M, N, K, 0);
    if ((c.path == EETQ_PATH_MFMA || (c.path == EETQ_PATH_TILESPLIT && c.detail == 1)) && (uintptr_t)y % 16 == 0)
        st = launch_gemm_mfma(xp, wp, sp, ep, static_cast<f16*>(y), M, N, K, static_cast<hipStream_t>(stream));
    if (st == EETQ_ERR_UNSUPPORTED)
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] eetq_w8a16_gemm_glu8: no gated write-out for this shape (M > 16 off the tiled MFMA "
                                          "kernel); run eetq_w8a16_gemm_act, then eetq_silu_mul_glu8_f16");
    return st;
}

int eetq_silu_mul_glu8_f16(const void* gate_up, void* out, int rows, int intermediate, void* stream)
{
    return launch_silu_mul(static_cast<const f16*>(gate_up), static_cast<f16*>(out), rows, intermediate,
                           static_cast<hipStream_t>(stream), true);
}

int eetq_silu_mul_f16(const void* gate_up, void* out, int rows, int intermediate, void* stream)
{
    return launch_silu_mul(static_cast<const f16*>(gate_up), static_cast<f16*>(out), rows, intermediate,
                           static_cast<hipStream_t>(stream));
}

int eetq_rotary_neox_kvcache_f16(const int64_t* positions, const int64_t* slots, int slot_stride, void* query,
                                 const void* key, const void* value, const void* cos_sin_cache, void* k_cache,
                                 void* v_cache, int batch, int q_heads, int k_heads, int head_size, int rot_dim,
                                 const long* strides, int max_positions, void* stream)
{
    EETQ_REQUIRE(strides, "null pointer");
    EETQ_REQUIRE(slot_stride == 0 || slot_stride == 1, "slot_stride must be 0 (one slot for the batch) or 1");
    return launch_rotary_kvcache(positions, slots, slot_stride, static_cast<f16*>(query), static_cast<const f16*>(key),
                                 static_cast<const f16*>(value), static_cast<const f16*>(cos_sin_cache),
                                 static_cast<f16*>(k_cache), static_cast<f16*>(v_cache), batch, q_heads, k_heads, head_size,
                                 rot_dim, strides[0], strides[1], strides[2], strides[3], strides[4], strides[5],
                                 max_positions, static_cast<hipStream_t>(stream));
}

int eetq_rotary_neox_kvcache_prefill_f16(const int64_t* positions, void* query, const void* key, const void* value,
                                         const void* cos_sin_cache, void* k_cache, void* v_cache, int batch, int tokens,
                                         const int64_t* first_row_dev, int first_row, int q_heads, int k_heads, int head_size,
                                         int rot_dim, const long* strides, int max_positions, void* stream)
{
    EETQ_REQUIRE(strides, "null pointer");
    EETQ_REQUIRE(batch >= 0 && tokens >= 0, "invalid shape");
    if (batch == 0 || tokens == 0) return EETQ_OK;
    EETQ_REQUIRE((long)batch * tokens < (1L << 31), "too many tokens for one launch");
    return launch_rotary_kvcache(positions, first_row_dev, 0, static_cast<f16*>(query), static_cast<const f16*>(key),
                                 static_cast<const f16*>(value), static_cast<const f16*>(cos_sin_cache),
                                 static_cast<f16*>(k_cache), static_cast<f16*>(v_cache), batch, q_heads, k_heads, head_size,
                                 rot_dim, strides[0], strides[1], strides[2], strides[3], strides[4], strides[5],
                                 max_positions, static_cast<hipStream_t>(stream), tokens, first_row);
}

int eetq_greedy_handover_f16(const void* logits, long row_stride, int vocab, int batch, int64_t* out_tokens, long out_stride,
                             int out_cols, int64_t* column, int64_t* next_token, int64_t* position, void* stream)
{
    return eetq::launch_greedy_handover(static_cast<const f16*>(logits), row_stride, vocab, batch, out_tokens, out_stride, out_cols,
                                        column, next_token, position, static_cast<hipStream_t>(stream));
}

int eetq_rope_decode_attention_f16(const int64_t* positions, const int64_t* slots, int slot_stride, const void* query,
                                   const void* key, const void* value, const void* cos_sin_cache, void* k_cache,
                                   void* v_cache, const void* mask, void* out, float* workspace, unsigned* tickets,
                                   int batch, int heads, int kv_heads, int max_positions, int head_dim, int splits,
                                   float scaling, const long* strides, const int64_t* kv_len, int kv_len_bias,
                                   int64_t* advance, void* stream)
{
    return launch_rope_attn_decode(positions, slots, slot_stride, static_cast<const f16*>(query),
                                   static_cast<const f16*>(key), static_cast<const f16*>(value),
                                   static_cast<const f16*>(cos_sin_cache), static_cast<f16*>(k_cache),
                                   static_cast<f16*>(v_cache), static_cast<const f16*>(mask), static_cast<f16*>(out),
                                   workspace, tickets, batch, heads, kv_heads, max_positions, head_dim, splits, scaling,
                                   strides, kv_len, kv_len_bias, advance, static_cast<hipStream_t>(stream));
}

int eetq_prefill_attention_f16(const void* q, const void* k, const void* v, void* out, int batch, int heads, int kv_heads, int q_tokens,
                               int keys, int head_dim, int causal_offset, float scaling, const long* strides, void* stream)
{
    return launch_prefill_attention(static_cast<const f16*>(q), static_cast<const f16*>(k), static_cast<const f16*>(v),
                                    static_cast<f16*>(out), batch, heads, kv_heads, q_tokens, keys, head_dim, causal_offset, scaling,
                                    strides, static_cast<hipStream_t>(stream));
}

int eetq_prefill_attention_supported(int head_dim) { return prefill_attention_supports(head_dim) ? 1 : 0; }

int eetq_decode_dropped_steps(unsigned long long* count, int reset)
{
    EETQ_REQUIRE(count, "null pointer");
    EETQ_TRY_HIP(hipDeviceSynchronize());
    unsigned a = 0, r = 0;
    int      st = attn_dropped_steps(&a, reset != 0);
    if (st != EETQ_OK) return st;
    st = rope_dropped_steps(&r, reset != 0);
    if (st != EETQ_OK) return st;
    *count = (unsigned long long)a + r;
    return EETQ_OK;
}

int eetq_diag_attn_stamps(unsigned long long* stamps)
{
    set_attn_stamps(stamps);
    return EETQ_OK;
}

int eetq_decode_attention_splits(int batch, int heads, int positions)
{
    if (batch <= 0 || heads <= 0 || positions <= 0) return 1;
    const long bh = (long)batch * heads, cus = device_cu_count();
    // about 0.85 workgroups per CU -- a bare read of the same 21 MB: 200 / 240 workgroups 4.7 - 4.8 us, 320 (a quarter of the CUs
    // with two) 5.7, 400 5.5 (profiles/r06_attn_probe.txt); the kernel itself at 40 heads: 5 chunks 9.5, 6 9.9, 8 11.7 us -- and
    // twice / four times that when a chunk would otherwise hold more than 512 rows (batch 4: 3 chunks 20.5 us, 2 23.5, 4 23.3)
    long splits = 1;
    for (long k = 1; k <= 4; k *= 2) {
        splits = (17 * k * cus / 20 + bh / 2) / bh;  // round(0.85 k CUs / (batch heads))
        if (splits < 1) splits = 1;
        if (((long)positions + splits - 1) / splits <= 512) break;
    }
    if (splits > 16) splits = 16;
    const long cap    = ((long)positions + 63) / 64;
    if (splits > cap) splits = cap;
    return (int)(splits < 1 ? 1 : splits);
}

int eetq_decode_attention_f16(const void* q, const void* k_cache, const void* v_cache, const void* mask, void* out,
                              float* workspace, int batch, int heads, int kv_heads, int positions, int head_dim,
                              int splits, float scaling, const long* strides, const int64_t* kv_len, int kv_len_bias,
                              int64_t* advance, void* stream)
{
    return launch_attn_decode(static_cast<const f16*>(q), static_cast<const f16*>(k_cache),
                              static_cast<const f16*>(v_cache), static_cast<const f16*>(mask), static_cast<f16*>(out),
                              workspace, batch, heads, kv_heads, positions, head_dim, splits, scaling, strides, kv_len,
                              kv_len_bias, advance, static_cast<hipStream_t>(stream));
}

}  // extern "C"
